/*
 * hdp_hip.h -- C ABI of libhdp_hip.so: the MI355X (gfx950) implementation of the
 * two data-parallel hot loops of the Heatwave Diagnostics Package.
 *
 * The reference (AgentOxygen/HDP) is pure Python + Numba and has no FFI layer;
 * the array-level boundary it exposes is the two Numba kernels
 *
 *   compute_percentiles        gufunc '(t),(d,b),(p)->(d,p)'   hdp/threshold.py:52-78
 *   compute_heatwave_metrics   njit, called through apply_ufunc with core dims
 *                              (time),(doy),(time),(),(),(),(year,end_points)->(metric,year)
 *                                                              hdp/metric.py:304-341,360-366
 *
 * plus the njit helpers the reference unit-tests directly (metric.py:11,63,85,105,140).
 * Every entry point below replaces one of those call sites; the Python host code in
 * hdp_amd/ binds them with ctypes (see INTEGRATION.md for the stub a maintainer of
 * the reference would add).
 *
 * Conventions
 *   - plain pointers and sizes only; all arrays are caller-allocated.
 *   - every function returns 0 on success or a negative HDP_E* code; the message is
 *     available from hdp_last_error() (thread-local).
 *   - "host" entry points take host pointers and block until the result is in the
 *     caller's buffer.  "_dev" entry points take device pointers, enqueue on the
 *     given hipStream_t (passed as void*, NULL = the library's stream) and do not
 *     synchronise.
 *   - a series ("cell") is one grid cell's time series; n_cells flattens every
 *     non-time dimension (lat, lon[, member]).
 *   - there is no CPU fallback: without a usable HIP device every compute entry
 *     point fails with HDP_ENODEV.
 */
#ifndef HDP_HIP_H
#define HDP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HDP_OK        0
#define HDP_EINVAL   -1   /* bad argument (shape, range, NULL) */
#define HDP_ENODEV   -2   /* no HIP device / hdp_init not called */
#define HDP_EHIP     -3   /* a HIP runtime call failed */
#define HDP_ENOMEM   -4   /* device or host allocation failed */
#define HDP_EUNSUP   -5   /* valid request outside what the kernels support */
#define HDP_EQUANT   -6   /* quantile outside [0,1]: numba raises ValueError here */

/* ---- lifecycle ----------------------------------------------------------- */

/* Select `device` (ordinal among visible HIP devices) for the calling process and
 * create the library stream.  One process drives one GPU (one rank per GPU). */
int hdp_init(int device);
int hdp_shutdown(void);
int hdp_device_count(void);
const char *hdp_last_error(void);
/* "gfx950 ... CUs ... LDS" description of the active device (static storage). */
const char *hdp_device_info(void);

/* ---- device memory plumbing (so a Python host needs no other GPU library) -- */
void *hdp_dev_alloc(size_t bytes);
int hdp_dev_free(void *p);
int hdp_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes);
int hdp_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes);
int hdp_dev_memset(void *dst_dev, int value, size_t bytes);
int hdp_sync(void *stream);
/* hipEvent-based timing on `stream`: returns an opaque handle / elapsed ms. */
void *hdp_event_create(void);
int hdp_event_record(void *event, void *stream);
int hdp_event_elapsed_ms(void *start, void *stop, float *ms); /* synchronises on stop */
int hdp_event_destroy(void *event);

/* ---- thresholds: replaces compute_percentiles (threshold.py:52-78) -------- */

/*
 * Plan = device-resident tables for one (calendar, window, quantile set):
 *   time_index [n_doy][S]  int64  time indices of the samples of each day-of-year
 *                                 row, -1 padded (the table threshold.py:35-39 builds;
 *                                 -1 samples the LAST time step, as NumPy indexing does)
 *   cols       [n_doy][W]  int32  for window row d, the W day-of-year rows whose
 *                                 samples form it (threshold.py:43-48; repeats allowed,
 *                                 this is where the reflected upper edge lives)
 *   q          [P]         double quantiles in [0,1]
 * Expanding time_index[cols[d][w]] over w reproduces row d of the reference's
 * [n_doy, W*S] gather table exactly.
 */
typedef struct hdp_threshold_plan hdp_threshold_plan;

int hdp_threshold_plan_create(const int64_t *time_index, int64_t n_doy, int64_t S,
                              const int32_t *cols, int64_t W,
                              const double *q, int64_t P, int64_t T,
                              hdp_threshold_plan **plan_out);
int hdp_threshold_plan_destroy(hdp_threshold_plan *plan);
/* Allocate up front whatever scratch a launch of up to n_cells cells would allocate on first use (the tiered image's
 * global tail; with time_major != 0 also the staging buffers, copy stream and events of hdp_thresholds_f32_tm_dev), so
 * that later launches neither synchronise nor call hipMalloc.  Needs hdp_init. */
int hdp_threshold_plan_reserve(hdp_threshold_plan *plan, int64_t n_cells, int time_major);
/* Human-readable name and shape of the kernel a launch of this plan runs (the HDP_THR_* environment switches are
 * read ONCE, when the plan is created, never at launch); thread-local storage, valid until the next call.
 *
 * Concurrency: a plan owns scratch that its launches write (the tiered image's global tail of the whole-cell kernel,
 * the staging buffers, copy stream and events of the time-major path) -- ONE launch of a plan in flight at a time.
 * Launches of one plan on one stream are ordered and safe; two streams or two host threads need two plans.  The
 * first launch of a plan may allocate that scratch (a stream synchronisation + hipMalloc); later launches with no
 * more cells than any earlier one allocate nothing. */
const char *hdp_threshold_plan_describe(const hdp_threshold_plan *plan);

/* x_dev [n_cells][T] float32 time-contiguous -> out_dev [n_cells][P][n_doy] float64.
 * PERCENTILE-MAJOR: the device layout, what hdp_metrics_f32_dev consumes (day-of-year rows are the
 * kernels' lanes, so both sides of the hand-off are coalesced).  The host entry point below returns
 * the reference's (cell, doy, percentile) order. */
int hdp_thresholds_f32_dev(const hdp_threshold_plan *plan, const float *x_dev,
                           int64_t n_cells, double *out_dev, void *stream);

/* Host buffers, arbitrary element strides (in elements) for cell and time. */
int hdp_thresholds_f32(const float *x, int64_t n_cells, int64_t T,
                       int64_t stride_cell, int64_t stride_time,
                       const int64_t *time_index, int64_t n_doy, int64_t S,
                       const int32_t *cols, int64_t W,
                       const double *q, int64_t P, double *out /* [n_cells][n_doy][P] */);

/* Literal gufunc operands (threshold.py:53-57): win [n_doy][B] int64 gather table
 * (negative indices wrap like NumPy).  Slow general path: one full sort per window;
 * used for unit-level parity and as an independent check of the plan kernel. */
int hdp_percentiles_table_f32(const float *x, int64_t n_cells, int64_t T,
                              int64_t stride_cell, int64_t stride_time,
                              const int64_t *win, int64_t n_doy, int64_t B,
                              const double *q, int64_t P, double *out);

/* ---- metrics: replaces compute_heatwave_metrics (metric.py:304-341) --------- */

/*
 * Plan = tables shared by every cell:
 *   doy_map [T]     int64  threshold row of each time step (metric.py:265-277)
 *   defs    [D][3]  int64  (min_duration, max_break, max_subs)  (metric.py:376-379)
 *   north, south [Y][2] int64 season [start,end) time indices per hemisphere
 *                          (metric.py:221-243).  Increasing, disjoint tables (what compute_hemisphere_ranges
 *                          builds) run the streaming kernels; any other table (overlapping or unordered ranges,
 *                          which compute_heatwave_metrics accepts) runs a slower per-series path.
 */
typedef struct hdp_metrics_plan hdp_metrics_plan;

int hdp_metrics_plan_create(const int64_t *doy_map, int64_t T, int64_t n_doy,
                            const int64_t *defs, int64_t D,
                            const int64_t *north, const int64_t *south, int64_t Y,
                            int64_t P, hdp_metrics_plan **plan_out);
int hdp_metrics_plan_destroy(hdp_metrics_plan *plan);
/* The default (split) metrics path keeps the exceedance words of two batches of series in an HBM
 * scratch owned by the plan (P * ceil(T/2048)*256 bytes per series, at most 2 x 4 GiB; the streaming
 * exceedance kernel of one batch runs on a stream of the plan's beside the state-machine kernel of the
 * previous batch, forked from and joined back into the caller's stream).  It is
 * allocated on first use; call this once up front to keep hdp_metrics_f32_dev free of
 * allocations (stream capture, latency-sensitive callers).  Not thread-safe per plan. */
int hdp_metrics_plan_reserve(hdp_metrics_plan *plan, int64_t n_cells);
/* Series per batch of the split path for a call of n_cells series (n_cells itself when the call is one batch). */
int64_t hdp_metrics_plan_batch_cells(const hdp_metrics_plan *plan, int64_t n_cells);
/* Years (seasons) per series in the device output: the Y of the plan.  Kept for sizing the output
 * buffer: 4 * P * D * hdp_metrics_year_pitch(plan) * n_cells int16 elements. */
int64_t hdp_metrics_year_pitch(const hdp_metrics_plan *plan);

/*
 * x_dev [n_cells][T] f32, thr_dev [n_thr_cells][P][n_doy] f64 (the layout hdp_thresholds_f32_dev
 * writes; the host entry point takes the reference's [n_thr_cells][n_doy][P]) where the thresholds of
 * cell c are row (c % n_thr_cells) (ensemble members share their cell's thresholds
 * when series are ordered member-major), is_south_dev [n_cells] u8 ->
 * out_dev [4][P][D][Y][n_cells] int16 (series-minor: a lane of the state-machine kernel is a series, so a
 * season's results leave as contiguous 2-byte values), metric order HWF, HWN, HWD, HWA
 * (metric.py:336-340).  Values are bounded by the season length (< 32768).
 */
int hdp_metrics_f32_dev(const hdp_metrics_plan *plan, const float *x_dev,
                        const double *thr_dev, int64_t n_thr_cells,
                        const uint8_t *is_south_dev, int64_t n_cells,
                        int16_t *out_dev, void *stream);

/* Human-readable name of the kernels a launch of this plan runs (thread-local storage). */
const char *hdp_metrics_plan_describe(const hdp_metrics_plan *plan);

/* ---- time-major device inputs --------------------------------------------------------------------
 * CMIP data come as (time, lat, lon) (the reference's own workflow chunks them time = -1, lat, lon:
 * docs/example_cmip_workflow/run_cmip_workflow.py:31-32).  x_tm_dev is [T][pitch_cells] float32, element (t, c) at
 * x_tm_dev[t * pitch_cells + c]; outputs as for the series-major entry points.  Chunks of cells are transposed into
 * plan-owned series-major staging buffers (at most 2 x 5 GiB, allocated on first use) on a copy stream of the plan
 * while the kernels work on the previous chunk; fork/join with `stream` is by events. */
int hdp_thresholds_f32_tm_dev(const hdp_threshold_plan *plan, const float *x_tm_dev, int64_t pitch_cells,
                              int64_t n_cells, double *out_dev, void *stream);
int hdp_metrics_f32_tm_dev(const hdp_metrics_plan *plan, const float *x_tm_dev, int64_t pitch_cells,
                           const double *thr_dev, int64_t n_thr_cells, const uint8_t *is_south_dev,
                           int64_t n_cells, int16_t *out_dev, void *stream);

/* Host buffers; out [P][D][n_cells][4][Y] int16 (the reference's block layout,
 * metric.py:368-369, narrowed; the Python adapter widens to int64). */
int hdp_metrics_f32(const float *x, int64_t n_cells, int64_t T,
                    int64_t stride_cell, int64_t stride_time,
                    const double *thr, int64_t n_thr_cells, int64_t n_doy, int64_t P,
                    const int64_t *doy_map, const int64_t *defs, int64_t D,
                    const int64_t *north, const int64_t *south,
                    const uint8_t *is_south, int64_t Y, int16_t *out);

/* Same call; out [4][P][D][n_cells][Y] int64: one contiguous plane per output variable HWF, HWN, HWD, HWA in
 * the dims compute_individual_metrics gives them, (percentile, definition, cells..., time), and in its dtype
 * (metric.py:418-431) -- widened and regrouped on the device, downloaded straight into place. */
int hdp_metrics_f32_planes_i64(const float *x, int64_t n_cells, int64_t T,
                               int64_t stride_cell, int64_t stride_time,
                               const double *thr, int64_t n_thr_cells, int64_t n_doy, int64_t P,
                               const int64_t *doy_map, const int64_t *defs, int64_t D,
                               const int64_t *north, const int64_t *south,
                               const uint8_t *is_south, int64_t Y, int64_t *out);

/* Same call; out [4][P][D][Y][n_cells] int16: the device layout itself, series-minor -- what a collective moves when
 * the cells of a grid are sharded over ranks (2 bytes per value; the shards of a cell axis that is last concatenate
 * row by row). */
int hdp_metrics_f32_layout_i16(const float *x, int64_t n_cells, int64_t T,
                               int64_t stride_cell, int64_t stride_time,
                               const double *thr, int64_t n_thr_cells, int64_t n_doy, int64_t P,
                               const int64_t *doy_map, const int64_t *defs, int64_t D,
                               const int64_t *north, const int64_t *south,
                               const uint8_t *is_south, int64_t Y, int16_t *out);

/* Sharded form of hdp_metrics_f32_planes_i64 (the reference's split over cells is the dask graph of
 * metric.py:444-452).  Needs the library communicator (hdp_comm_init_rank, below).  This rank passes ITS cells of a
 * grid of n_total cells -- the contiguous range [rank * shard, rank * shard + n_loc), shard = ceil(n_total / world) --
 * as n_mem members x n_loc series (member-major; x [n_mem * n_loc][T]) with those cells' thresholds thr
 * [n_loc][n_doy][P].  The int16 result stays on the device, is all-gathered there (ncclAllGather, 2 bytes per value
 * on the wire) and is widened and regrouped once, on the gathered buffer: out [4][P][D][n_mem * n_total][Y] int64,
 * series = member * n_total + grid cell, the complete result on every rank.  *wire_bytes (may be NULL): bytes this
 * rank handed to the collective.  Collective: every rank of the communicator must call it (n_loc may be 0). */
int hdp_metrics_f32_planes_i64_sharded(const float *x, int64_t n_mem, int64_t n_loc, int64_t T,
                                       int64_t stride_cell, int64_t stride_time,
                                       const double *thr, int64_t n_doy, int64_t P,
                                       const int64_t *doy_map, const int64_t *defs, int64_t D,
                                       const int64_t *north, const int64_t *south,
                                       const uint8_t *is_south, int64_t Y, int64_t n_total,
                                       int64_t *out, int64_t *wire_bytes);
/* A rank that fails before the data collective (bad arguments, an n_loc that is not its share, the shard or gathered
 * buffer's allocation, its local pass) still meets the others in a 4-byte status all-gather, and every rank then returns
 * an error without entering the data all-gather: no rank is left blocked in it. */

/* The regrouping half of the sharded call on a host-supplied gathered buffer (no communicator): gathered
 * [world][4 * P * D][Y][n_mem * shard] int16, shard = ceil(n_total / world), zero columns past a rank's cells ->
 * out [4][P][D][n_mem * n_total][Y] int64.  Unit-level: pins the multi-rank layout on one GPU. */
int hdp_metrics_planes_i64_regroup(const int16_t *gathered, int64_t world, int64_t n_mem, int64_t n_total,
                                   int64_t P, int64_t D, int64_t Y, int64_t *out);

/* ---- unit-level mirrors of the njit helpers (for the known-answer tests) ---- */

/* metric.py:11-60: hot [n_series][T] u8 -> ids [n_series][T] int64 */
int hdp_index_heatwaves(const uint8_t *hot, int64_t n_series, int64_t T,
                        int64_t min_duration, int64_t max_break, int64_t max_subs,
                        int64_t *ids);
/* metric.py:63-172: ids [n_series][T] int64, ranges [Y][2] (any order, may overlap)
 * -> out [n_series][4][Y] int64 (HWF,HWN,HWD,trunc(HWA)) and hwa [n_series][Y] f64 */
int hdp_season_metrics(const int64_t *ids, int64_t n_series, int64_t T,
                       const int64_t *ranges, int64_t Y, int64_t *out, double *hwa);
/* metric.py:280-301: measure [n_series][T] f32, thr [n_series][n_doy] f64 -> hot u8 */
int hdp_indicate_hot_days(const float *measure, int64_t n_series, int64_t T,
                          const double *thr, int64_t n_doy, const int64_t *doy_map,
                          uint8_t *hot);

/* ---- heat index pre-step: replaces the ufunc heat_index (hdp/measure.py:61-94) ---------------
 * NWS regression, element-wise: temp [deg F] f32, rel_humid [%] f32 -> heat index [deg F] f32.
 * Arithmetic follows Numba's typing of the reference ufunc float32(float32, float32): float64
 * throughout (its literals are float64) except the float32 product rel_humid*temp of the last
 * polynomial term; the result is rounded to float32 once. */
int hdp_heat_index_f32(const float *temp_f, const float *rel_humid, int64_t n, float *out);
int hdp_heat_index_f32_dev(const float *temp_f_dev, const float *rel_humid_dev, int64_t n,
                           float *out_dev, void *stream);
/* Fused form of what format_standard_measures does around it (measure.py:185-189): Celsius in,
 * Celsius out, with the reference's float32 conversions (t*1.8+32, (hi-32)/1.8) in between. */
int hdp_heat_index_celsius_f32_dev(const float *temp_c_dev, const float *rel_humid_dev, int64_t n,
                                   float *out_c_dev, void *stream);

/* ---- weighted spatial mean: replaces compute_weighted_spatial_mean
 * (hdp/graphics/figure.py:14-15, da.weighted(cos(deg2rad(lat))).mean(dim=["lat","lon"])) --------
 * out[r] = sum_c w[c] v[r][c] / sum_c w[c] over the non-NaN v[r][c] (NaN when no weight is left),
 * float64 accumulation in a fixed order.  The _dev form reads the int16 metrics where
 * hdp_metrics_f32_dev left them: rows = (metric, percentile, definition, season), n = series. */
int hdp_weighted_mean_i16_dev(const int16_t *v_dev, int64_t n_rows, int64_t n, const double *w_dev,
                              double *out_dev, void *stream);
int hdp_weighted_mean_f64(const double *v, int64_t n_rows, int64_t n, const double *w, double *out);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI ----------------------------------------------
 * Grid cells are independent in both passes (reference docs/testing.rst:21; the reference's only parallelism is the
 * dask split over cells, threshold.py:161-169, metric.py:444-452), so ranks own contiguous cell ranges and compute
 * without any exchange; the one collective is the all-gather that reassembles the int16 metrics (and, if wanted,
 * the float64 thresholds) on every rank.  RCCL is linked directly: a caller needs no torch.
 *   rank 0: hdp_comm_unique_id(id) -> ship the HDP_COMM_ID_BYTES bytes to the other ranks by any channel
 *   every rank (after hdp_init): hdp_comm_init_rank(id, rank, world)   (collective)
 *   hdp_allgather_dev(send, bytes, recv, stream): recv [world][bytes] <- every rank's send [bytes]; stream-ordered.
 *   hdp_allgather_direct_dev: the same result by one grouped ncclSend/ncclRecv per peer (full-mesh xGMI). */
#define HDP_COMM_ID_BYTES 128
int hdp_comm_unique_id(void *id_out);
int hdp_comm_init_rank(const void *id, int rank, int world);
int hdp_comm_destroy(void);
int hdp_rccl_version(int *version); /* ncclGetVersion of the librccl this process mapped (no communicator needed) */
int hdp_comm_rank(void);   /* -1 without a communicator */
int hdp_comm_world(void);  /* 0 without a communicator */
int hdp_allgather_dev(const void *send_dev, size_t bytes_per_rank, void *recv_dev, void *stream);
int hdp_allgather_direct_dev(const void *send_dev, size_t bytes_per_rank, void *recv_dev, void *stream);

/* ---- synthetic inputs for bench.py (utils.py:61-78 formula) ---- */
/* x_dev [n_cells][T]: 20 + 2 sin(2 pi (beta + t)/365) - 10|lat|/90 + noise + trend,
 * beta = 90 (south) / 270 (north), noise = u(seed,cell,t) * noise_scale,
 * trend = t * trend_per_day.  lat_dev [n_cells] f32. */
int hdp_generate_series_dev(float *x_dev, int64_t n_cells, int64_t T, int64_t cell_offset,
                            const float *lat_dev, uint64_t seed, float noise_scale,
                            float trend_per_day, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HDP_HIP_H */
