// Internal declarations shared by the translation units of libhdp_hip.so.
// gfx950 only: wave64, 160 KiB LDS per CU, 256 CUs in 8 XCDs.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "hdp_hip.h"

namespace hdp {

constexpr int kWave = 64;
constexpr size_t kLdsPerCU = 160 * 1024;

// ---- error plumbing ---------------------------------------------------------
int set_error(int code, const char *fmt, ...);
hipStream_t default_stream();
bool device_ready();
// Value of an HDP_* selector variable, or `dflt` when it is not set.  Selectors are read when a plan is created,
// never at launch; the first time one is found set, a notice naming it goes to stderr.
long long env_option(const char *name, long long dflt);

#define HDP_HIP_TRY(expr)                                                          \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess)                                                          \
      return hdp::set_error(HDP_EHIP, "%s failed: %s (%s:%d)", #expr,              \
                            hipGetErrorString(_e), __FILE__, __LINE__);            \
  } while (0)

#define HDP_REQUIRE(cond, code, ...)                                               \
  do {                                                                             \
    if (!(cond)) return hdp::set_error(code, __VA_ARGS__);                         \
  } while (0)

// Device buffer with RAII (host-side shim only).
struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  // Invariant: bytes == 0 whenever p == nullptr (a failed allocation leaves the buffer empty, so the
  // "bytes >= need" guards of the plan-owned scratch buffers re-allocate instead of launching on a null pointer).
  hipError_t alloc(size_t n) {
    release();
    if (n == 0) return hipSuccess;
    void *q = nullptr;
    const hipError_t e = hipMalloc(&q, n);
    if (e != hipSuccess || q == nullptr) return e == hipSuccess ? hipErrorOutOfMemory : e;
    p = q;
    bytes = n;
    return hipSuccess;
  }
  hipError_t upload(const void *src, size_t n) {
    hipError_t e = alloc(n);
    if (e != hipSuccess || n == 0) return e;
    return hipMemcpy(p, src, n, hipMemcpyHostToDevice);
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  template <class T> T *as() const { return static_cast<T *>(p); }
};

// ---- quantile bookkeeping (numba np.quantile arithmetic, arraymath.py) --------
enum QMode : int32_t { Q_INTERP = 0, Q_MIN = 1, Q_MAX = 2 };

struct QuantileParam {  // one per requested quantile, device-visible POD
  int32_t mode;
  int32_t pad;
  double w_lo;  // (1 - m)
  double w_hi;  // m
};

}  // namespace hdp

// ---- plans (opaque in the C ABI) -------------------------------------------------
struct hdp_threshold_plan {
  int64_t n_doy = 0, S = 0, W = 0, P = 0, T = 0;
  int64_t n = 0;          // samples per window = W * S
  int32_t S_pad = 0;      // column pitch in LDS (floats): sentinel + S + sentinel, odd
  int32_t Wp = 0;         // W rounded up to a multiple of 4
  int32_t epl = 0;        // register elements per lane for the column sort (pow2)
  int32_t rows_per_block = 0, n_blocks = 0, ncols_max = 0;
  int32_t RP = 0;         // rows_per_block rounded up to 64
  int32_t steps_top = 0, steps_bot = 0, nt_top = 0, nt_bot = 0;
  size_t lds_bytes = 0;
  // device tables
  hdp::DevBuf blk_row0, blk_nrows, blk_ncols, blk_list_off, blk_list_len;  // int32 [n_blocks]
  hdp::DevBuf load_list;   // int2 (t, dest) concatenated over blocks
  hdp::DevBuf cols_local;  // uint16 [n_doy][W] local column of each window member
  hdp::DevBuf qparam;      // QuantileParam [P]
  hdp::DevBuf tgt_top, tgt_bot;  // int2 (rank, slot) sorted by rank
  hdp::DevBuf blk_sort_off, sort_slots;  // per block: the LDS column slots it loads and sorts
  bool select_only = false;  // LDS sized without merge heads: only the rank-selection kernel can run this plan
  int32_t n_merge = 0;      // waves that merge (ceil(rows_per_block / 64)); the rest produce
  mutable hdp::DevBuf clk;       // HDP_THR_DEBUG=8 phase clocks (-DHDP_DEBUG_ABLATIONS builds only)
  // time-major input (hdp_thresholds_f32_tm_dev): two series-major staging chunks, a copy stream, fork/join events
  mutable hdp::DevBuf tm_stage;
  mutable hipStream_t tm_stream = nullptr;
  mutable hipEvent_t tm_fork = nullptr, tm_copied[2] = {nullptr, nullptr}, tm_used[2] = {nullptr, nullptr};
  ~hdp_threshold_plan() {
    if (tm_fork) (void)hipEventDestroy(tm_fork);
    for (int i = 0; i < 2; ++i) {
      if (tm_copied[i]) (void)hipEventDestroy(tm_copied[i]);
      if (tm_used[i]) (void)hipEventDestroy(tm_used[i]);
    }
    if (tm_stream) (void)hipStreamDestroy(tm_stream);
  }
  // lane-per-column kernel (S <= 100, W <= 16): one lane sorts one column in registers, no cross-lane stage
  int32_t lane_stride = 0;       // lane kernel: bytes between consecutive samples of every column (0: irregular)
  int32_t lane_n_segs = 0;       // blocked lane kernel, segmented walks: runs of requested ranks, each on its own merging waves
  hdp::DevBuf lane_segs;         // ThrSeg [lane_n_segs]
  int32_t lane_n_merge = 0;      // merging waves of the lane kernel (runs x n_merge in the segmented form)
  bool lane = false;
  int32_t lane_n = 0;            // register slots per column: the kernel's template parameter N >= S
  size_t lane_lds_bytes = 0;     // dynamic LDS of the lane kernel (its head strips are smaller)
  int32_t lane_tier_k = 0;       // samples of a column kept in LDS (== S: all of them); the rest in lane_tail
  int32_t lane_img_pitch = 0;    // LDS words per image column (odd)
  mutable hdp::DevBuf lane_tail; // tiered image: per-workgroup global tail of the sorted columns
  hdp::DevBuf tixl;              // int32 per block [S][64 * tasks]: byte offset of sample s of local column c
  hdp::DevBuf blk_tixl_off;      // int32 [n_blocks] first element of each block in tixl
  // HDP_THR_* selectors (testing and A/B only; every value gives the same results): read ONCE, when the plan is
  // created, and kept here -- a launch never looks at the environment.  -1 / 0 = not set.
  int32_t opt_pipe = -1, opt_select = -1, opt_lane = -1;
  int64_t opt_grid = 0;
};

struct hdp_metrics_plan {
  int64_t T = 0, n_doy = 0, D = 0, Y = 0, P = 0;
  int64_t Ypitch = 0;
  int64_t dmax = 1;           // max over definitions of max(min_duration, 1)
  bool uniform_seasons = false;  // consecutive seasons >= dmax + 64 days apart (fast kernel)
  bool regular_calendar = false; // doy_map[t] == t mod n_doy for every t (noleap / 360-day records that start on day 0)
  bool defs_fit16 = false;       // every min_duration and max_break in [0, 16383], max_subs >= 0: packed 16-bit state machines
  hdp::DevBuf doy_map;   // uint16 [T rounded up to 64]
  hdp::DevBuf defs;      // int32 [D][3]
  hdp::DevBuf defs16;    // int32 [D][3]: the same definitions, those with max_break = 0 first (packed state machines)
  hdp::DevBuf def_perm;  // int32 [D]: position in defs16 -> index in defs
  hdp::DevBuf seasons;   // int2 [2][Y]  (north, south)
  mutable hdp::DevBuf bits_scratch;  // split path: exceedance words of two batches of series (double buffer)
  mutable hdp::DevBuf rows_scratch;  // (percentile, definition)-per-lane kernels: [4][P][D][batch][Ypitch] int16
  mutable hdp::DevBuf tm_stage;      // time-major input: two series-major staging batches
  // season tables that are not increasing and disjoint (user-supplied): served by the per-series unit kernels
  bool ordered_seasons = true;
  hdp::DevBuf ranges64;              // int64 [2][Y][2] north then south, as given (any order, may overlap)
  std::vector<int64_t> defs_host;    // [D][3]
  std::vector<int32_t> defs16_host;  // [D][3] in the order of defs16
  // split path: the streaming exceedance kernel of batch b+1 runs on a second stream beside the
  // VALU-bound state-machine kernel of batch b (created on first use, owned by the plan)
  // HDP_METRICS_* selectors (testing and A/B only; every value gives the same results), read once at plan creation
  int32_t opt_general = 0, opt_fused = 0, opt_cells = 1, opt_packed = 1, opt_overlap = 1, opt_pairs = 1, opt_cw = 0, opt_years = 1, opt_years_lds = 16384, opt_simple = 1;
  int64_t opt_batch = 0;
  mutable hipStream_t aux_stream = nullptr, aux_stream2 = nullptr;
  mutable hipEvent_t ev_fork = nullptr, ev_exceed[2] = {nullptr, nullptr}, ev_state[2] = {nullptr, nullptr};
  ~hdp_metrics_plan() {
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    for (int i = 0; i < 2; ++i) {
      if (ev_exceed[i]) (void)hipEventDestroy(ev_exceed[i]);
      if (ev_state[i]) (void)hipEventDestroy(ev_state[i]);
    }
    if (aux_stream) (void)hipStreamDestroy(aux_stream);
    if (aux_stream2) (void)hipStreamDestroy(aux_stream2);
  }
};

namespace hdp {
// kernel launchers (defined in the .hip files)
// x_pitch > 0: elements between the series of consecutive cells (default: T)
int launch_thresholds(const hdp_threshold_plan *plan, const float *x_dev, int64_t n_cells,
                      double *out_dev, hipStream_t stream, int64_t x_pitch = 0);
// tm_pitch > 0: x_dev is time-major [T][tm_pitch] (element (t, c) at x_dev[t * tm_pitch + c]), else series-major
int launch_metrics(const hdp_metrics_plan *plan, const float *x_dev, const double *thr_dev,
                   int64_t n_thr_cells, const uint8_t *is_south_dev, int64_t n_cells,
                   int16_t *out_dev, hipStream_t stream, int64_t tm_pitch = 0);
bool metrics_year_words(const hdp_metrics_plan *plan);  // exceed_years_kernel + year-aligned scratch words apply
int launch_metrics_any_ranges(const hdp_metrics_plan *plan, const float *x_dev, const double *thr_dev,
                              int64_t n_thr_cells, const uint8_t *is_south_dev, int64_t n_cells, int16_t *out_dev,
                              hipStream_t stream, int64_t tm_pitch);
// x_tm_dev time-major [T][pitch]: chunks of cells are transposed on the plan's copy stream beside the kernel
int launch_thresholds_tm(const hdp_threshold_plan *plan, const float *x_tm_dev, int64_t pitch, int64_t n_cells,
                         double *out_dev, hipStream_t stream);
int reserve_metrics_scratch(const hdp_metrics_plan *plan, int64_t n_cells);
int64_t metrics_batch_cells(const hdp_metrics_plan *plan, int64_t n_cells, int64_t n_thr_cells);
int launch_table_percentiles(const float *x_dev, int64_t n_cells, int64_t T, const int64_t *win_dev,
                             int64_t n_doy, int64_t B, const QuantileParam *qp_dev,
                             const int32_t *klo_dev, const int32_t *khi_dev, int64_t P,
                             double *out_dev, hipStream_t stream);
int launch_index_heatwaves(const uint8_t *hot_dev, int64_t n_series, int64_t T, int64_t min_dur,
                           int64_t max_break, int64_t max_subs, int64_t *ids_dev, hipStream_t stream);
int launch_season_metrics(const int64_t *ids_dev, int64_t n_series, int64_t T,
                          const int64_t *ranges_dev, int64_t Y, int64_t *out_dev, double *hwa_dev,
                          hipStream_t stream);
int launch_indicate_hot_days(const float *x_dev, int64_t n_series, int64_t T, const double *thr_dev,
                             int64_t n_doy, const int64_t *doy_map_dev, uint8_t *hot_dev,
                             hipStream_t stream);
int launch_generate(float *x_dev, int64_t n_cells, int64_t T, int64_t cell_offset, const float *lat_dev,
                    uint64_t seed, float noise_scale, float trend_per_day, hipStream_t stream);
int launch_heat_index(const float *temp_dev, const float *rh_dev, int64_t n, float *out_dev, bool celsius,
                      hipStream_t stream);
int launch_weighted_row_mean_i16(const int16_t *v_dev, int64_t n_rows, int64_t n, const double *w_dev,
                                 double *out_dev, hipStream_t stream);
int launch_weighted_row_mean_f64(const double *v_dev, int64_t n_rows, int64_t n, const double *w_dev,
                                 double *out_dev, hipStream_t stream);
// dst_pitch > 0: elements between consecutive series of the destination (default: T)
int launch_transpose(const float *src_dev, int64_t src_pitch, int64_t T, int64_t n, float *dst_dev,
                     hipStream_t stream, bool beside_state_machines = false, int64_t dst_pitch = 0);
int launch_swap_last2_f64(const double *src_dev, int64_t n, int64_t A, int64_t B, double *dst_dev,
                          hipStream_t stream);
int launch_metrics_repack(const int16_t *dev_layout, int64_t P, int64_t D, int64_t n_cells, int64_t Y,
                          int16_t *ref_layout, hipStream_t stream);

int launch_metrics_planes_i64(const int16_t *dev_layout, int64_t P, int64_t D, int64_t n_cells, int64_t Y,
                              int64_t *planes, hipStream_t stream);

int launch_metrics_planes_i64_gathered(const int16_t *gathered, int64_t MPD, int64_t Y, int64_t world, int64_t shard,
                                       int64_t n_mem, int64_t n_total, int64_t r0, int64_t nr, int64_t *planes,
                                       hipStream_t stream);

// numba rank arithmetic for one quantile over n samples; returns HDP_EQUANT for q
// outside [0,1] (or NaN).  k_lo/k_hi are 0-based ASCENDING order-statistic indices.
int quantile_param(double q, int64_t n, QuantileParam *qp, int64_t *k_lo, int64_t *k_hi);
}  // namespace hdp
