// Day-of-year rolling-window percentile thresholds on gfx950.
//
// Replaces the Numba gufunc compute_percentiles (reference hdp/threshold.py:52-78):
// for every grid cell and every day-of-year row, the quantiles of the B = W*S samples
// whose time indices the window table lists (threshold.py:12-49).
//
// Algorithm (one workgroup per cell, looping over blocks of day-of-year rows):
//   1. load   the samples of the block's day-of-year COLUMNS (S samples each: one per
//             year x member) from HBM into LDS through a host-built (t -> LDS slot) list
//             sorted by t, so the HBM side is read in contiguous runs;
//   2. sort   every column once, descending, in registers (bitonic network over
//             64*EPL elements, one wave per column) -- valid for every window that uses
//             the column, and adjacent windows share 14 of their 15 columns;
//   3. merge  one lane per day-of-year row: W-way merge of the window's sorted columns
//             from the top (and/or from the bottom, whichever end the requested ranks
//             are nearer to), recording the order statistics numba's quantile needs;
//   4. interpolate in float64 with numba's operation order, lower*(1-m) + upper*m,
//             no FMA contraction (this file is compiled with -ffp-contract=off).
//
// This path is LDS/VALU (sort) bound, not MFMA work; its roofline is HBM bandwidth:
// algorithmic bytes per cell = 4*T (read) + 8*n_doy*P (write).
#include "hdp_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <limits>

namespace hdp {

struct ThrDev {
  const int32_t *blk_row0, *blk_nrows, *blk_ncols, *blk_list_off, *blk_list_len;
  const int2 *load_list;
  const uint16_t *cols_local;
  const QuantileParam *qp;
  const int2 *tgt_top, *tgt_bot;
  int n_doy, S, W, P, T, S_pad, Wp, RP, n_blocks, ncols_max;
  int steps_top, steps_bot, nt_top, nt_bot, n;
};

constexpr int kThrThreads = 256;

__device__ __forceinline__ float f32_nan() { return __int_as_float(0x7fc00000); }

// ---- column sort: 64*EPL elements held as v[r] = element (r*64 + lane), descending ----
// One compare-exchange stage: partner = e ^ mask, the element whose `top` bit is clear
// keeps the larger value.  After full unrolling mask/top are compile-time constants.
template <int EPL>
__device__ __forceinline__ void cmp_stage(float (&v)[EPL], int lane, int mask, int top) {
  const int ml = mask & 63;
  const int mr = mask >> 6;
  float nv[EPL];
#pragma unroll
  for (int r = 0; r < EPL; ++r) {
    const float self = v[r];
    float other = v[r ^ mr];
    if (ml) other = __shfl_xor(other, ml, 64);
    const bool keep_max = (top >= 64) ? ((r & (top >> 6)) == 0) : ((lane & top) == 0);
    const bool take = keep_max ? (other > self) : (other < self);
    nv[r] = take ? other : self;
  }
#pragma unroll
  for (int r = 0; r < EPL; ++r) v[r] = nv[r];
}

template <int EPL>
__device__ __forceinline__ void bitonic_desc(float (&v)[EPL], int lane) {
  constexpr int N2 = 64 * EPL;
  // the first step of each merge mirrors (partner = e ^ (k-1)), the rest are
  // half-cleaners (partner = e ^ j); every comparator keeps the larger value at the
  // smaller index, so -inf padding at the tail never moves.
#pragma unroll
  for (int k = 2; k <= N2; k <<= 1) {
    cmp_stage<EPL>(v, lane, k - 1, k >> 1);
#pragma unroll
    for (int j = k >> 2; j >= 1; j >>= 1) cmp_stage<EPL>(v, lane, j, j);
  }
}

// Sort one LDS column (S values at col[0..S)) descending; NaN -> flagged and replaced
// by 0 (any NaN in a window makes every quantile NaN: numba _collect_percentiles).
template <int EPL>
__device__ __forceinline__ void sort_column(float *col, int S, uint32_t *flag_out, int lane) {
  float v[EPL];
  uint32_t n_nan = 0, n_pos = 0, n_neg = 0;
#pragma unroll
  for (int r = 0; r < EPL; ++r) {
    const int e = r * 64 + lane;
    float x = (e < S) ? col[e] : -INFINITY;
    const bool is_nan = (x != x);
    const bool is_pos = (e < S) && (x == INFINITY);
    const bool is_neg = (e < S) && (x == -INFINITY);
    n_nan += __popcll(__ballot(is_nan));
    n_pos += __popcll(__ballot(is_pos));
    n_neg += __popcll(__ballot(is_neg));
    v[r] = is_nan ? 0.0f : x;
  }
  bitonic_desc<EPL>(v, lane);
#pragma unroll
  for (int r = 0; r < EPL; ++r) {
    const int e = r * 64 + lane;
    if (e < S) col[e] = v[r];
  }
  if (lane == 0) *flag_out = (n_nan ? 0x80000000u : 0u) | (n_pos << 15) | n_neg;
}

// ---- W-way merge, one lane per day-of-year row -----------------------------------------
template <bool TOP>
__device__ __forceinline__ void merge_row(const ThrDev &pd, const float *colbuf, float *hbuf,
                                          uint16_t *posb, float *rec, const uint16_t *cl,
                                          int r /* row within block */) {
  const int RP = pd.RP;
  const int steps = TOP ? pd.steps_top : pd.steps_bot;
  const int nt = TOP ? pd.nt_top : pd.nt_bot;
  const int2 *tgt = TOP ? pd.tgt_top : pd.tgt_bot;
  if (steps == 0) return;
  for (int j = 0; j < pd.Wp; ++j) {
    const int idx = ((j >> 2) * RP + r) * 4 + (j & 3);
    float h = f32_nan();
    int pos = 0;
    if (j < pd.W) {
      pos = int(cl[j]) * pd.S_pad + (TOP ? 1 : pd.S);
      h = colbuf[pos];
    }
    hbuf[idx] = h;
    posb[idx] = (uint16_t)pos;
  }
  const float4 *hb4 = reinterpret_cast<const float4 *>(hbuf);
  const int ng = pd.Wp >> 2;
  int k = 0;
  for (int step = 0; step < steps; ++step) {
    float best = TOP ? -INFINITY : INFINITY;
    int bj = 0;
    for (int g = 0; g < ng; ++g) {
      const float4 h = hb4[g * RP + r];
      if (TOP) {
        if (h.x >= best) { best = h.x; bj = 4 * g; }
        if (h.y >= best) { best = h.y; bj = 4 * g + 1; }
        if (h.z >= best) { best = h.z; bj = 4 * g + 2; }
        if (h.w >= best) { best = h.w; bj = 4 * g + 3; }
      } else {
        if (h.x <= best) { best = h.x; bj = 4 * g; }
        if (h.y <= best) { best = h.y; bj = 4 * g + 1; }
        if (h.z <= best) { best = h.z; bj = 4 * g + 2; }
        if (h.w <= best) { best = h.w; bj = 4 * g + 3; }
      }
    }
    while (k < nt && tgt[k].x == step) {  // wave-uniform
      rec[tgt[k].y * RP + r] = best;
      ++k;
    }
    const int idx = ((bj >> 2) * RP + r) * 4 + (bj & 3);
    const int p = int(posb[idx]) + (TOP ? 1 : -1);
    posb[idx] = (uint16_t)p;
    hbuf[idx] = colbuf[p];  // runs onto the NaN sentinel when the column is exhausted
  }
}

// numba _collect_percentiles_inner: value of quantile p from the recorded order statistics
__device__ __forceinline__ double finish_quantile(const QuantileParam &qp, float lo, float hi,
                                                  bool has_nan, int n_pos, int n_neg, int n) {
  if (has_nan) return __longlong_as_double(0x7ff8000000000000LL);
  const double qnan = __longlong_as_double(0x7ff8000000000000LL);
  const bool all_finite = (n_pos + n_neg) == 0;
  if (qp.mode == Q_INTERP) {
    const double a = __dmul_rn((double)lo, qp.w_lo);
    const double b = __dmul_rn((double)hi, qp.w_hi);
    return __dadd_rn(a, b);
  }
  if (qp.mode == Q_MAX) {
    double val = (double)hi;
    if (!all_finite && !isfinite(val)) val = qnan;
    return val;
  }
  double val = (double)lo;  // Q_MIN
  if (!all_finite) {
    const int n_fin = n - (n_pos + n_neg);
    if (n_fin == 0) val = qnan;
    if (n_pos == 1 && n == 2) val = qnan;
    if (n_neg > 1) val = qnan;
    if (n_fin == 1 && n_pos > 1 && n_neg != 1) val = qnan;
  }
  return val;
}

template <int EPL>
__global__ __launch_bounds__(kThrThreads) void thresholds_kernel(ThrDev pd, const float *__restrict__ x,
                                                                 int64_t n_cells,
                                                                 double *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  constexpr int nwaves = kThrThreads / 64;

  // LDS carve (all offsets multiples of 16 bytes)
  size_t off = 0;
  float *colbuf = reinterpret_cast<float *>(smem + off);
  off += (size_t(pd.ncols_max) * pd.S_pad * 4 + 15) & ~size_t(15);
  uint32_t *flags = reinterpret_cast<uint32_t *>(smem + off);
  off += (size_t(pd.ncols_max) * 4 + 15) & ~size_t(15);
  float *hbuf = reinterpret_cast<float *>(smem + off);
  off += size_t(pd.Wp) * pd.RP * 4;
  float *rec = reinterpret_cast<float *>(smem + off);
  off += size_t(2 * pd.P) * pd.RP * 4;
  uint16_t *posb = reinterpret_cast<uint16_t *>(smem + off);

  const int64_t cell = blockIdx.x;
  if (cell >= n_cells) return;
  const float *xc = x + cell * int64_t(pd.T);

  for (int b = 0; b < pd.n_blocks; ++b) {
    const int row0 = pd.blk_row0[b];
    const int nrows = pd.blk_nrows[b];
    const int ncols = pd.blk_ncols[b];
    const int2 *list = pd.load_list + pd.blk_list_off[b];
    const int llen = pd.blk_list_len[b];

    // 1. sentinels + load
    for (int i = tid; i < ncols; i += kThrThreads) {
      colbuf[i * pd.S_pad] = f32_nan();
      colbuf[i * pd.S_pad + pd.S + 1] = f32_nan();
    }
#pragma unroll 4
    for (int i = tid; i < llen; i += kThrThreads) {
      const int2 e = list[i];
      colbuf[e.y] = xc[e.x];
    }
    __syncthreads();

    // 2. sort every column once
    for (int lc = wave; lc < ncols; lc += nwaves)
      sort_column<EPL>(colbuf + lc * pd.S_pad + 1, pd.S, &flags[lc], lane);
    __syncthreads();

    // 3. + 4. merge and interpolate, one lane per row
    if (tid < nrows) {
      const int row = row0 + tid;
      const uint16_t *cl = pd.cols_local + size_t(row) * pd.W;
      merge_row<true>(pd, colbuf, hbuf, posb, rec, cl, tid);
      merge_row<false>(pd, colbuf, hbuf, posb, rec, cl, tid);
      bool has_nan = false;
      int n_pos = 0, n_neg = 0;
      for (int j = 0; j < pd.W; ++j) {
        const uint32_t f = flags[cl[j]];
        has_nan |= (f >> 31) != 0;
        n_pos += (f >> 15) & 0x7fff;
        n_neg += f & 0x7fff;
      }
      double *o = out + (cell * pd.n_doy + row) * int64_t(pd.P);
      for (int p = 0; p < pd.P; ++p) {
        const float lo = rec[(2 * p) * pd.RP + tid];
        const float hi = rec[(2 * p + 1) * pd.RP + tid];
        o[p] = finish_quantile(pd.qp[p], lo, hi, has_nan, n_pos, n_neg, pd.n);
      }
    }
    __syncthreads();
  }
}

// ---- literal-table path: gather B samples, full bitonic sort in LDS, select -----------------
__global__ __launch_bounds__(256) void table_percentiles_kernel(
    const float *__restrict__ x, int64_t n_cells, int64_t T, const int64_t *__restrict__ win,
    int n_doy, int B, int n2, const QuantileParam *__restrict__ qp, const int32_t *__restrict__ klo,
    const int32_t *__restrict__ khi, int P, double *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *buf = reinterpret_cast<float *>(smem);
  __shared__ int s_counts[3];
  const int tid = threadIdx.x;
  const int64_t cell = blockIdx.x / n_doy;
  const int row = blockIdx.x % n_doy;
  const float *xc = x + cell * T;
  if (tid < 3) s_counts[tid] = 0;
  __syncthreads();
  int c_nan = 0, c_pos = 0, c_neg = 0;
  for (int i = tid; i < B; i += blockDim.x) {
    int64_t t = win[int64_t(row) * B + i];
    if (t < 0) t += T;  // NumPy negative indexing (-1 = last time step)
    float v = xc[t];
    if (v != v) { ++c_nan; v = 0.0f; }
    if (v == INFINITY) ++c_pos;
    if (v == -INFINITY) ++c_neg;
    buf[i] = v;
  }
  if (c_nan) atomicAdd(&s_counts[0], c_nan);
  if (c_pos) atomicAdd(&s_counts[1], c_pos);
  if (c_neg) atomicAdd(&s_counts[2], c_neg);
  __syncthreads();
  // ascending bitonic network; every comparator keeps the smaller value at the smaller
  // index, so the virtual +inf padding in [B, n2) never moves and is never touched.
  for (int k = 2; k <= n2; k <<= 1) {
    for (int j = k >> 1; j >= 1; j >>= 1) {
      const bool mirror = (j == (k >> 1));
      const int half = j;
      for (int c = tid; c < (n2 >> 1); c += blockDim.x) {
        const int lo_bits = c & (half - 1);
        const int i = ((c - lo_bits) << 1) | lo_bits;  // index with bit `half` clear
        const int p = mirror ? (i ^ (k - 1)) : (i | half);
        if (p < B) {
          const float a = buf[i], bb = buf[p];
          if (bb < a) { buf[i] = bb; buf[p] = a; }
        }
      }
      __syncthreads();
    }
  }
  if (tid < P) {
    const QuantileParam q = qp[tid];
    float lo = buf[klo[tid]], hi = buf[khi[tid]];
    if (q.mode == Q_MAX) hi = buf[B - 1];
    if (q.mode == Q_MIN) lo = buf[0];
    out[(cell * n_doy + row) * int64_t(P) + tid] =
        finish_quantile(q, lo, hi, s_counts[0] != 0, s_counts[1], s_counts[2], B);
  }
}

// ---- host: numba rank arithmetic ---------------------------------------------------------
int quantile_param(double q, int64_t n, QuantileParam *qp, int64_t *k_lo, int64_t *k_hi) {
  if (!(q >= 0.0 && q <= 1.0)) return HDP_EQUANT;  // also rejects NaN
  const double pct = q * 100.0;
  qp->pad = 0;
  qp->w_lo = 1.0;
  qp->w_hi = 0.0;
  if (pct == 100.0) {
    qp->mode = Q_MAX;
    *k_lo = *k_hi = n - 1;
    return HDP_OK;
  }
  if (pct == 0.0) {
    qp->mode = Q_MIN;
    *k_lo = *k_hi = 0;
    return HDP_OK;
  }
  qp->mode = Q_INTERP;
  volatile double frac = pct / 100.0;           // np.true_divide(percentile, 100.0)
  volatile double prod = double(n - 1) * frac;  // (n - 1) * ...
  volatile double rank = 1.0 + prod;            // 1 + ...
  const double f = std::floor(rank);
  volatile double m = rank - f;
  volatile double one_minus_m = 1.0 - m;
  qp->w_lo = one_minus_m;
  qp->w_hi = m;
  int64_t fi = (int64_t)f;
  *k_lo = fi - 1;
  *k_hi = std::min<int64_t>(fi, n - 1);
  if (*k_lo < 0) *k_lo = 0;
  return HDP_OK;
}

// ---- launchers --------------------------------------------------------------------------------
template <int EPL>
static int launch_thr_epl(const ThrDev &pd, size_t lds, const float *x, int64_t n_cells, double *out,
                          hipStream_t stream) {
  auto kern = thresholds_kernel<EPL>;
  HDP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)n_cells), dim3(kThrThreads), lds, stream, pd, x, n_cells, out);
  HDP_HIP_TRY(hipGetLastError());
  return HDP_OK;
}

int launch_thresholds(const hdp_threshold_plan *plan, const float *x_dev, int64_t n_cells,
                      double *out_dev, hipStream_t stream) {
  if (n_cells == 0) return HDP_OK;
  HDP_REQUIRE(n_cells < (int64_t(1) << 31), HDP_EUNSUP, "n_cells %lld exceeds one launch",
              (long long)n_cells);
  ThrDev pd;
  pd.blk_row0 = plan->blk_row0.as<int32_t>();
  pd.blk_nrows = plan->blk_nrows.as<int32_t>();
  pd.blk_ncols = plan->blk_ncols.as<int32_t>();
  pd.blk_list_off = plan->blk_list_off.as<int32_t>();
  pd.blk_list_len = plan->blk_list_len.as<int32_t>();
  pd.load_list = plan->load_list.as<int2>();
  pd.cols_local = plan->cols_local.as<uint16_t>();
  pd.qp = plan->qparam.as<QuantileParam>();
  pd.tgt_top = plan->tgt_top.as<int2>();
  pd.tgt_bot = plan->tgt_bot.as<int2>();
  pd.n_doy = (int)plan->n_doy;
  pd.S = (int)plan->S;
  pd.W = (int)plan->W;
  pd.P = (int)plan->P;
  pd.T = (int)plan->T;
  pd.S_pad = plan->S_pad;
  pd.Wp = plan->Wp;
  pd.RP = plan->RP;
  pd.n_blocks = plan->n_blocks;
  pd.ncols_max = plan->ncols_max;
  pd.steps_top = plan->steps_top;
  pd.steps_bot = plan->steps_bot;
  pd.nt_top = plan->nt_top;
  pd.nt_bot = plan->nt_bot;
  pd.n = (int)plan->n;
  switch (plan->epl) {
    case 1: return launch_thr_epl<1>(pd, plan->lds_bytes, x_dev, n_cells, out_dev, stream);
    case 2: return launch_thr_epl<2>(pd, plan->lds_bytes, x_dev, n_cells, out_dev, stream);
    case 4: return launch_thr_epl<4>(pd, plan->lds_bytes, x_dev, n_cells, out_dev, stream);
    case 8: return launch_thr_epl<8>(pd, plan->lds_bytes, x_dev, n_cells, out_dev, stream);
    case 16: return launch_thr_epl<16>(pd, plan->lds_bytes, x_dev, n_cells, out_dev, stream);
    case 32: return launch_thr_epl<32>(pd, plan->lds_bytes, x_dev, n_cells, out_dev, stream);
    default: return set_error(HDP_EUNSUP, "samples per day-of-year S=%lld not supported (max 2048)",
                              (long long)plan->S);
  }
}

int launch_table_percentiles(const float *x_dev, int64_t n_cells, int64_t T, const int64_t *win_dev,
                             int64_t n_doy, int64_t B, const QuantileParam *qp_dev,
                             const int32_t *klo_dev, const int32_t *khi_dev, int64_t P,
                             double *out_dev, hipStream_t stream) {
  if (n_cells == 0 || n_doy == 0) return HDP_OK;
  int n2 = 2;
  while (n2 < B) n2 <<= 1;
  const size_t lds = size_t(B) * 4;
  HDP_REQUIRE(lds <= 150 * 1024, HDP_EUNSUP, "window of %lld samples exceeds the LDS sort limit",
              (long long)B);
  HDP_REQUIRE(P <= 256, HDP_EUNSUP, "at most 256 quantiles per call on the table path");
  HDP_REQUIRE(n_cells * n_doy < (int64_t(1) << 31), HDP_EUNSUP, "too many (cell, doy) pairs");
  HDP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(table_percentiles_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(table_percentiles_kernel, dim3((unsigned)(n_cells * n_doy)), dim3(256), lds, stream,
                     x_dev, n_cells, T, win_dev, (int)n_doy, (int)B, n2, qp_dev, klo_dev, khi_dev, (int)P,
                     out_dev);
  HDP_HIP_TRY(hipGetLastError());
  return HDP_OK;
}

}  // namespace hdp

// ---- plan construction (host) -----------------------------------------------------------------
using hdp::set_error;

extern "C" int hdp_threshold_plan_create(const int64_t *time_index, int64_t n_doy, int64_t S,
                                         const int32_t *cols, int64_t W, const double *q, int64_t P,
                                         int64_t T, hdp_threshold_plan **plan_out) {
  HDP_REQUIRE(plan_out, HDP_EINVAL, "plan_out is NULL");
  *plan_out = nullptr;
  HDP_REQUIRE(hdp::device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_REQUIRE(time_index && cols && q, HDP_EINVAL, "NULL table");
  HDP_REQUIRE(n_doy > 0 && S > 0 && W > 0 && P > 0 && T > 0, HDP_EINVAL,
              "n_doy, S, W, P, T must be positive");
  HDP_REQUIRE(T < (int64_t(1) << 31), HDP_EUNSUP, "T too large");
  HDP_REQUIRE(n_doy < 65536, HDP_EUNSUP, "n_doy too large");
  HDP_REQUIRE(S <= 2048, HDP_EUNSUP, "S=%lld samples per day-of-year exceeds 2048", (long long)S);
  HDP_REQUIRE(P <= 4096, HDP_EUNSUP, "too many quantiles");
  for (int64_t i = 0; i < n_doy * S; ++i)
    HDP_REQUIRE(time_index[i] >= -T && time_index[i] < T, HDP_EINVAL,
                "time_index[%lld]=%lld outside [-T, T)", (long long)i, (long long)time_index[i]);
  for (int64_t i = 0; i < n_doy * W; ++i)
    HDP_REQUIRE(cols[i] >= 0 && cols[i] < n_doy, HDP_EINVAL, "cols[%lld]=%d outside [0, n_doy)",
                (long long)i, cols[i]);

  auto *pl = new hdp_threshold_plan();
  pl->n_doy = n_doy; pl->S = S; pl->W = W; pl->P = P; pl->T = T;
  pl->n = W * S;
  pl->Wp = int((W + 3) & ~int64_t(3));
  int spad = int(S) + 2;
  if ((spad & 1) == 0) ++spad;  // odd pitch: a fixed slot across consecutive columns hits all banks
  pl->S_pad = spad;
  int epl = 1;
  while (64 * epl < S) epl <<= 1;
  pl->epl = epl;

  // quantile parameters and merge targets
  std::vector<hdp::QuantileParam> qp(P);
  struct Tgt { int rank, slot; };
  std::vector<Tgt> top, bot;
  const int64_t n = pl->n;
  for (int64_t p = 0; p < P; ++p) {
    int64_t klo, khi;
    int rc = hdp::quantile_param(q[p], n, &qp[p], &klo, &khi);
    if (rc != HDP_OK) {
      delete pl;
      return set_error(HDP_EQUANT, "Quantiles must be in the range [0, 1]");
    }
    if (qp[p].mode == hdp::Q_MAX) {
      top.push_back({0, int(2 * p + 1)});
    } else if (qp[p].mode == hdp::Q_MIN) {
      bot.push_back({0, int(2 * p)});
    } else {
      const int64_t from_top = (n - 1 - klo) + 1;  // merge steps needed coming from the top
      const int64_t from_bot = khi + 1;
      if (from_top <= from_bot) {
        top.push_back({int(n - 1 - klo), int(2 * p)});
        top.push_back({int(n - 1 - khi), int(2 * p + 1)});
      } else {
        bot.push_back({int(klo), int(2 * p)});
        bot.push_back({int(khi), int(2 * p + 1)});
      }
    }
  }
  auto by_rank = [](const Tgt &a, const Tgt &b) { return a.rank < b.rank; };
  std::stable_sort(top.begin(), top.end(), by_rank);
  std::stable_sort(bot.begin(), bot.end(), by_rank);
  pl->nt_top = (int)top.size();
  pl->nt_bot = (int)bot.size();
  pl->steps_top = top.empty() ? 0 : top.back().rank + 1;
  pl->steps_bot = bot.empty() ? 0 : bot.back().rank + 1;

  // choose rows per block against the LDS budget
  auto lds_for = [&](int rows, int ncols) -> size_t {
    const int RP = (rows + 63) & ~63;
    size_t b = (size_t(ncols) * spad * 4 + 15) & ~size_t(15);
    b += (size_t(ncols) * 4 + 15) & ~size_t(15);
    b += size_t(pl->Wp) * RP * 4;       // heads
    b += size_t(2 * P) * RP * 4;        // recorded order statistics
    b += size_t(pl->Wp) * RP * 2;       // positions
    return b;
  };
  auto cols_of_block = [&](int row0, int rows, std::vector<int> &set) {
    set.clear();
    std::vector<char> seen(n_doy, 0);
    for (int r = row0; r < row0 + rows; ++r)
      for (int64_t j = 0; j < W; ++j) {
        const int c = cols[int64_t(r) * W + j];
        if (!seen[c]) { seen[c] = 1; set.push_back(c); }
      }
    std::sort(set.begin(), set.end());
  };
  auto max_lds_for_rows = [&](int rows, int *ncols_max) -> size_t {
    size_t worst = 0;
    int cm = 0;
    std::vector<int> set;
    for (int r0 = 0; r0 < n_doy; r0 += rows) {
      const int nr = (int)std::min<int64_t>(rows, n_doy - r0);
      cols_of_block(r0, nr, set);
      cm = std::max(cm, (int)set.size());
    }
    worst = lds_for(rows, cm);
    *ncols_max = cm;
    return worst;
  };
  const size_t kMaxLds = 160 * 1024 - 1024;
  int rows = 0;
  if (const char *env = getenv("HDP_THR_ROWS")) rows = atoi(env);
  if (rows <= 0) {
    // largest row count whose worst block fits `cap` bytes of LDS
    auto fit = [&](size_t cap) -> int {
      for (int r = (int)std::min<int64_t>(n_doy, hdp::kThrThreads); r >= 1; --r) {
        int cm_;
        if (max_lds_for_rows(r, &cm_) <= cap) return r;
      }
      return 0;
    };
    // prefer 3 workgroups per CU, then 2, then whatever fits one CU
    int r = fit(52 * 1024);
    if (r < std::min<int64_t>(n_doy, 32)) {
      r = fit(78 * 1024);
      if (r < std::min<int64_t>(n_doy, 16)) r = fit(kMaxLds);
    }
    if (r > 0) {
      const int nb = int((n_doy + r - 1) / r);
      rows = int((n_doy + nb - 1) / nb);  // balance the blocks
    }
  }
  rows = (int)std::min<int64_t>(rows, std::min<int64_t>(n_doy, hdp::kThrThreads));
  int cm = 0;
  if (rows <= 0 || max_lds_for_rows(rows, &cm) > kMaxLds) {
    delete pl;
    return set_error(HDP_EUNSUP, "window of %lld x %lld samples does not fit the 160 KiB LDS",
                     (long long)W, (long long)S);
  }
  pl->rows_per_block = rows;
  pl->RP = (rows + 63) & ~63;
  pl->ncols_max = cm;
  pl->lds_bytes = lds_for(rows, cm);
  HDP_REQUIRE(size_t(cm) * spad < 65536, HDP_EUNSUP, "column buffer exceeds 16-bit LDS indexing");
  pl->n_blocks = int((n_doy + rows - 1) / rows);

  // per-block tables
  std::vector<int32_t> row0s, nrows, ncols, loff, llen;
  std::vector<int2> list;
  std::vector<uint16_t> cl(size_t(n_doy) * W);
  std::vector<int> set, local(n_doy);
  for (int b = 0; b < pl->n_blocks; ++b) {
    const int r0 = b * rows;
    const int nr = (int)std::min<int64_t>(rows, n_doy - r0);
    cols_of_block(r0, nr, set);
    std::fill(local.begin(), local.end(), -1);
    for (size_t i = 0; i < set.size(); ++i) local[set[i]] = (int)i;
    for (int r = r0; r < r0 + nr; ++r)
      for (int64_t j = 0; j < W; ++j) cl[size_t(r) * W + j] = (uint16_t)local[cols[int64_t(r) * W + j]];
    const size_t start = list.size();
    for (size_t i = 0; i < set.size(); ++i)
      for (int64_t s = 0; s < S; ++s) {
        int64_t t = time_index[int64_t(set[i]) * S + s];
        if (t < 0) t += T;  // NumPy negative indexing: -1 is the last time step
        list.push_back(make_int2((int)t, int(i) * spad + 1 + (int)s));
      }
    std::stable_sort(list.begin() + start, list.end(), [](const int2 &a, const int2 &b) { return a.x < b.x; });
    row0s.push_back(r0);
    nrows.push_back(nr);
    ncols.push_back((int)set.size());
    loff.push_back((int)start);
    llen.push_back((int)(list.size() - start));
  }
  std::vector<int2> ttop(std::max<size_t>(1, top.size())), tbot(std::max<size_t>(1, bot.size()));
  for (size_t i = 0; i < top.size(); ++i) ttop[i] = make_int2(top[i].rank, top[i].slot);
  for (size_t i = 0; i < bot.size(); ++i) tbot[i] = make_int2(bot[i].rank, bot[i].slot);

  hipError_t e = hipSuccess;
  auto up = [&](hdp::DevBuf &d, const void *src, size_t bytes) {
    if (e == hipSuccess) e = d.upload(src, bytes);
  };
  up(pl->blk_row0, row0s.data(), row0s.size() * 4);
  up(pl->blk_nrows, nrows.data(), nrows.size() * 4);
  up(pl->blk_ncols, ncols.data(), ncols.size() * 4);
  up(pl->blk_list_off, loff.data(), loff.size() * 4);
  up(pl->blk_list_len, llen.data(), llen.size() * 4);
  up(pl->load_list, list.data(), list.size() * sizeof(int2));
  up(pl->cols_local, cl.data(), cl.size() * 2);
  up(pl->qparam, qp.data(), qp.size() * sizeof(hdp::QuantileParam));
  up(pl->tgt_top, ttop.data(), ttop.size() * sizeof(int2));
  up(pl->tgt_bot, tbot.data(), tbot.size() * sizeof(int2));
  if (e != hipSuccess) {
    delete pl;
    return set_error(HDP_EHIP, "uploading threshold plan tables failed: %s", hipGetErrorString(e));
  }
  *plan_out = pl;
  return HDP_OK;
}

extern "C" int hdp_threshold_plan_destroy(hdp_threshold_plan *plan) {
  delete plan;
  return HDP_OK;
}
