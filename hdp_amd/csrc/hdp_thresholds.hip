// Day-of-year rolling-window percentile thresholds on gfx950.
//
// Replaces the Numba gufunc compute_percentiles (reference hdp/threshold.py:52-78):
// for every grid cell and every day-of-year row, the quantiles of the B = W*S samples
// whose time indices the window table lists (threshold.py:12-49).
//
// Common idea: sort every day-of-year COLUMN (S samples: one per year x member) once,
// descending -- valid for every window that uses the column, and adjacent windows share
// 14 of their 15 columns -- then take the requested order statistics of each row's window
// out of its W sorted columns and interpolate in float64 with numba's operation order,
// lower*(1-m) + upper*m, no FMA contraction (this file is compiled with -ffp-contract=off).
//
// Three kernels, same results bit for bit (tests/test_thresholds_kernels_gpu.py); the plan picks one (thr_variant):
//   thresholds_lane_kernel<N, NG, TIER, ROWS>   S <= 100, W <= 16 -- the headline config (S = 100, W = 15).  One LANE sorts
//             one column in N registers (Batcher merge exchange, nothing crosses lanes); merging waves run a W-way
//             merge per row (one lane per row) out of the LDS image of the previous item.  ROWS = 384 ("whole cell"): all
//             365 columns and all 365 merge chains of a cell in one 12-wave workgroup per CU, the image tiered (top 60
//             samples of a column in LDS, the rest in a per-workgroup global tail); ROWS = 128: blocks of rows, untiered.
//             Walks that start deep (quantile sets away from the tail) are cut into runs that enter the merge at a pivot
//             (merge_row_seg).
//   thresholds_kernel<EPL, false>          any S <= 2048 (and windows wider than 15 columns): one workgroup per cell, load -> LDS,
//             wave sort (lane-major bitonic network), W-way merge per row.
//   thresholds_kernel<EPL, true>           the same with a rank SELECTION per (row, requested
//             rank) instead of the merge, for deep ranks (10-member ensemble: S = 1000): value-pivot rounds, then
//             a popped finish (select_rows).
// plus table_percentiles_kernel, the literal [n_doy, B] table path of the reference's gufunc operands (unit-level parity).
//
// This path is VALU/LDS (sort + merge) work, not MFMA work; its roofline is HBM bandwidth:
// algorithmic bytes per cell = 4*T (read) + 8*n_doy*P (write).
#include "hdp_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <limits>

namespace hdp {

struct ThrSeg {  // one run of requested ranks of a row (device-visible POD, 32 bytes)
  int32_t top;        // 1: descending merge (ranks counted from the largest sample), 0: ascending
  int32_t pivot_pos;  // > 0: enter the merge at the mean of the window columns' samples at this position (1-based, from
                      //      the walk's own end), refined once per lane; 0: enter at the end of the window
  int32_t limit;      // entry is valid while no more than this many samples rank before the pivot (first rank - 1)
  int32_t tgt_off;    // first target of the run in tgt_top / tgt_bot
  int32_t nt;         // targets of the run
  int32_t steps;      // last target's rank + 1
  int32_t pad0, pad1;
};

struct ThrDev {
  const int32_t *blk_row0, *blk_nrows, *blk_ncols, *blk_list_off, *blk_list_len;
  const int2 *load_list;
  const uint16_t *cols_local;
  const QuantileParam *qp;
  const int2 *tgt_top, *tgt_bot;
  const int32_t *blk_sort_off, *sort_slots;  // one-workgroup-per-cell kernel: LDS column slots a block (re)loads and sorts
  const int32_t *tixl, *blk_tixl_off;  // lane-per-column kernel: per block [S][64 * tasks] byte offsets of the samples
  // lane-per-column kernel, tiered image: the top `tier_k` samples of a column live in LDS (column pitch `img_pitch`
  // words), samples tier_k.. in a per-workgroup global tail [parity][sample - tier_k][tail_pitch] (tier_k == S: all in LDS)
  int tier_k, img_pitch, tail_pitch;
  // lane kernel, blocked form, SEGMENTED walks (round 4): the requested ranks of a row are cut into up to kMaxSegs runs; a
  // run is walked by its own merging wave (own head strips), entered at a PIVOT instead of at the end of the window
  int n_segs;            // 0: classic (one lane walks a row's descending and then its ascending merge)
  const struct ThrSeg *segs;
  float *tail;
  int n_merge;                                 // merging waves
  int lane_stride;                             // lane kernel: bytes between consecutive samples of every column (0: irregular, use the table)
  int select;                                  // one-workgroup-per-cell kernel: rank selection instead of the merge
  int n_doy, S, W, P, T, S_pad, Wp, RP, n_blocks, ncols_max;
  int64_t xp;  // elements from one cell's series to the next (T, or the padded pitch of the time-major staging)
  int steps_top, steps_bot, nt_top, nt_bot, n;
  // Timing ablations and instrumentation: compiled in only with -DHDP_DEBUG_ABLATIONS (a release build ignores
  // HDP_THR_DEBUG and carries none of this code).  Bit mask; results are wrong under 1, 2, 4:
  //   1 no merge, 2 no sort, 4 no sample loads, 8 phase clocks of the one-workgroup-per-cell kernel (forces it),
  //   32 phase clocks of the lane kernel (merging wave / first producer), 512 with 32: start-up and step
  //   loop of the merge instead of the producer phases, 64 roles by wave number instead of by SIMD, 1024 busy ticks of every wave of the lane kernel (by role rank),
  //   4096 print the kernel variant chosen.  The clocks cost about 10 % and serialise on global atomics.
  int debug;
  unsigned long long *clk;  // [8] accumulated s_memtime ticks (debug & 8, debug & 32); [8 + rank] busy ticks per wave (debug & 1024)
  long long grid_override;  // HDP_THR_GRID as read at plan creation (0: the occupancy-derived grid)
};

#ifdef HDP_DEBUG_ABLATIONS
#define HDP_DBG(pd, mask) ((pd).debug & (mask))
#else
#define HDP_DBG(pd, mask) 0
#endif

constexpr int kThrThreads = 512;  // 8 waves: all of them load and sort, ceil(rows/64) of them merge

__device__ __forceinline__ float f32_nan() { return __int_as_float(0x7fc00000); }

// Order-preserving float <-> int32 key (an involution on the bit pattern): signed integer
// comparison of keys == float comparison of the NaN-free values, -0 just below +0.  Sorted
// columns are stored as keys so the merge can use integer max3/min3, and the sentinels around
// each column (INT_MAX before, INT_MIN after) lose against every real value including +-inf.
__device__ __forceinline__ int f32_key(float f) {
  const int b = __float_as_int(f);
  return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float key_f32(int k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }
// Sentinels: strictly beyond every real key (+inf is 0x7f800000, -inf is 0x807fffff; NaNs never reach
// the keys).  Not INT_MAX / INT_MIN: packed with a payload and read as a double (pk_make below) these
// two stay finite, normal numbers.
constexpr int kKeyMax = 0x7f900000;
constexpr int kKeyMin = (int)0x80400000;

// ---- 16-lane-row sorter for S <= 128: four columns per wave, no LDS traffic ------------------
// Each 16-lane DPP row owns one column; lane l of the row holds elements e = 8*l + i (i = 0..7).
// Distances 1, 2, 4 are register-to-register, distances 8..64 are lane xor 1, 2, 4 (+ the mirrored
// first step of each merge), all of which DPP serves inside a row: quad_perm, row_half_mirror,
// row_mirror and a row_shl/row_shr pair.  Same comparator network as bitonic_desc_lm below.
template <int CTRL, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_mov(float old, float src) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xf,
                                                     BANK_MASK, false));
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov1(float src) {  // every lane is written: no `old` to preserve
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(src), CTRL, 0xf, 0xf, true));
}
constexpr int kDppXor1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int kDppXor3 = 0x1B;         // quad_perm [3,2,1,0]
constexpr int kDppHalfMirror = 0x141;  // lane ^ 7 inside each half row
constexpr int kDppMirror = 0x140;      // lane ^ 15 inside each row
constexpr int kDppShl4 = 0x104, kDppShr4 = 0x114;

template <int XOR>
__device__ __forceinline__ float row_xor(float v) {
  if constexpr (XOR == 1) return dpp_mov1<kDppXor1>(v);
  else if constexpr (XOR == 2) return dpp_mov1<kDppXor2>(v);
  else if constexpr (XOR == 3) return dpp_mov1<kDppXor3>(v);
  else if constexpr (XOR == 7) return dpp_mov1<kDppHalfMirror>(v);
  else if constexpr (XOR == 15) return dpp_mov1<kDppMirror>(v);
  else {  // XOR == 4: lanes 0-3, 8-11 read lane+4; lanes 4-7, 12-15 read lane-4
    static_assert(XOR == 4, "unsupported row xor");
    float t = dpp_mov<kDppShl4, 0x5>(v, v);
    return dpp_mov<kDppShr4, 0xa>(t, v);
  }
}

// max / min through v_med3_f32 (median with +-inf): one instruction each and, unlike fmaxf/fminf,
// no canonicalising v_max in front (the keys are NaN-free by construction here)
__device__ __forceinline__ float fmax_nn(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, INFINITY); }
__device__ __forceinline__ float fmin_nn(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, -INFINITY); }

__device__ __forceinline__ void ce_reg(float &a, float &b) {  // larger value to the lower index
  const float hi = fmax_nn(a, b), lo = fmin_nn(a, b);
  a = hi;
  b = lo;
}

// ---- wave sorter for S > 128: one column of 64*EPL keys per wave, lane-major ---------------------
// Element e = lane * EPL + r, so the log2(EPL) short distances of every merge are register-to-register
// and only 21 of the 55 stages of a 1024-key sort cross lanes (45 with e = r * 64 + lane); of those,
// lane xor 1, 2, 3, 4, 7, 8, 15 are DPP moves and only xor 16, 31, 63 go through ds_bpermute.
// The network (mirrored first step of every merge, then half-cleaners) keeps the larger key at the smaller index,
// so -inf padding at the tail never moves.
template <int ML>
__device__ __forceinline__ float lane_xor(float v) {
  if constexpr (ML == 1 || ML == 2 || ML == 3 || ML == 4 || ML == 7 || ML == 15) return row_xor<ML>(v);
  else if constexpr (ML == 8) return dpp_mov1<0x128>(v);  // row_ror:8
  else return __shfl_xor(v, ML, 64);
}
template <int EPL, int MASK, int TOP>
__device__ __forceinline__ void lm_stage(float (&v)[EPL], int lane) {
  constexpr int MR = MASK & (EPL - 1), ML = MASK / EPL;
  if constexpr (ML == 0) {
#pragma unroll
    for (int r = 0; r < EPL; ++r)
      if ((r & TOP) == 0) ce_reg(v[r], v[r ^ MR]);
  } else {
    // +inf on lanes that keep the larger key, -inf on the others: med3 is then max or min
    const float lim = (lane & (TOP / EPL)) ? -INFINITY : INFINITY;
    float o[EPL];
#pragma unroll
    for (int r = 0; r < EPL; ++r) o[r] = lane_xor<ML>(v[r ^ MR]);
#pragma unroll
    for (int r = 0; r < EPL; ++r) v[r] = __builtin_amdgcn_fmed3f(v[r], o[r], lim);
  }
}
template <int EPL, int J>
__device__ __forceinline__ void lm_clean(float (&v)[EPL], int lane) {
  if constexpr (J >= 1) {
    lm_stage<EPL, J, J>(v, lane);
    lm_clean<EPL, J / 2>(v, lane);
  }
}
template <int EPL, int K = 2>
__device__ __forceinline__ void bitonic_desc_lm(float (&v)[EPL], int lane) {
  if constexpr (K <= 64 * EPL) {
    lm_stage<EPL, K - 1, K / 2>(v, lane);  // mirrored first step of the merge
    lm_clean<EPL, K / 4>(v, lane);         // half-cleaners
    bitonic_desc_lm<EPL, 2 * K>(v, lane);
  }
}

// cross-lane compare-exchange: partner lane = lane ^ XOR, partner register = MIRROR ? 7 - i : i
// `lim` = +inf on lanes that keep the larger value, -inf on lanes that keep the smaller one:
// med3(self, other, lim) is then max or min in ONE instruction
template <int XOR, bool MIRROR>
__device__ __forceinline__ void ce_cross(float (&v)[8], float lim) {
  float o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = row_xor<XOR>(v[MIRROR ? 7 - i : i]);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_fmed3f(v[i], o[i], lim);
}

__device__ __forceinline__ void reg_tail(float (&v)[8]) {  // half-cleaners at distances 4, 2, 1
  ce_reg(v[0], v[4]); ce_reg(v[1], v[5]); ce_reg(v[2], v[6]); ce_reg(v[3], v[7]);
  ce_reg(v[0], v[2]); ce_reg(v[1], v[3]); ce_reg(v[4], v[6]); ce_reg(v[5], v[7]);
  ce_reg(v[0], v[1]); ce_reg(v[2], v[3]); ce_reg(v[4], v[5]); ce_reg(v[6], v[7]);
}

// LPC = lanes per column (1, 2, 4, 8, 16): a column of up to 8*LPC keys is spread over LPC
// neighbouring lanes (8 keys each), so a wave sorts 64/LPC columns at once and short columns
// (few years of data) do not pay for the full 128-key network.
template <int LPC>
__device__ __forceinline__ void sort_group_desc(float (&v)[8], int l /* lane within the column group */) {
  // k = 2, 4, 8: inside the lane
  ce_reg(v[0], v[1]); ce_reg(v[2], v[3]); ce_reg(v[4], v[5]); ce_reg(v[6], v[7]);
  ce_reg(v[0], v[3]); ce_reg(v[1], v[2]); ce_reg(v[4], v[7]); ce_reg(v[5], v[6]);
  ce_reg(v[0], v[1]); ce_reg(v[2], v[3]); ce_reg(v[4], v[5]); ce_reg(v[6], v[7]);
  ce_reg(v[0], v[7]); ce_reg(v[1], v[6]); ce_reg(v[2], v[5]); ce_reg(v[3], v[4]);
  ce_reg(v[0], v[2]); ce_reg(v[1], v[3]); ce_reg(v[4], v[6]); ce_reg(v[5], v[7]);
  ce_reg(v[0], v[1]); ce_reg(v[2], v[3]); ce_reg(v[4], v[5]); ce_reg(v[6], v[7]);
  const float b0 = (l & 1) ? -INFINITY : INFINITY, b1 = (l & 2) ? -INFINITY : INFINITY;
  const float b2 = (l & 4) ? -INFINITY : INFINITY, b3 = (l & 8) ? -INFINITY : INFINITY;
  if constexpr (LPC >= 2) {  // k = 16
    ce_cross<1, true>(v, b0);
    reg_tail(v);
  }
  if constexpr (LPC >= 4) {  // k = 32
    ce_cross<3, true>(v, b1);
    ce_cross<1, false>(v, b0);
    reg_tail(v);
  }
  if constexpr (LPC >= 8) {  // k = 64
    ce_cross<7, true>(v, b2);
    ce_cross<2, false>(v, b1);
    ce_cross<1, false>(v, b0);
    reg_tail(v);
  }
  if constexpr (LPC >= 16) {  // k = 128
    ce_cross<15, true>(v, b3);
    ce_cross<4, false>(v, b2);
    ce_cross<2, false>(v, b1);
    ce_cross<1, false>(v, b0);
    reg_tail(v);
  }
}

// Sort 64/LPC LDS columns (lc0 + lane / LPC) at once; S <= 8 * LPC.
template <int LPC>
__device__ __forceinline__ void sort_columns_rows(float *colbuf, int S_pad, int S, int lc0, int ncols,
                                                  uint32_t *flags, int lane) {
  const int grp = lane / LPC, l = lane % LPC;
  const int lc = lc0 + grp;
  const bool active = lc < ncols;
  float *col = colbuf + (active ? lc : lc0) * S_pad + 1;
  float v[8];
  uint32_t cnt = 0;  // nan << 20 | +inf << 10 | -inf
  bool special = false;  // exponent all ones: NaN or infinity (rare; counted only when present)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int e = l * 8 + i;
    const bool real = active && e < S;
    v[i] = real ? col[e] : -INFINITY;
    special |= real && ((__float_as_uint(v[i]) & 0x7f800000u) == 0x7f800000u);
  }
  if (__ballot(special) != 0) {  // wave-uniform
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool real = active && (l * 8 + i) < S;
      float x = v[i];
      if (x != x) { cnt += 1u << 20; x = 0.0f; }
      if (real && x == INFINITY) cnt += 1u << 10;
      if (real && x == -INFINITY) cnt += 1u;
      v[i] = x;
    }
  }
  sort_group_desc<LPC>(v, l);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int e = l * 8 + i;
    if (active && e < S) col[e] = __int_as_float(f32_key(v[i]));
  }
  // group-wide sum of the packed counters (each field < 1024)
  if constexpr (LPC >= 2) cnt += __builtin_amdgcn_update_dpp(0, (int)cnt, kDppXor1, 0xf, 0xf, false);
  if constexpr (LPC >= 4) cnt += __builtin_amdgcn_update_dpp(0, (int)cnt, kDppXor2, 0xf, 0xf, false);
  if constexpr (LPC >= 8) cnt += __builtin_amdgcn_update_dpp(0, (int)cnt, kDppHalfMirror, 0xf, 0xf, false);
  if constexpr (LPC >= 16) cnt += __builtin_amdgcn_update_dpp(0, (int)cnt, kDppMirror, 0xf, 0xf, false);
  if (active && l == 0)
    flags[lc] = ((cnt >> 20) ? 0x80000000u : 0u) | (((cnt >> 10) & 0x3ffu) << 15) | (cnt & 0x3ffu);
}

template <int LPC>
__device__ __forceinline__ void sort_block_rows(float *colbuf, int S_pad, int S, int ncols, uint32_t *flags,
                                                int wave, int nwaves, int lane) {
  constexpr int kCols = 64 / LPC;
  for (int lc0 = wave * kCols; lc0 < ncols; lc0 += nwaves * kCols)
    sort_columns_rows<LPC>(colbuf, S_pad, S, lc0, ncols, flags, lane);
}

// Sort one LDS column (S values at col[0..S)) descending; NaN -> flagged and replaced
// by 0 (any NaN in a window makes every quantile NaN: numba _collect_percentiles).
template <int EPL>
__device__ __forceinline__ void sort_column(float *col, int S, uint32_t *flag_out, int lane) {
  float v[EPL];
  uint32_t n_nan = 0, n_pos = 0, n_neg = 0;
  bool special = false;  // exponent all ones: NaN or infinity (rare; counted only when present)
#pragma unroll
  for (int r = 0; r < EPL; ++r) {
    // a sorting network does not care which slot an input starts in: read with consecutive lanes on consecutive
    // words (conflict-free) although the sorted output is lane-major
    const int e = r * 64 + lane;
    const float x = col[min(e, S - 1)];
    special |= (__float_as_uint(x) & 0x7f800000u) == 0x7f800000u;
    v[r] = (e < S) ? x : -INFINITY;
  }
  if (__ballot(special) != 0) {  // wave-uniform
#pragma unroll
    for (int r = 0; r < EPL; ++r) {
      const int e = r * 64 + lane;
      const float x = v[r];
      const bool is_nan = (x != x);
      n_nan += __popcll(__ballot(is_nan));
      n_pos += __popcll(__ballot((e < S) && (x == INFINITY)));
      n_neg += __popcll(__ballot((e < S) && (x == -INFINITY)));
      v[r] = is_nan ? 0.0f : x;
    }
  }
  bitonic_desc_lm<EPL>(v, lane);
#pragma unroll
  for (int r = 0; r < EPL; ++r) {
    const int e = lane * EPL + r;
    if (e < S) col[e] = __int_as_float(f32_key(v[r]));
  }
  if (lane == 0) *flag_out = (n_nan ? 0x80000000u : 0u) | (n_pos << 15) | n_neg;
}

// ---- W-way merge, one lane per day-of-year row -----------------------------------------
// Heads and their LDS positions live in a lane-private LDS strip laid out [group of 4][row][4], so
// one step is: all head/position reads issued back to back (NG x ds_read_b128 + NG x ds_read_b64),
// a compare tree that carries the winner's position along, one dependent read of the winner's
// next element, two small writes.  NG = ceil(W/4) is a template parameter so the reads are not
// serialised by a runtime loop (NG = 0: generic fallback).
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

template <bool TOP>
__device__ __forceinline__ int kbest(int a, int b) { return TOP ? max(a, b) : min(a, b); }

// Winner of one group of four heads (keys) with its payload.
template <bool TOP>
__device__ __forceinline__ void group_winner(const int4 &h, const uint4 &q, int &m, uint32_t &pay) {
  m = kbest<TOP>(kbest<TOP>(h.x, h.y), kbest<TOP>(h.z, h.w));
  pay = q.x;
  pay = (h.y == m) ? q.y : pay;
  pay = (h.z == m) ? q.z : pay;
  pay = (h.w == m) ? q.w : pay;
}

// what to do when the merge reaches a requested rank (host-built list, sorted by rank)
enum EmitKind : int { E_TOP_PAIR = 0, E_BOT_PAIR = 1, E_SAME = 2, E_MAX = 3, E_MIN = 4 };

struct RowFlags {  // NaN / infinity census of the row's window (numba's special cases); n_pos < 0: a NaN is present
  int n_pos, n_neg;
};

// Each requested quantile needs two ADJACENT order statistics, so it is finished the moment the
// second one appears: `best` is the current merge output, `prev` the one before it.
// Read-only plan tables indexed by a wave-uniform value: loaded through the constant address space so
// the compiler emits scalar loads (s_load, scalar cache) instead of a vector load with a full
// vmcnt(0) round trip in the middle of the merge.
__device__ __forceinline__ long long ldk64(const void *p) {
  typedef const __attribute__((address_space(4))) long long *kptr;
  return *reinterpret_cast<kptr>(reinterpret_cast<uintptr_t>(p));
}
__device__ __forceinline__ int2 ldk(const int2 *p) {
  const long long v = ldk64(p);
  return make_int2(int(v), int(v >> 32));
}
__device__ __forceinline__ QuantileParam ldk(const QuantileParam *p) {
  static_assert(sizeof(QuantileParam) == 24, "QuantileParam layout");
  QuantileParam q;
  q.mode = int(ldk64(p));
  q.pad = 0;
  q.w_lo = __longlong_as_double(ldk64(reinterpret_cast<const char *>(p) + 8));
  q.w_hi = __longlong_as_double(ldk64(reinterpret_cast<const char *>(p) + 16));
  return q;
}

// RAW: best / prev are float bit patterns (the lane-per-column kernel's image), not order-preserving keys
template <bool TOP, bool RAW = false>
__device__ __forceinline__ void emit_targets(const ThrDev &pd, const int2 *tgt, int nt, int &k, int &next_rank,
                                             int step, int best, int prev, const RowFlags &rf, bool store,
                                             double *orow) {
  int kk = __builtin_amdgcn_readfirstlane(k);
  int2 t = ldk(&tgt[kk]);
  do {
    const int p = t.y & 0xffff, kind = t.y >> 16;
    const float fb = RAW ? __int_as_float(best) : key_f32(best), fp = RAW ? __int_as_float(prev) : key_f32(prev);
    float lo = fb, hi = fb;
    if (kind == E_TOP_PAIR) hi = fp;
    if (kind == E_BOT_PAIR) lo = fp;
    const QuantileParam qp = ldk(&pd.qp[p]);
    // percentile-major output: consecutive lanes (rows) write consecutive doubles
    if (store) orow[size_t(p) * pd.n_doy] = finish_quantile(qp, lo, hi, rf.n_pos < 0, rf.n_pos, rf.n_neg, pd.n);
    ++kk;
    t = kk < nt ? ldk(&tgt[kk]) : make_int2(-1, 0);
  } while (t.x == step);
  k = kk;
  next_rank = t.x;
}

// ---- W-way merge, one lane per day-of-year row ------------------------------------------------
// A head is ONE 64-bit word: (order-preserving key, payload = LDS position << 2 | group), laid out so
// that, read as a double, it is a positive normal number whose order is the (key, payload) order:
//   hi = 0 | key ^ 0x80000000 (31 top bits),  lo = key bit 0 | payload (31 bits).
// A compare-exchange of (key, payload) pairs is then v_max_f64 + v_min_f64 -- two instructions instead
// of max, min, compare and two selects -- and moves the bits untouched (no arithmetic, no rounding).
// Heads live in a lane-private LDS strip [group of 4][row][4], sorted best-first inside a group; the
// group tops are cached, also sorted, in registers.  A step
//   1. pops the best cached top m[0],
//   2. issues, together, the read of that column's next key and of the winner's group,
//   3. bubbles the new head into the group (three compare-exchanges), writes the group back,
//   4. bubbles the group's new top into the cached tops (three more).
// One LDS round trip and about 25 issue slots per step.  NG = ceil(W/4) is a template parameter
// (NG = 0: generic rescan loop on int keys for very wide windows, payload = position << 8 | slot).
__device__ __forceinline__ double pk_max(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));  // not fmax(): no canonicalising pre-pass
  return r;
}
__device__ __forceinline__ double pk_min(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double pk_make(int key, uint32_t pay) {
  const uint32_t hi = (uint32_t(key) >> 1) ^ 0x40000000u;
  const uint32_t lo = (uint32_t(key) << 31) | pay;
  return __hiloint2double(int(hi), int(lo));
}
__device__ __forceinline__ int pk_key(double d) {
  const uint32_t hi = uint32_t(__double2hiint(d)), lo = uint32_t(__double2loint(d));
  return int(((hi << 1) | (lo >> 31)) ^ 0x80000000u);
}
// compare-exchange: the better head ends up in a
template <bool TOP>
__device__ __forceinline__ void ce_pk(double &a, double &b) {
  const double hi = pk_max(a, b), lo = pk_min(a, b);
  a = TOP ? hi : lo;
  b = TOP ? lo : hi;
}
// k[1..N) is sorted best-first; k[0] is new: bubble it down to its place
template <bool TOP, int N>
__device__ __forceinline__ void insert_front(double (&k)[N]) {
#pragma unroll
  for (int i = 0; i + 1 < N; ++i) ce_pk<TOP>(k[i], k[i + 1]);
}
template <bool TOP, int N>
__device__ __forceinline__ void sort_best_first(double (&k)[N]) {
  if constexpr (N == 4) {
    ce_pk<TOP>(k[0], k[1]); ce_pk<TOP>(k[2], k[3]); ce_pk<TOP>(k[0], k[2]); ce_pk<TOP>(k[1], k[3]);
    ce_pk<TOP>(k[1], k[2]);
  } else {
#pragma unroll
    for (int a = 0; a < N; ++a)
#pragma unroll
      for (int i = 0; i + 1 < N - a; ++i) ce_pk<TOP>(k[i], k[i + 1]);
  }
}

template <bool TOP, int NG>
__device__ __forceinline__ void merge_row(const ThrDev &pd, const float *colbuf_f, float *hbuf_f,
                                          uint32_t *posb, const uint16_t *cl,
                                          int r /* row within block */, const RowFlags &rf, bool store,
                                          double *orow) {
  const int *colbuf = reinterpret_cast<const int *>(colbuf_f);
  int *hbuf = reinterpret_cast<int *>(hbuf_f);
  const int RP = pd.RP;
  const int steps = TOP ? pd.steps_top : pd.steps_bot;
  const int nt = TOP ? pd.nt_top : pd.nt_bot;
  const int2 *tgt = TOP ? pd.tgt_top : pd.tgt_bot;
  if (steps == 0) return;
  const int worst = TOP ? kKeyMin : kKeyMax;
  if constexpr (NG > 0) {
    // strip halves: heads 0,1 of every (group, row) in the first, heads 2,3 in the second, so that a
    // wave's 16-byte accesses are stride-16 in both (no bank conflicts)
    double2 *sa = reinterpret_cast<double2 *>(hbuf);
    double2 *sb = reinterpret_cast<double2 *>(posb);
    double m[NG];
    // the row's local column ids: one batch of independent loads (read again by the other direction
    // rather than kept live across this merge)
    int clr[4 * NG];
#pragma unroll
    for (int j = 0; j < 4 * NG; ++j) clr[j] = (j < pd.W) ? int(cl[j]) : 0;
    unsigned long long tm0 = 0, tm1 = 0;
    if (HDP_DBG(pd, 512)) tm0 = __builtin_readcyclecounter();
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      double hd[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = 4 * g + i;
        const int pos = (j < pd.W) ? clr[j] * pd.S_pad + (TOP ? 1 : pd.S) : 0;
        hd[i] = pk_make((j < pd.W) ? colbuf[pos] : worst, (uint32_t(pos) << 2) | uint32_t(g));
      }
      sort_best_first<TOP, 4>(hd);
      sa[g * RP + r] = make_double2(hd[0], hd[1]);
      sb[g * RP + r] = make_double2(hd[2], hd[3]);
      m[g] = hd[0];
    }
    sort_best_first<TOP, NG>(m);
    int k = 0;
    int next_rank = nt > 0 ? ldk(&tgt[0]).x : -1;
    double prev = pk_make(worst, 0);
    if (HDP_DBG(pd, 512)) {
      asm volatile("" ::"v"(m[0]));
      tm1 = __builtin_readcyclecounter();
    }
    // Steps run in emission-free stretches: inside a stretch only LDS operations are in flight, so the
    // wait counters stay partial (a loop that can also reach the emission code, with its scalar loads,
    // makes the compiler drain the previous step's strip writes before every step).
    auto do_step = [&]() {
      prev = m[0];
      const uint32_t lo = uint32_t(__double2loint(m[0]));
      const uint32_t g = lo & 3u;
      const int gidx = __mul24(int(g), RP) + r;
      // payload = position << 2 | group: position * 4 is the byte offset into the key image
      const int nk = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(colbuf) +
                                                     (lo & 0x7ffffffcu) + (TOP ? 4 : -4));
      const double2 ha = sa[gidx];
      const double2 hb = sb[gidx];
      __builtin_amdgcn_sched_barrier(0);  // all three reads are in flight before anything waits on one
      double hd[4] = {pk_make(nk, (lo & 0x7fffffffu) + (TOP ? 4u : -4u)), ha.y, hb.x, hb.y};
      insert_front<TOP, 4>(hd);
      sa[gidx] = make_double2(hd[0], hd[1]);
      sb[gidx] = make_double2(hd[2], hd[3]);
      m[0] = hd[0];
      insert_front<TOP, NG>(m);
    };
    int step = 0;
    {
      while (true) {
        const int stop = (next_rank >= 0 && next_rank < steps) ? next_rank : steps;  // wave-uniform
        for (; step < stop; ++step) do_step();
        if (step >= steps) break;
        // step == next_rank: m[0] is order statistic `step`, prev the one before it
        emit_targets<TOP>(pd, tgt, nt, k, next_rank, step, pk_key(m[0]), pk_key(prev), rf, store, orow);
        next_rank = __builtin_amdgcn_readfirstlane(next_rank);
      }
    }
    if (HDP_DBG(pd, 512) && r == 0) {
      asm volatile("" ::"v"(m[0]));
      const unsigned long long tm2 = __builtin_readcyclecounter();
      atomicAdd(&pd.clk[4], tm1 - tm0);
      atomicAdd(&pd.clk[5], tm2 - tm1);
    }
    return;
  } else {
    for (int j = 0; j < pd.Wp; ++j) {
      const int idx = ((j >> 2) * RP + r) * 4 + (j & 3);
      int h = worst;
      int pos = 0;
      if (j < pd.W) {
        pos = int(cl[j]) * pd.S_pad + (TOP ? 1 : pd.S);
        h = colbuf[pos];
      }
      hbuf[idx] = h;
      posb[idx] = (uint32_t(pos) << 8) | uint32_t(j);
    }
    int4 *hb4 = reinterpret_cast<int4 *>(hbuf);
    uint4 *pb4 = reinterpret_cast<uint4 *>(posb);
    int k = 0;
    int next_rank = nt > 0 ? ldk(&tgt[0]).x : -1;
    int prev = worst;
    const int ng = pd.Wp >> 2;
    for (int step = 0; step < steps; ++step) {
      int best = worst;
      uint32_t bp = 0;
      for (int g = 0; g < ng; ++g) {
        int m;
        uint32_t pay;
        group_winner<TOP>(hb4[g * RP + r], pb4[g * RP + r], m, pay);
        const bool t = TOP ? (m >= best) : (m <= best);
        best = t ? m : best;
        bp = t ? pay : bp;
      }
      if (step == next_rank) emit_targets<TOP>(pd, tgt, nt, k, next_rank, step, best, prev, rf, store, orow);
      prev = best;
      const int bj = int(bp & 0xffu);
      const int idx = ((bj >> 2) * RP + r) * 4 + (bj & 3);
      const int p = int(bp >> 8) + (TOP ? 1 : -1);
      posb[idx] = (uint32_t(p) << 8) | uint32_t(bj);
      hbuf[idx] = colbuf[p];
    }
  }
}


template <int NG>
__device__ __forceinline__ void merge_both(const ThrDev &pd, const float *colbuf, float *hbuf, uint32_t *posb,
                                           const uint32_t *flags, const uint16_t *cl, int r, bool store,
                                           double *orow) {
  RowFlags rf{0, 0};
  uint32_t nan_or = 0;
  if constexpr (NG > 0) {
#pragma unroll
    for (int j = 0; j < 4 * NG; ++j) {
      const uint32_t f = (j < pd.W) ? flags[cl[j]] : 0u;
      nan_or |= f;
      rf.n_pos += (f >> 15) & 0x7fff;
      rf.n_neg += f & 0x7fff;
    }
  } else {
    for (int j = 0; j < pd.W; ++j) {
      const uint32_t f = flags[cl[j]];
      nan_or |= f;
      rf.n_pos += (f >> 15) & 0x7fff;
      rf.n_neg += f & 0x7fff;
    }
  }
  if (nan_or >> 31) rf.n_pos = -1;
  merge_row<true, NG>(pd, colbuf, hbuf, posb, cl, r, rf, store, orow);
  merge_row<false, NG>(pd, colbuf, hbuf, posb, cl, r, rf, store, orow);
}


// ---- lean heads for the lane-per-column kernel ----------------------------------------------------------------
// One wave issues about one instruction per 4.6 cycles whatever the instruction (tools/ubench), so a merging
// wave's step time is its instruction count plus whatever LDS latency it cannot cover -- the SIMD itself is mostly
// idle under it.  This form of the merge therefore spends as few instructions per step as the data layout allows:
//  * the image holds the samples' RAW float bits (NaNs scrubbed): used as the high word of a double they order like
//    the floats themselves (sign-magnitude both; the exponent field of the double is all ones only for NaN
//    patterns, which never occur, and the sentinels 0x7fe00000 / 0xffe00000 lie beyond +-inf and are finite
//    doubles), so a head is (float bits, payload) with NO key transformation -- pk_make is gone;
//  * payload = byte offset of the key in the image | group: next key at (payload & ~3) + 4, new payload = payload + 4;
//    among equal keys the payload orders positive values one way and negative values the other -- either is a
//    total order, which is all the merge needs (equal keys are equal values);
//  * a group's strip keeps only its 2nd..4th head (the top is the cached top being popped): 24 bytes per (group,
//    row) in two arrays with compile-time pitch (128 rows), read and written as b128 + b64;
//  * software-pipelined like merge_row_pl: the next step's reads are issued as soon as the next winner is known.
constexpr int kLeanRows = 128;                       // strip pitch of the blocked form (rows per block <= 128)
constexpr int kWholeRows = 384;                      // strip pitch of the whole-cell form (all day-of-year rows in one workgroup)
constexpr int kWholeThreads = 768;                   // 6 merging + 6 producing waves
#ifndef HDP_TIERK
#define HDP_TIERK 60
#endif
constexpr int kTierK = HDP_TIERK;                           // tiered image: samples of a column kept in LDS (plans with S > 64)
constexpr uint32_t kRawMax = 0x7fe00000u;            // above +inf (0x7f800000), finite as the high word of a double
constexpr uint32_t kRawMin = 0xffe00000u;            // below -inf (0xff800000)
constexpr int kHG = 3;        // heads per group of the lean merge (round 4: groups of three, NG = 3 or 5 groups; rounds 2-3: four of four)
constexpr int kListPitch = 16;  // uint16 entries per row of the window column lists in LDS (3 * NG <= 15 used)
constexpr uint32_t kPayGroupShift = 18;  // payload: LDS byte address in bits 0..17, group in 18..20, tail position 21..27, tail flag 30
template <int NG, int ROWS>
constexpr size_t lean_strip_bytes() { return size_t(NG) * ROWS * 16; }

// The emission tables of one merge direction held ACROSS THE LANES of a few registers: lane k has target k's rank, its
// (quantile | kind << 16) word and that quantile's parameters.  An emission then reads them with v_readlane (a
// wave-uniform k) instead of a chain of scalar loads whose latency the merging wave sat out ten times per row (0.8 of
// 7.9 ms per 65 536 cells at C3, measured by ablation).  Loaded once per kernel; plans with more than 64 targets per
// direction keep the scalar-load form.
struct TgtLanes {
  int rank, slot, mode;
  double w_lo, w_hi;
  bool ok;
};
template <bool TOP>
__device__ __forceinline__ TgtLanes load_tgt_lanes(const ThrDev &pd, int lane) {
  const int nt = TOP ? pd.nt_top : pd.nt_bot;
  const int2 *tgt = TOP ? pd.tgt_top : pd.tgt_bot;
  TgtLanes t{-1, 0, 0, 0.0, 0.0, nt <= 64};
  if (t.ok && lane < nt) {
    const int2 r = tgt[lane];
    const QuantileParam qp = pd.qp[r.y & 0xffff];
    t.rank = r.x;
    t.slot = r.y;
    t.mode = qp.mode;
    t.w_lo = qp.w_lo;
    t.w_hi = qp.w_hi;
  }
  return t;
}
// the same for one run of a segmented plan: targets tgt[0 .. nt)
__device__ __forceinline__ TgtLanes load_tgt_lanes_seg(const ThrDev &pd, const int2 *tgt, int nt, int lane) {
  TgtLanes t{-1, 0, 0, 0.0, 0.0, nt <= 64};
  if (t.ok && lane < nt) {
    const int2 r = tgt[lane];
    const QuantileParam qp = pd.qp[r.y & 0xffff];
    t.rank = r.x;
    t.slot = r.y;
    t.mode = qp.mode;
    t.w_lo = qp.w_lo;
    t.w_hi = qp.w_hi;
  }
  return t;
}
__device__ __forceinline__ double readlane_f64(double v, int k) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), k), __builtin_amdgcn_readlane(__double2loint(v), k));
}
template <bool TOP>
__device__ __forceinline__ void emit_targets_lanes(const ThrDev &pd, const TgtLanes &tl, int nt, int &k, int &next_rank,
                                                   int step, int best, int prev, const RowFlags &rf, double *orow,
                                                   bool store = true) {
  int kk = __builtin_amdgcn_readfirstlane(k);
  int rank_k = step;
  do {
    const int slot = __builtin_amdgcn_readlane(tl.slot, kk);
    const int p = slot & 0xffff, kind = slot >> 16;
    const float fb = __int_as_float(best), fp = __int_as_float(prev);
    float lo = fb, hi = fb;
    if (kind == E_TOP_PAIR) hi = fp;
    if (kind == E_BOT_PAIR) lo = fp;
    QuantileParam qp;
    qp.mode = __builtin_amdgcn_readlane(tl.mode, kk);
    qp.pad = 0;
    qp.w_lo = readlane_f64(tl.w_lo, kk);
    qp.w_hi = readlane_f64(tl.w_hi, kk);
    if (store) orow[size_t(p) * pd.n_doy] = finish_quantile(qp, lo, hi, rf.n_pos < 0, rf.n_pos, rf.n_neg, pd.n);
    ++kk;
    rank_k = kk < nt ? __builtin_amdgcn_readlane(tl.rank, min(kk, 63)) : -1;
  } while (rank_k == step);
  k = kk;
  next_rank = rank_k;
}

// 32-bit read at an absolute LDS byte address
__device__ __forceinline__ uint32_t lds_u32(uint32_t addr) {
  return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>(uintptr_t(addr));
}

template <bool TOP, int NG, bool TIER, int ROWS>
__device__ __forceinline__ void merge_row_lean(const ThrDev &pd, const unsigned char *image, unsigned char *strips,
                                               const float *tail_cur, const uint16_t *cl, int r, const RowFlags &rf,
                                               double *orow, const TgtLanes &tl, int prio_phase = -1) {
  static_assert(NG >= 1 && NG <= 7, "group id is three payload bits");
  const int steps = TOP ? pd.steps_top : pd.steps_bot;
  const int nt = TOP ? pd.nt_top : pd.nt_bot;
  const int2 *tgt = TOP ? pd.tgt_top : pd.tgt_bot;
  if (steps == 0) return;
  auto better = [](double a, double b) { return TOP ? pk_max(a, b) : pk_min(a, b); };
  auto worse = [](double a, double b) { return TOP ? pk_min(a, b) : pk_max(a, b); };
  auto head = [](uint32_t bits, uint32_t pay) { return __hiloint2double(int(bits), int(pay)); };
  // strips: [g][row] = (2nd, 3rd head of the group) 16 bytes -- one ds_read_b128 and one ds_write_b128 per step
  unsigned char *const sA = strips + size_t(r) * 16;
  double m[NG];
  {
    // column heads: first (TOP) or last (bottom) sample of each window column.  The row's column list is padded to
    // 3 * NG entries with the pseudo column (all sentinels), so this is branch-free: two 16-byte reads of the list,
    // every head read in flight at once.
    uint32_t pos[kHG * NG];
    // absolute LDS byte address of the image (the low 32 bits of a generic pointer into LDS): heads carry absolute
    // addresses, so a step's next-key read needs no base add
    const uint32_t img0 = uint32_t(reinterpret_cast<uintptr_t>(image));
    {
      const uint4 *cl4 = reinterpret_cast<const uint4 *>(cl);
#pragma unroll
      for (int v4 = 0; v4 < 2; ++v4) {
        const uint4 q4 = cl4[v4];
        const uint32_t wv[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j0 = v4 * 8 + 2 * u;
          if (j0 < kHG * NG) pos[j0] = img0 + uint32_t(int(wv[u] & 0xffffu) * pd.img_pitch + (TOP ? 1 : pd.S)) * 4u;
          if (j0 + 1 < kHG * NG) pos[j0 + 1] = img0 + uint32_t(int(wv[u] >> 16) * pd.img_pitch + (TOP ? 1 : pd.S)) * 4u;
        }
      }
    }
    uint32_t kb[kHG * NG];
#pragma unroll
    for (int j = 0; j < kHG * NG; ++j) kb[j] = lds_u32(pos[j]);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      double hd[kHG];
#pragma unroll
      for (int i = 0; i < kHG; ++i) hd[i] = head(kb[kHG * g + i], pos[kHG * g + i] | (uint32_t(g) << kPayGroupShift));
      sort_best_first<TOP, kHG>(hd);
      *reinterpret_cast<double2 *>(sA + g * (ROWS * 16)) = make_double2(hd[1], hd[2]);
      m[g] = hd[0];
    }
    sort_best_first<TOP, NG>(m);
  }
  int k = 0;
  int next_rank = nt > 0 ? (tl.ok ? __builtin_amdgcn_readlane(tl.rank, 0) : ldk(&tgt[0]).x) : -1;
  double prev = head(TOP ? kRawMin : kRawMax, 0);

  uint32_t nk;      // next key (float bits) of the popped column
  double2 h12;      // popped group's 2nd and 3rd head
  uint32_t lo_cur;  // payload of the popped head
  uint32_t aA;      // absolute LDS address of the popped group's strip: computed for the read, reused by the write-back
  const uint32_t sA0 = uint32_t(reinterpret_cast<uintptr_t>(sA));
  typedef double v2f64 __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) v2f64 *lds_d2;
  auto issue = [&](double top) {
    lo_cur = uint32_t(__double2loint(top));
    const uint32_t g = __builtin_amdgcn_ubfe(lo_cur, kPayGroupShift, 3);
    aA = __umul24(g, uint32_t(ROWS * 16)) + sA0;
    nk = lds_u32((lo_cur & 0x3ffffu) + (TOP ? 4u : uint32_t(-4)));  // payloads hold absolute LDS addresses
    {
      const v2f64 t = *reinterpret_cast<lds_d2>(uintptr_t(aA));
      h12 = make_double2(t.x, t.y);
    }
  };
  issue(m[0]);
  // One step (round 4: written back BEFORE the next step's reads are issued).  Rounds 2-3 issued the next step's reads
  // first and finished the group in their shadow, which needs a bypass when two consecutive steps pop from the same
  // group (the strip's write-back is still behind the read): one compare, six selects and a move per step.  The kernel
  // turned out to be bound as much by the instructions it issues as by the chain's latency (PMC round 4: one vector
  // instruction per 5.8 SIMD-cycles), and without the bypass a step is 8 instructions shorter at the price of 6 more on the
  // dependent chain: 14.46 -> 13.73 ms per 131 072 cells.  A wave's LDS operations execute in order, so the reads see the
  // write-back.  `prev` (the order statistic before the current one) is only read by an emission: it is captured by the
  // caller before the last step of a stretch, not in every step.
  auto do_step = [&]() {
    uint32_t pay = lo_cur + (TOP ? 4u : uint32_t(-4));
    if constexpr (TOP && TIER) {
      // Tiered image: slot tier_k + 1 of a column holds the marker 0x7ff00000 | column (a NaN pattern no sample has;
      // as a signed int it is above every sample and both sentinels).  Reading it means the column's next sample
      // lives in the workgroup's global tail: fetch it there.  A head that came from the tail keeps pointing at
      // slot tier_k (so its "next key" read finds the marker, and with it the column, again) and carries its own
      // position in payload bits 21..27 under flag bit 30.  Rare by construction (a column must supply more than
      // tier_k of a window's top ranks), so the wave takes this branch only when some lane needs it.
      const bool mk = int(nk) >= int(0x7ff00000u);
      if (__ballot(mk) != 0) {
        const uint32_t c = nk & 0xfffffu;
        const uint32_t p1n = (lo_cur & 0x40000000u) ? ((lo_cur >> 21) & 0x7fu) + 1u : uint32_t(pd.tier_k) + 1u;
        uint32_t v = kRawMin;  // past the column's last sample: the losing sentinel
        if (mk && p1n <= uint32_t(pd.S)) {
          const uint32_t off = ((p1n - uint32_t(pd.tier_k) - 1u) * uint32_t(pd.tail_pitch) + c) * 4u;
          // sc0 sc1: from L2 -- this CU's L1 may still hold the previous item's bytes of the same address
          asm volatile("global_load_dword %0, %1, %2 sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(off), "s"(tail_cur) : "memory");
        }
        if (mk) {
          nk = v;
          pay = 0x40000000u | (p1n << 21) | (lo_cur & 0x1fffffu);   // address and group kept
        }
      }
    }
    const double fresh = head(nk, pay);
    const double t0 = better(fresh, h12.x);  // the group's new top
    const double w1 = worse(fresh, h12.x);
    const double b1 = better(w1, h12.y);
    const double b2 = worse(w1, h12.y);
    *reinterpret_cast<lds_d2>(uintptr_t(aA)) = v2f64{b1, b2};
    double m0n = t0;
    if constexpr (NG >= 2) m0n = better(t0, m[1]);
    __builtin_amdgcn_sched_barrier(0);
    issue(m0n);  // the next step's reads go out before the cached tops are finished
    __builtin_amdgcn_sched_barrier(0);
    m[0] = m0n;
    if constexpr (NG >= 2) {
      double w = worse(t0, m[1]);
#pragma unroll
      for (int i = 1; i + 1 < NG; ++i) {
        const double nb = better(w, m[i + 1]);
        w = worse(w, m[i + 1]);
        m[i] = nb;
      }
      m[NG - 1] = w;
    }
  };
  int step = 0;
  while (true) {
    const int stop = (next_rank >= 0 && next_rank < steps) ? next_rank : steps;  // wave-uniform
    if (step < stop) {
      for (; step < stop - 1; ++step) do_step();
      prev = m[0];  // order statistic stop - 1: what an emission at `stop` pairs with
      do_step();
      ++step;
    }
    if (step >= steps) break;
#if defined(HDP_LANE_ABL) && (HDP_LANE_ABL & 4)
    {  // ablation: no emission (timing only)
      int kk = k + 1;
      while (kk < nt && ldk(&tgt[kk]).x == step) ++kk;
      k = kk;
      next_rank = kk < nt ? ldk(&tgt[kk]).x : -1;
    }
#else
    if (tl.ok)  // wave-uniform
      emit_targets_lanes<TOP>(pd, tl, nt, k, next_rank, step, __double2hiint(m[0]), __double2hiint(prev), rf, orow);
    else
      emit_targets<TOP, true>(pd, tgt, nt, k, next_rank, step, __double2hiint(m[0]), __double2hiint(prev), rf, true, orow);
#endif
    next_rank = __builtin_amdgcn_readfirstlane(next_rank);
#ifndef HDP_MERGE_UNFAIR
    // Two merging waves share a SIMD on half of a CU (six merging waves, four SIMDs).  At equal priority the issue
    // arbiter serves the OLDER wave first, and the younger one's 151-step chain came out 15 % longer (70.7 k against
    // 61 k cycles per cell, measured per wave: HDP_THR_DEBUG=1024) -- the whole workgroup waits for it at the barrier.
    // The two therefore trade places every 16 steps: the wave whose turn it is runs at priority 3, the other at 2.
    if (prio_phase >= 0) {  // wave-uniform
#ifndef HDP_MERGE_BIAS
#define HDP_MERGE_BIAS 2
#endif
      // the younger wave (prio_phase 1) is favoured in HDP_MERGE_BIAS of every 4 sixteen-step slices
      const bool young_turn = ((step >> 4) & 3) < HDP_MERGE_BIAS;
      if (young_turn == (prio_phase == 1)) __builtin_amdgcn_s_setprio(3);
      else __builtin_amdgcn_s_setprio(2);
    }
#endif
  }
}

template <int NG, bool TIER, int ROWS>
__device__ __forceinline__ void merge_both_lean(const ThrDev &pd, const unsigned char *image, unsigned char *strips,
                                                const float *tail_cur, const uint32_t *flags, const uint16_t *cl, int r,
                                                double *orow, const TgtLanes &tl_top, const TgtLanes &tl_bot,
                                                int prio_phase = -1) {
  // NaN / infinity census of the window: the columns' words are OR-ed first and summed only when one of them is set
  // (almost never: 15 reads and ORs instead of 15 reads and five operations each, per row and item)
  RowFlags rf{0, 0};
  uint32_t fw[kHG * NG], nan_or = 0;
#pragma unroll
  for (int j = 0; j < kHG * NG; ++j) {
    fw[j] = (j < pd.W) ? flags[cl[j]] : 0u;
    nan_or |= fw[j];
  }
  if (__ballot(nan_or != 0) != 0) {  // wave-uniform
#pragma unroll
    for (int j = 0; j < kHG * NG; ++j) {
      rf.n_pos += (fw[j] >> 15) & 0x7fff;
      rf.n_neg += fw[j] & 0x7fff;
    }
    if (nan_or >> 31) rf.n_pos = -1;
  }
  merge_row_lean<true, NG, TIER, ROWS>(pd, image, strips, tail_cur, cl, r, rf, orow, tl_top, prio_phase);
  merge_row_lean<false, NG, false, ROWS>(pd, image, strips, tail_cur, cl, r, rf, orow, tl_bot, prio_phase);
  if (prio_phase >= 0) __builtin_amdgcn_s_setprio(3);
}

// ---- segmented walks: enter the W-way merge at a pivot ----------------------------------------------------------------
// A multiway merge can be entered at ANY value v, not only at an end of the window: if c_j is the number of column j's
// samples ranked before v, the heads (c_0 .. c_W-1) are exactly the merge state after C = sum c_j pops -- whatever v is, as
// long as every sample before it counts and none after it does.  So a walk whose first requested rank R1 lies deep does
// not start at rank 0: the lane takes v = the p-th sample of the window's centre column (p chosen on the host so that C
// falls a few standard deviations short of R1), finds every c_j by a stride descent in the sorted LDS column (7 probes
// per column, all columns in flight together), builds its heads there and walks on from rank C.  A plan's ranks are cut
// into runs (ThrSeg) at the large gaps between them; every run has its own merging wave per 64 rows and its own strips,
// so the chains are short AND parallel (the median set: ~250 steps instead of 817; ranks spread over the whole window:
// six chains of ~150 instead of two of 750).  Exactness never depends on the pivot: only the length of the walk does.  If a
// lane's C overshoots (more than `limit` samples before the pivot, so the run's first rank would be missed) the wave
// takes a shallower pivot, at worst none (C = 0).  Lanes start from different ranks C; the wave counts ranks from the
// smallest and a lane joins in when the count reaches its own, so emissions stay wave-uniform.
#ifndef HDP_SEG_REFINE
#define HDP_SEG_REFINE 1  // pivot refinement rounds of a segmented walk's entry (0: the host's pivot only)
#endif
template <bool TOP, int NG, int ROWS>
__device__ __forceinline__ void merge_row_seg(const ThrDev &pd, const ThrSeg &sg, const int2 *tgt, const unsigned char *image,
                                              unsigned char *strips, const uint16_t *cl, int r, const RowFlags &rf,
                                              double *orow, bool store, const TgtLanes &tl) {
  static_assert(NG >= 1 && NG <= 7, "group id is three payload bits");
  const int steps = sg.steps, nt = sg.nt;
  if (steps == 0) return;
  auto better = [](double a, double b) { return TOP ? pk_max(a, b) : pk_min(a, b); };
  auto worse = [](double a, double b) { return TOP ? pk_min(a, b) : pk_max(a, b); };
  auto head = [](uint32_t bits, uint32_t pay) { return __hiloint2double(int(bits), int(pay)); };
  // merge order of two samples (raw float bits, sentinels included): the order of the doubles they are the high words of
  auto before = [](uint32_t k, uint32_t v) {  // k is popped before v by this walk (strictly)
    const double a = __hiloint2double(int(k), 0), b = __hiloint2double(int(v), 0);
    return TOP ? a > b : a < b;
  };
  const int rp = pd.RP;  // strip pitch of a segmented plan: the block's rows rounded up to 64 (not the ROWS of the classic form)
  unsigned char *const sA = strips + size_t(r) * 16;
  const uint32_t img0 = uint32_t(reinterpret_cast<uintptr_t>(image));
  const int S = pd.S, W = pd.W;
  constexpr int NC = kHG * NG;  // window slots (padding entries: the pseudo column)
  uint32_t cbase[NC];           // LDS byte address of slot 0 of every window column
  {
    const uint4 *cl4 = reinterpret_cast<const uint4 *>(cl);
#pragma unroll
    for (int v4 = 0; v4 < 2; ++v4) {
      const uint4 q4 = cl4[v4];
      const uint32_t wv[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j0 = v4 * 8 + 2 * u;
        if (j0 < NC) cbase[j0] = img0 + uint32_t(int(wv[u] & 0xffffu) * pd.img_pitch) * 4u;
        if (j0 + 1 < NC) cbase[j0 + 1] = img0 + uint32_t(int(wv[u] >> 16) * pd.img_pitch) * 4u;
      }
    }
  }
  // ---- entry: slot of every column's first head, and the rank C the lane starts from
  uint32_t start[NC];
  int C = 0;
  // pos_j = samples of column j ranked before v, by descent in strides from the column's own end; returns their sum.
  // The columns are sorted descending: slot 1 the largest, slot S the smallest, sentinels at 0 and S + 1 (never "before"
  // anything).  TOP: pos_j counts from slot 1 down;  bottom: pos_j counts from slot S up (slot S + 1 - idx).
  auto count_before = [&](uint32_t v, uint32_t (&pos)[NC]) {
#pragma unroll
    for (int j = 0; j < NC; ++j) pos[j] = 0;
    for (int stride = 64; stride >= 1; stride >>= 1) {  // S <= 100 in this kernel: 7 strides reach 127
      uint32_t idx[NC], k[NC];
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        idx[j] = min(pos[j] + uint32_t(stride), uint32_t(S + 1));
        k[j] = lds_u32(cbase[j] + (TOP ? idx[j] : uint32_t(S + 1) - idx[j]) * 4u);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NC; ++j) pos[j] = before(k[j], v) ? idx[j] : pos[j];
      __builtin_amdgcn_sched_barrier(0);
    }
    int c = 0;
#pragma unroll
    for (int j = 0; j < NC; ++j)
      if (j < W) c += int(pos[j]);
    return c;
  };
  {
    int p = sg.pivot_pos;  // wave-uniform
    float va = 0.0f;       // the accepted pivot (as a value)
    while (true) {
#pragma unroll
      for (int j = 0; j < NC; ++j) start[j] = uint32_t(TOP ? 1 : S);
      C = 0;
      if (p <= 0) break;
      // pivot: the MEAN of the window columns' p-th samples counted from the walk's own end (any value will do; one
      // column's p-th sample sits at a quantile level that is off by sqrt(p (S - p) / S) / S -- +-68 ranks of the window at
      // p = 38 of 100, measured as entry ranks 204 .. 441 inside one wave -- the mean of fifteen is four times closer).
      // An infinite or overflowing mean is still a valid pivot (nothing or everything ranks before it).
      uint32_t v;
      {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < NC; ++j)
          if (j < W) acc += __uint_as_float(lds_u32(cbase[j] + uint32_t(TOP ? p : S + 1 - p) * 4u));
        acc *= 1.0f / float(W);
        v = __float_as_uint(acc);
        va = acc;
        if (acc != acc) v = TOP ? kRawMax : kRawMin;   // +inf and -inf in one window: enter at the end instead
      }
      uint32_t pos[NC];
      C = count_before(v, pos);
#pragma unroll
      for (int j = 0; j < NC; ++j)
        if (j < W) start[j] = TOP ? pos[j] + 1u : uint32_t(S) - pos[j];
      if (__ballot(C > sg.limit) == 0) break;  // every lane can still emit the run's first rank (with its predecessor)
      p -= max(2, p >> 3);                     // some lane overshot: a slightly shallower pivot for the whole wave
    }
    // ---- refinement (round 4, second build): the host's pivot leaves 3.5 standard deviations of head room, and the wave
    // walks from its LOWEST lane's entry -- 112 steps before the run's first rank on average for the spread set, 140 for
    // the median set: at ~300 cycles a step that was most of a run.  Every lane now moves its own pivot towards its limit:
    // first by the mean distance to the keys (limit - slack - C) / W slots further along the columns it cuts, then by secant
    // steps on the count; a candidate that overshoots the limit is dropped (it still serves the next secant step).  A
    // round is one more count pass (7 strides: ~600 instructions, ~1 k cycles); it ends when every lane of the wave is
    // within kEntryNear ranks of its limit.
    if (p > 0 && HDP_SEG_REFINE > 0) {
      constexpr int kEntryNear = 8, kSlack = 6;
      float vb = 0.0f;
      int Cb = 0;
      bool have_b = false;
      for (int it = 0; it < HDP_SEG_REFINE; ++it) {
        const int gap = sg.limit - C;  // >= 0
        if (__ballot(gap > kEntryNear) == 0) break;
        const int target = sg.limit - (it == 0 ? max(kSlack, gap >> 2) : kSlack);
        float vn = va;
        if (gap > kEntryNear && va == va) {
          if (!have_b) {
            const int sh = (target - C + W - 1) / W;  // slots further along every column, >= 1
            float acc = 0.0f;
            int nroom = 0;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
              const uint32_t pj = TOP ? start[j] - 1u : uint32_t(S) - start[j];  // samples of column j before the pivot
              const uint32_t slot = pj + uint32_t(sh);                             // in walk order, 1-based
              const bool room = (j < W) && pj >= 1u && pj < uint32_t(S) && slot <= uint32_t(S);
              const float kf = __uint_as_float(lds_u32(cbase[j] + (TOP ? (room ? slot : 1u) : uint32_t(S + 1) - (room ? slot : 1u)) * 4u));
              acc += room ? kf - va : 0.0f;
              nroom += room ? 1 : 0;
            }
            if (nroom > 0) vn = va + acc * float(W) / (float(nroom) * float(nroom));
          } else if (Cb != C) {
            vn = va + float(target - C) * (vb - va) / float(Cb - C);
          }
          if (!(vn == vn) || fabsf(vn) == INFINITY) vn = va;
        }
        uint32_t pos[NC];
        const int Cn = count_before(__float_as_uint(vn), pos);
        const bool take = Cn <= sg.limit && Cn > C;
        if (take || Cn != C) {  // the other point of the next secant step: the pivot left behind, or the one that overshot
          vb = take ? va : vn;
          Cb = take ? C : Cn;
          have_b = true;
        }
        if (take) {
          va = vn;
          C = Cn;
#pragma unroll
          for (int j = 0; j < NC; ++j)
            if (j < W) start[j] = TOP ? pos[j] + 1u : uint32_t(S) - pos[j];
        }
      }
    }
  }
  double m[NG];
  {
    uint32_t pos[NC], kb[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) pos[j] = cbase[j] + start[j] * 4u;
#pragma unroll
    for (int j = 0; j < NC; ++j) kb[j] = lds_u32(pos[j]);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      double hd[kHG];
#pragma unroll
      for (int i = 0; i < kHG; ++i) hd[i] = head(kb[kHG * g + i], pos[kHG * g + i] | (uint32_t(g) << kPayGroupShift));
      sort_best_first<TOP, kHG>(hd);
      *reinterpret_cast<double2 *>(sA + g * (rp * 16)) = make_double2(hd[1], hd[2]);
      m[g] = hd[0];
    }
    sort_best_first<TOP, NG>(m);
  }
  // wave-wide range of the entry ranks (every lane of a merging wave runs a merge: reductions over all 64 lanes)
  int c_min = C, c_max = C;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    c_min = min(c_min, __shfl_xor(c_min, d, 64));
    c_max = max(c_max, __shfl_xor(c_max, d, 64));
  }
  c_min = __builtin_amdgcn_readfirstlane(c_min);
  c_max = __builtin_amdgcn_readfirstlane(c_max);
  if (HDP_DBG(pd, 2048) && (threadIdx.x & 63) == 0) {  // entry statistics of the segmented walks
    atomicAdd(&pd.clk[24], (unsigned long long)c_min);
    atomicAdd(&pd.clk[25], (unsigned long long)c_max);
    atomicAdd(&pd.clk[26], (unsigned long long)sg.limit);
    atomicAdd(&pd.clk[27], 1ull);
  }

  int k = 0;
  int next_rank = nt > 0 ? (tl.ok ? __builtin_amdgcn_readlane(tl.rank, 0) : ldk(&tgt[0]).x) : -1;
  double prev = head(TOP ? kRawMin : kRawMax, 0);
  uint32_t nk, lo_cur, aA;
  double2 h12;
  const uint32_t sA0 = uint32_t(reinterpret_cast<uintptr_t>(sA));
  typedef double v2f64 __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) v2f64 *lds_d2;
  auto issue = [&](double top) {
    lo_cur = uint32_t(__double2loint(top));
    const uint32_t g = __builtin_amdgcn_ubfe(lo_cur, kPayGroupShift, 3);
    aA = __umul24(g, uint32_t(rp * 16)) + sA0;
    nk = lds_u32((lo_cur & 0x3ffffu) + (TOP ? 4u : uint32_t(-4)));
    const v2f64 t = *reinterpret_cast<lds_d2>(uintptr_t(aA));
    h12 = make_double2(t.x, t.y);
  };
  auto do_step = [&]() {  // as merge_row_lean's (untiered)
    const uint32_t pay = lo_cur + (TOP ? 4u : uint32_t(-4));
    const double fresh = head(nk, pay);
    const double t0 = better(fresh, h12.x);
    const double w1 = worse(fresh, h12.x);
    const double b1 = better(w1, h12.y);
    const double b2 = worse(w1, h12.y);
    *reinterpret_cast<lds_d2>(uintptr_t(aA)) = v2f64{b1, b2};
    double m0n = t0;
    if constexpr (NG >= 2) m0n = better(t0, m[1]);
    __builtin_amdgcn_sched_barrier(0);
    issue(m0n);
    __builtin_amdgcn_sched_barrier(0);
    m[0] = m0n;
    if constexpr (NG >= 2) {
      double w = worse(t0, m[1]);
#pragma unroll
      for (int i = 1; i + 1 < NG; ++i) {
        const double nb = better(w, m[i + 1]);
        w = worse(w, m[i + 1]);
        m[i] = nb;
      }
      m[NG - 1] = w;
    }
  };
  issue(m[0]);
  // ranks c_min .. c_max - 1: lanes join in as the count reaches their entry rank (no requested rank lies here:
  // every lane has C <= limit < the run's first rank)
  int step = c_min;
  for (; step < c_max; ++step) {
    if (step >= C) {
      prev = m[0];
      do_step();
    }
  }
  while (true) {
    const int stop = (next_rank >= 0 && next_rank < steps) ? next_rank : steps;  // wave-uniform
    if (step < stop) {
      for (; step < stop - 1; ++step) do_step();
      prev = m[0];
      do_step();
      ++step;
    }
    if (step >= steps) break;
    if (tl.ok)  // wave-uniform
      emit_targets_lanes<TOP>(pd, tl, nt, k, next_rank, step, __double2hiint(m[0]), __double2hiint(prev), rf, orow, store);
    else
      emit_targets<TOP, true>(pd, tgt, nt, k, next_rank, step, __double2hiint(m[0]), __double2hiint(prev), rf, store, orow);
    next_rank = __builtin_amdgcn_readfirstlane(next_rank);
  }
}

// ---- rank selection (many samples per column) ---------------------------------------------------
// The merge walks every order statistic down to the deepest requested rank, one lane per row: with
// S = 1000 samples per column (a 10-member ensemble) that is 3000 dependent steps on the two dozen
// lanes whose rows fit the LDS.  Here one lane owns one (row, requested rank) pair instead and finds
// its order statistic directly in the window's W sorted columns:
//   keep, per column j, an interval [lo_j, hi_j] around c_j = the number of its keys ranked above the
//   wanted element; take the middle key of the widest interval as pivot, count by bisection inside
//   every interval the keys ranked above the pivot (ties are ordered by column, then position, so
//   the ranking is total), and move every lo_j or every hi_j to those counts depending on which side
//   of the wanted rank the pivot fell.  The pivot's own interval at least halves each round.
// When the intervals have closed, sum c_j = R and the R-th largest key is the best of the W heads
// col_j[c_j + 1]; the (R + 1)-th is the runner-up of those heads and the winner's successor.
// The adjacent pair (R, R + 1) is exactly what one interpolated quantile needs.
#ifndef HDP_SELECT_WARM
#define HDP_SELECT_WARM 1
#endif
#ifndef HDP_SELECT_VROUNDS
#define HDP_SELECT_VROUNDS 6  // value-pivot rounds at most
#endif
#ifndef HDP_SELECT_POPGO
#define HDP_SELECT_POPGO 16  // another value round while any lane of the wave misses by more than this
#endif
#ifndef HDP_SELECT_POPMAX
#define HDP_SELECT_POPMAX 16  // a miss of at most this many keys is popped, not bisected
// (rounds, go, max) on 4096 C5 cells: (4, 8, 8) 14.1 ms, (4, 16, 16) 13.1, (5, 16, 16) 12.6, (6, 16, 16) 12.5, (8, 16, 16) 12.5,
// (6, 4, 16) 13.1, (6, 16, 32) 12.5, (4, 64, 64) 14.3: the secant stalls at a few keys, and a lane left to the key pivots
// holds its whole wave
#endif
// pos[j] = key[j] > thr[j] ? idx[j] : pos[j] for all NC columns.  Written out by threes when NC allows: the compiler pairs
// every v_cmp with its v_cndmask and has to put two idle states between them (a VALU-written mask may not be read by
// the next two VALU instructions on gfx950); compare, compare, compare, select, select, select needs none.
template <int NC>
__device__ __forceinline__ void take_if_above(uint32_t (&pos)[NC], const uint32_t (&idx)[NC], const int (&key)[NC],
                                              const int (&thr)[NC]) {
  if constexpr (NC % 3 == 0) {
#pragma unroll
    for (int j = 0; j < NC; j += 3) {
      unsigned long long m0, m1, m2;
      asm volatile(
          "v_cmp_gt_i32_e64 %[m0], %[k0], %[t0]\n\t"
          "v_cmp_gt_i32_e64 %[m1], %[k1], %[t1]\n\t"
          "v_cmp_gt_i32_e64 %[m2], %[k2], %[t2]\n\t"
          "v_cndmask_b32_e64 %[p0], %[p0], %[i0], %[m0]\n\t"
          "v_cndmask_b32_e64 %[p1], %[p1], %[i1], %[m1]\n\t"
          "v_cndmask_b32_e64 %[p2], %[p2], %[i2], %[m2]"
          : [p0] "+v"(pos[j]), [p1] "+v"(pos[j + 1]), [p2] "+v"(pos[j + 2]), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2)
          : [k0] "v"(key[j]), [k1] "v"(key[j + 1]), [k2] "v"(key[j + 2]), [t0] "v"(thr[j]), [t1] "v"(thr[j + 1]),
            [t2] "v"(thr[j + 2]), [i0] "v"(idx[j]), [i1] "v"(idx[j + 1]), [i2] "v"(idx[j + 2]));
    }
  } else {
#pragma unroll
    for (int j = 0; j < NC; ++j) pos[j] = (key[j] > thr[j]) ? idx[j] : pos[j];
  }
}

template <int NC>
__device__ __forceinline__ void select_rows(const ThrDev &pd, const int *colk, const uint32_t *flags, int row0,
                                            int nrows, int tid, int64_t cell, double *__restrict__ out) {
  const int nt_all = pd.nt_top + pd.nt_bot;
  const int S = pd.S, W = pd.W;
  for (int task = tid; task < nrows * nt_all; task += kThrThreads) {
    const int r = task % nrows, ti = task / nrows;
    const bool top = ti < pd.nt_top;
    const int2 t = top ? pd.tgt_top[ti] : pd.tgt_bot[ti - pd.nt_top];
    const int p = t.y & 0xffff, kind = t.y >> 16;
    const bool pair = (kind == E_TOP_PAIR) || (kind == E_BOT_PAIR);
    // a = R-th largest (0-based), b = (R+1)-th largest.  Top list, rank k: best = k-th largest, prev = (k-1)-th;
    // bottom list, rank k: best = k-th smallest = (n-1-k)-th largest, prev = (n-k)-th largest.
    const int R = top ? (pair ? t.x - 1 : t.x) : (pd.n - 1 - t.x);
    const uint16_t *cl = pd.cols_local + size_t(row0 + r) * W;

    // Everything below is in LDS BYTE ADDRESSES (round 4): lo/hi/pos of column j are the addresses of its slots lo_j,
    // hi_j, pos_j (slot 0 = the sentinel before the column), so a stride trip costs add, min, read, compare, select per
    // column and nothing else -- the index form spent a sixth instruction on the address and, because the compiler kept
    // the "this column does not move" test as a scalar mask, two scalar ones per column on top.
    const uint32_t colk_a = uint32_t(uintptr_t(colk));
    uint32_t lim[NC], lo[NC], hi[NC];
    RowFlags rf{0, 0};
    uint32_t nan_or = 0;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const int c = (j < W) ? int(cl[j]) : 0;
      const uint32_t f = (j < W) ? flags[c] : 0u;
      nan_or |= f;
      rf.n_pos += (f >> 15) & 0x7fff;
      rf.n_neg += f & 0x7fff;
      lo[j] = colk_a + uint32_t(c * pd.S_pad) * 4u;
      lim[j] = lo[j] + uint32_t(S + 1) * 4u;  // the sentinel after the column is never above
      hi[j] = lo[j] + ((j < W) ? uint32_t(min(S, R)) * 4u : 0u);  // no column holds more than the R keys above the wanted one
    }
    if (nan_or >> 31) rf.n_pos = -1;
    uint32_t sum_base = 0;  // G = (sum of pos - sum of slot-0 addresses) / 4
#pragma unroll
    for (int j = 0; j < NC; ++j) sum_base += lo[j];
    const uint32_t R4 = uint32_t(R) * 4u + sum_base;  // the wanted sum of addresses

    // ---- value rounds + pops (round 4 of the build) ----------------------------------------------------------------
    // The pivot of these rounds is a VALUE v, not a key: count G(v) = keys above v over the window, and G - R is how far v
    // missed.  Round 1 takes the mean over the window's columns of their keys at the expected position R / W; round 2 moves
    // v by the mean distance to the keys (G - R) / W slots further along the columns v actually cuts; later rounds are
    // secant steps on G(v), which is smooth at this scale (15 000 samples): |G - R| goes 115 .. 850 -> 10 .. 60 -> 2 .. 25
    // -> 2 -> 1 on the bench generator.  After every count the miss bounds every column's remaining move, so the
    // intervals -- and with them the number of strides -- collapse to that width at once.  When the miss is down to a
    // few keys the lane does not bisect on (closing fifteen intervals of width 1 - 2 with key pivots took five to eight
    // more rounds, and a wave runs as long as its slowest lane): it POPS that many keys off the W heads, forwards or
    // backwards, with the heads packed into doubles (key in the high bits, LDS address in the low ones) so the best head
    // and its column come out of one chain of v_max_f64.  Whatever is still open after that -- ties by the thousand,
    // infinities in the mean, a stalled secant -- goes through the key-pivot loop below, which always terminates.
    constexpr int kValueRounds = HDP_SELECT_VROUNDS, kPopMax = HDP_SELECT_POPMAX, kPopGo = HDP_SELECT_POPGO;
    uint32_t pos[NC];
    if (HDP_SELECT_WARM) {
      float v, v0 = 0.f;
      int e = 0, e0 = 0;
      {
        const uint32_t m4 = uint32_t(min(max((R + (W >> 1)) / W, 1), min(S, R))) * 4u;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < NC; ++j) acc += (j < W) ? key_f32(int(lds_u32(lo[j] + m4))) : 0.f;
        v = acc / float(W);
      }
      for (int k = 0; k < kValueRounds; ++k) {
        uint32_t ww = 0;
#pragma unroll
        for (int j = 0; j < NC; ++j) ww = max(ww, hi[j] - lo[j]);
        const bool open = ww != 0;
        if (__ballot(open) == 0) break;
        const int pkey = f32_key(v);
        int thr[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) {
          pos[j] = lo[j];
          thr[j] = (j < W) ? pkey : 0x7fffffff;
          asm volatile("" : "+v"(thr[j]));
        }
        int nb = 11;
        while (nb > 0 && __ballot(open && (ww >> (nb + 1)) != 0) == 0) --nb;
        for (int sb = nb - 1; sb >= 0; --sb) {
          const uint32_t stride = 4u << sb;
          uint32_t idx[NC];
          int kk[NC];
#pragma unroll
          for (int j = 0; j < NC; ++j) {
            idx[j] = min(pos[j] + stride, lim[j]);
            kk[j] = int(lds_u32(idx[j]));
          }
          __builtin_amdgcn_sched_barrier(0);
          take_if_above<NC>(pos, idx, kk, thr);
          __builtin_amdgcn_sched_barrier(0);
        }
        uint32_t G4 = 0;
#pragma unroll
        for (int j = 0; j < NC; ++j) G4 += pos[j];
        if (open) {
          // A column whose count lies past its interval stops at the end of the stride range, at or past hi_j: G is then
          // an underestimate, but only on the side where that keeps every conclusion below valid.
          e0 = e;
          e = int(G4 - R4);
          if (e == 0) {
#pragma unroll
            for (int j = 0; j < NC; ++j) lo[j] = hi[j] = pos[j];
          } else if (e > 0) {  // v ranks below the wanted key
#pragma unroll
            for (int j = 0; j < NC; ++j) {
              hi[j] = min(hi[j], pos[j]);
              lo[j] = uint32_t(max(int(lo[j]), int(pos[j]) - e));
            }
          } else {  // v ranks above it
#pragma unroll
            for (int j = 0; j < NC; ++j) {
              lo[j] = pos[j];
              hi[j] = min(hi[j], pos[j] + uint32_t(-e));
            }
          }
          float vn = v;
          if (k == 0) {
            // the keys (G - R) / W slots from where v cuts each column, in the direction v has to move; columns v does not
            // cut (none or all of their keys above it) have no say, and the step is scaled up for their absence
            const int sh4 = ((e + (e > 0 ? 4 * W - 1 : -(4 * W - 1))) / (4 * W)) * 4;  // bytes, away from zero
            float acc = 0.f;
            int nroom = 0;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
              const uint32_t slot = pos[j] - uint32_t(sh4);
              const uint32_t first = lim[j] - uint32_t(S) * 4u;  // slot 1
              const bool room = (j < W) && pos[j] >= first && pos[j] < lim[j] - 4u && slot >= first && slot < lim[j];
              const float kf = key_f32(int(lds_u32(room ? slot : first)));
              acc += room ? kf - v : 0.f;
              nroom += room ? 1 : 0;
            }
            if (nroom > 0) vn = v + acc * float(W) / (float(nroom) * float(nroom));
          } else {
            const float dG = float(e - e0);
            if (dG != 0.f) vn = v - float(e) * (v - v0) / dG;
          }
          if (!(vn == vn)) vn = v;
          v0 = v;
          v = vn;
        }
        if (k >= 1 && __ballot(open && e != 0 && abs(e) > kPopGo * 4) == 0) break;
      }
      // pops: |e| / 4 keys to add (e < 0, from the heads after pos) or to give back (e > 0, from the keys at pos)
      uint32_t wl = 0;
#pragma unroll
      for (int j = 0; j < NC; ++j) wl = max(wl, hi[j] - lo[j]);
      const bool popl = wl != 0 && e != 0 && abs(e) <= kPopMax * 4;
      if (__ballot(popl) != 0) {
        const bool fwd = e < 0;
        const int n = popl ? (abs(e) >> 2) : 0;
        const uint32_t flip = fwd ? 0u : 0xffffffffu;  // backwards the smallest key goes first: ~key reverses the order
        const uint32_t step = fwd ? 4u : uint32_t(-4);
        // an int as a double is exact and leaves its low 22 bits clear: 18 of address and 4 of column number, which keeps
        // the packed heads of a window that lists one column twice (the days past the year's end do) apart
        auto pack = [](int key, uint32_t addr) {
          const double d = double(key);
          return __hiloint2double(__double2hiint(d), int(uint32_t(__double2loint(d)) | addr));
        };
        double hk[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) {
          const uint32_t ad = pos[j] + (fwd ? 4u : 0u);
          const int key = int(lds_u32(ad) ^ flip);
          hk[j] = (j < W) ? pack(key, ad | uint32_t(j) << 18) : pack(int(0x80000000), 0u);
        }
        for (int i = 0; __ballot(i < n) != 0; ++i) {
          double best = hk[0];
#pragma unroll
          for (int j = 1; j < NC; ++j) best = pk_max(best, hk[j]);
          const uint32_t bl = uint32_t(__double2loint(best));
          const uint32_t na = (bl & 0x3ffffu) + step;
          const double nw = pack(int(lds_u32(na) ^ flip), na | (bl & 0x3c0000u));
          const uint32_t match = (i < n) ? bl : 0xffffffffu;  // no packed low word is all ones (addresses end below 2^18)
#pragma unroll
          for (int j = 0; j < NC; ++j) hk[j] = (uint32_t(__double2loint(hk[j])) == match) ? nw : hk[j];
        }
        if (popl) {
#pragma unroll
          for (int j = 0; j < NC; ++j) {
            const uint32_t ad = uint32_t(__double2loint(hk[j])) & 0x3ffffu;
            if (j < W) lo[j] = hi[j] = fwd ? ad - 4u : ad;
          }
        }
      }
    }

    while (true) {
      // widest interval -> pivot (its middle key: the pivot's own interval at least halves every round)
      int wj = 0;
      uint32_t ww = hi[0] - lo[0], wlo = lo[0];
#pragma unroll
      for (int j = 1; j < NC; ++j) {
        const uint32_t w = hi[j] - lo[j];
        const bool better = w > ww;
        ww = better ? w : ww;
        wj = better ? j : wj;
        wlo = better ? lo[j] : wlo;
      }
      if (ww == 0) break;
      const uint32_t mid = wlo + ((((ww >> 2) + 1) >> 1) << 2);  // slot in [lo + 1, hi]
      const int pkey = int(lds_u32(mid));
      constexpr bool vp = false;
      // Per column: c_j = number of keys ranked above the pivot.  lo_j <= c_j <= hi_j is known, so a
      // descent in power-of-two strides from lo_j finds it without looking at hi_j: key > thr[j] is
      // "ranked above the pivot" (an equal key of an earlier column ranks above it, of a later column
      // below it); the pivot's own column and the padding columns never move.
      int thr[NC];
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        const bool fixed = (j == wj) || (j >= W);
        pos[j] = (j == wj) ? mid - 4u : lo[j];
        thr[j] = fixed ? 0x7fffffff : pkey - ((j < wj && !vp) ? 1 : 0);
        asm volatile("" : "+v"(thr[j]));  // a plain register operand: no per-column scalar mask in the stride loop
      }
      int nb = 11;  // strides 2^(nb-1) .. 1 cover every interval of the wave's lanes
      while (nb > 0 && __ballot((ww >> (nb + 1)) != 0) == 0) --nb;
      for (int sb = nb - 1; sb >= 0; --sb) {
        const uint32_t stride = 4u << sb;
        // all NC reads of a stride in flight before the first compare (the scheduler otherwise may pair every
        // read with its own wait, which exposes the LDS latency NC times per stride: 147 k -> 222 k cycles per block)
        uint32_t idx[NC];
        int k[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) {
          idx[j] = min(pos[j] + stride, lim[j]);
          k[j] = int(lds_u32(idx[j]));
        }
        __builtin_amdgcn_sched_barrier(0);
        take_if_above<NC>(pos, idx, k, thr);
        __builtin_amdgcn_sched_barrier(0);
      }
      uint32_t G4 = 0;
#pragma unroll
      for (int j = 0; j < NC; ++j) G4 += pos[j];
      if (G4 == R4) {  // the pivot is the wanted element
#pragma unroll
        for (int j = 0; j < NC; ++j) lo[j] = hi[j] = pos[j];
      } else if (G4 > R4) {  // pivot ranked below it: at most these many keys of each column are above it ...
        const uint32_t back = HDP_SELECT_WARM ? G4 - R4 : (1u << 30);  // ... and at most G - R fewer than above the pivot
#pragma unroll
        for (int j = 0; j < NC; ++j) {
          hi[j] = pos[j];
          lo[j] = max(int(lo[j]), int(pos[j] - back));  // LDS addresses are far below 2^31
        }
      } else {  // pivot ranked above it (so the pivot, when it is a key, counts in its own column)
        const uint32_t fwd = HDP_SELECT_WARM ? R4 - G4 - (vp ? 0u : 4u) : (1u << 30);  // keys still to be placed
#pragma unroll
        for (int j = 0; j < NC; ++j) {
          lo[j] = pos[j] + ((j == wj) ? 4u : 0u);
          hi[j] = min(hi[j], lo[j] + fwd);
        }
      }
    }

    // heads after the R keys above the wanted one
    int a = kKeyMin, b = kKeyMin;
    uint32_t apos = lo[0];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const int k = (j < W) ? int(lds_u32(lo[j] + 4u)) : kKeyMin;
      const bool win = k > a;
      b = win ? a : max(b, k);
      apos = win ? lo[j] + 4u : apos;
      a = win ? k : a;
    }
    if (pair) b = max(b, int(lds_u32(apos + 4u)));  // R + 1 <= n - 1 for a pair: at most the sentinel's slot S + 1
    const float fa = key_f32(a), fb = key_f32(b);
    const QuantileParam qp = pd.qp[p];
    const float q_hi = fa, q_lo = pair ? fb : fa;
    out[cell * pd.n_doy * int64_t(pd.P) + size_t(p) * pd.n_doy + row0 + r] =
        finish_quantile(qp, q_lo, q_hi, rf.n_pos < 0, rf.n_pos, rf.n_neg, pd.n);
  }
}

// SELECT: rank selection instead of the merge; its interval state wants more than 128 VGPRs, and the
// deep-merge plans it serves fill the LDS with one workgroup per CU anyway (2 waves per SIMD).
template <int EPL, bool SELECT>
// second bound: 4 waves per SIMD = two 512-thread workgroups per CU (<= 128 VGPRs)
__global__ __launch_bounds__(kThrThreads, SELECT ? 2 : 4) void thresholds_kernel(ThrDev pd, const float *__restrict__ x,
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
  off += 2 * ((size_t(pd.ncols_max) * 4 + 15) & ~size_t(15));  // (the plan sizes two copies)
  float *hbuf = reinterpret_cast<float *>(smem + off);
  off += size_t(pd.Wp) * pd.RP * 4;
  uint32_t *posb = reinterpret_cast<uint32_t *>(smem + off);

  const int64_t cell = blockIdx.x;
  if (cell >= n_cells) return;
  const float *xc = x + cell * pd.xp;

  for (int b = 0; b < pd.n_blocks; ++b) {
    const int row0 = pd.blk_row0[b];
    const int nrows = pd.blk_nrows[b];
    const int ncols = pd.blk_ncols[b];
    const int2 *list = pd.load_list + pd.blk_list_off[b];
    const int llen = pd.blk_list_len[b];

    unsigned long long t_a = 0, t_b = 0, t_c = 0, t_d = 0;
    if (HDP_DBG(pd, 8)) t_a = __builtin_readcyclecounter();
    // 1. sentinels + load
    // `ncols` LDS column slots are (re)loaded and sorted by this block; the others keep the sorted columns of
    // earlier blocks (ring schedule of select-only plans; everywhere else the list is 0 .. ncols-1)
    const int32_t *slots = pd.sort_slots + pd.blk_sort_off[b];
    for (int i = tid; i < ncols; i += kThrThreads) {
      const int lc = slots[i];
      colbuf[lc * pd.S_pad] = __int_as_float(kKeyMax);          // below every ascending walk
      colbuf[lc * pd.S_pad + pd.S + 1] = __int_as_float(kKeyMin);  // below every descending walk
    }
    if (!HDP_DBG(pd, 4)) {
      // batches of kLoadBatch independent (list entry -> sample -> LDS) chains per thread: all list
      // reads of a batch are issued before the first sample read, all sample reads before the first
      // LDS write, so a thread keeps kLoadBatch HBM requests in flight instead of one
      constexpr int kLoadBatch = 24;  // 8 -> 24: the C5 load phase is six dependent list -> sample rounds shorter (-5.6 % per step)
      for (int base = tid; base < llen; base += kThrThreads * kLoadBatch) {
        int2 e[kLoadBatch];
        float v[kLoadBatch];
#pragma unroll
        for (int u = 0; u < kLoadBatch; ++u) e[u] = list[min(base + u * kThrThreads, llen - 1)];
#pragma unroll
        for (int u = 0; u < kLoadBatch; ++u) v[u] = xc[e[u].x];
#pragma unroll
        for (int u = 0; u < kLoadBatch; ++u)
          if (base + u * kThrThreads < llen) colbuf[e[u].y] = v[u];
      }
    }
    __syncthreads();
    if (HDP_DBG(pd, 8)) t_b = __builtin_readcyclecounter();

    // 2. sort every column once
    if (HDP_DBG(pd, 2)) {
    } else if (pd.S <= 128 && EPL <= 2) {
      if (pd.S <= 8) sort_block_rows<1>(colbuf, pd.S_pad, pd.S, ncols, flags, wave, nwaves, lane);
      else if (pd.S <= 16) sort_block_rows<2>(colbuf, pd.S_pad, pd.S, ncols, flags, wave, nwaves, lane);
      else if (pd.S <= 32) sort_block_rows<4>(colbuf, pd.S_pad, pd.S, ncols, flags, wave, nwaves, lane);
      else if (pd.S <= 64) sort_block_rows<8>(colbuf, pd.S_pad, pd.S, ncols, flags, wave, nwaves, lane);
      else sort_block_rows<16>(colbuf, pd.S_pad, pd.S, ncols, flags, wave, nwaves, lane);
    } else {
      for (int i = wave; i < ncols; i += nwaves) {
        const int lc = slots[i];
        sort_column<EPL>(colbuf + lc * pd.S_pad + 1, pd.S, &flags[lc], lane);
      }
    }
    __syncthreads();
    if (HDP_DBG(pd, 8)) t_c = __builtin_readcyclecounter();

    // 3. + 4. merge and interpolate, one lane per row; every quantile is stored as soon as its
    //         second order statistic comes out of the merge
    if constexpr (SELECT) {
      const int *colk = reinterpret_cast<const int *>(colbuf);
      if (HDP_DBG(pd, 1)) {
      } else if (pd.W <= 4) select_rows<4>(pd, colk, flags, row0, nrows, tid, cell, out);
      else if (pd.W <= 8) select_rows<8>(pd, colk, flags, row0, nrows, tid, cell, out);
      else if (pd.W == 15) select_rows<15>(pd, colk, flags, row0, nrows, tid, cell, out);  // the default radius 7
      else select_rows<16>(pd, colk, flags, row0, nrows, tid, cell, out);
    } else if (tid < nrows) {
      const int row = row0 + tid;
      const uint16_t *cl = pd.cols_local + size_t(row) * pd.W;
      double *orow = out + cell * pd.n_doy * int64_t(pd.P) + row;  // [cell][P][n_doy]
      if (!HDP_DBG(pd, 1)) {
        switch (pd.Wp >> 2) {
          case 1: merge_both<1>(pd, colbuf, hbuf, posb, flags, cl, tid, true, orow); break;
          case 2: merge_both<2>(pd, colbuf, hbuf, posb, flags, cl, tid, true, orow); break;
          case 4: merge_both<4>(pd, colbuf, hbuf, posb, flags, cl, tid, true, orow); break;
          default: merge_both<0>(pd, colbuf, hbuf, posb, flags, cl, tid, true, orow); break;
        }
      }
    }
    __syncthreads();
    if (HDP_DBG(pd, 8) && tid == 0) {
      t_d = __builtin_readcyclecounter();
      atomicAdd(&pd.clk[0], t_b - t_a);
      atomicAdd(&pd.clk[1], t_c - t_b);
      atomicAdd(&pd.clk[2], t_d - t_c);
      atomicAdd(&pd.clk[3], 1ull);
    }
  }
}

// ---- lane-per-column producers (S <= 104): the column sort as a register network -----------------------
// The 16-lane-row sorter above spends 10 of its 28 stages crossing lanes (a v_mov_dpp + v_med3 pair per key) and
// pads every column to 128 slots.  Here ONE LANE owns one day-of-year column: its S samples sit in N >= S registers
// and are sorted by Batcher's merge exchange (Knuth 5.2.2 M) for exactly N keys, fully unrolled -- compare-exchanges
// are v_max + v_min between two registers, nothing crosses lanes, no padding to a power of two (N = 100: 1077
// compare-exchanges per 64 columns against 4 x 450 instructions per 4 columns before: 3.3x fewer vector
// instructions for the sort).  Loads are coalesced without any transposition: the samples of one year of 64
// adjacent columns are 64 adjacent time steps.  The merging waves, the LDS image and the emission are the ones of
// the round-2 pipelined kernel this one replaced; results are bit-identical to the one-workgroup-per-cell kernel
// (tests/test_thresholds_kernels_gpu.py).
constexpr int ilog2_ceil(int n) {
  int t = 0;
  while ((1 << t) < n) ++t;
  return t;
}
// compare-exchange of two samples (float bits), larger to `a`.  asm volatile: the statements keep their program
// order, so the N keys plus one temporary are all that is ever live -- left to itself the scheduler stretches the
// live ranges of a 1000-comparator network until it spills (N = 100: 256 VGPRs and 417 spills, measured).
// The operands are NaN-free float bit patterns; denormals are preserved (FLOAT_DENORM_MODE_32 = 3 in these kernels).
__device__ __forceinline__ void ce_key(int &a, int &b) {
  int hi;
  asm volatile("v_max_f32 %0, %1, %2\n\tv_min_f32 %1, %1, %2" : "=&v"(hi), "+v"(b) : "v"(a));
  a = hi;
}
// The passes (p, r, d) of merge exchange for N keys, as compile-time constants: pass k compares v[i] with v[i + d]
// for every i with (i & p) == r.  Written as a recursion over the pass number so that every index is a constant
// after inlining (a `break` inside an unrolled loop left run-time indices, i.e. the keys in scratch memory).
struct LanePass {
  int p, r, d;
};
template <int N>
struct LaneNet {
  static constexpr int t = ilog2_ceil(N);
  static constexpr int count() {
    int c = 0;
    for (int pi = t - 1; pi >= 0; --pi) {
      const int p = 1 << pi;
      int q = 1 << (t - 1);
      for (;;) {
        ++c;
        if (q == p) break;
        q >>= 1;
      }
    }
    return c;
  }
  static constexpr LanePass pass(int k) {
    int c = 0;
    for (int pi = t - 1; pi >= 0; --pi) {
      const int p = 1 << pi;
      int q = 1 << (t - 1), r = 0, d = p;
      for (;;) {
        if (c == k) return LanePass{p, r, d};
        ++c;
        if (q == p) break;
        d = q - p;
        q >>= 1;
        r = p;
      }
    }
    return LanePass{0, 0, 0};
  }
};
template <int N, int K = 0>
__device__ __forceinline__ void sort_lane_desc(int (&v)[N]) {  // larger key to the lower index
  if constexpr (N > 1 && K < LaneNet<N>::count()) {
    constexpr LanePass ps = LaneNet<N>::pass(K);
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (i + ps.d < N && (i & ps.p) == ps.r) ce_key(v[i], v[i + ps.d]);
    sort_lane_desc<N, K + 1>(v);
  }
}

// global_load_dword with a wave-uniform base (SGPR pair) and a 32-bit byte offset per lane: no 64-bit address
// arithmetic in vector registers, and the result may land in the register that held the offset.  The caller
// waits (s_waitcnt vmcnt(0)) before it uses the value: the compiler does not know this is a load.
__device__ __forceinline__ int ld_saddr(const void *sbase, uint32_t voff) {
  int r;
  asm volatile("global_load_dword %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
  return r;
}
// the same, overwriting the offset register with the loaded value
__device__ __forceinline__ void ld_saddr_inplace(const void *sbase, int &reg) {
  asm volatile("global_load_dword %0, %0, %1" : "+v"(reg) : "s"(sbase) : "memory");
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// To the compiler an asm load has produced its value when the statement ends, so ordinary code that reads the
// register may be scheduled ABOVE the s_waitcnt (it was: the census ran on the byte offsets).  An empty volatile asm
// that "rewrites" the register after the wait makes every later use depend on a statement that stays below it.
__device__ __forceinline__ void landed(int &reg) { asm volatile("" : "+v"(reg)); }

// slots below this index hold a sample for every S the N-slot kernel is chosen for (S > the next smaller N)
constexpr int lane_first_pad_slot(int N) {
  constexpr int kN[] = {0, 16, 32, 64, 80, 100};
  int prev = 0;
  for (int n : kN)
    if (n < N) prev = n;
  return prev;
}

// tasks (64 columns each) one producer wave sorts and holds per block
template <int N>
constexpr int lane_tasks_per_wave() { return N >= 64 ? 1 : (N >= 32 ? 2 : 4); }

#ifndef HDP_WHOLE_MINW
#define HDP_WHOLE_MINW (ROWS == kWholeRows ? 3 : 4)
#endif
// ROWS = strip pitch: kLeanRows (blocks of <= 128 rows, up to 8 waves, two or three workgroups per CU) or kWholeRows (every
// day-of-year row of a cell in one 12-wave workgroup per CU: no halo columns -- each column is sorted once per cell --
// and all 365 merge chains of the cell in flight at once)
template <int N, int NG, bool TIER, int ROWS>
__global__ __launch_bounds__(ROWS == kWholeRows ? kWholeThreads : kThrThreads, HDP_WHOLE_MINW) void thresholds_lane_kernel(ThrDev pd, const float *__restrict__ x,
                                                                      int64_t n_cells, double *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = int(blockDim.x) >> 6;  // merging waves + as many producers as the block's tasks need (<= 8)
  constexpr int TPW = lane_tasks_per_wave<N>();

  size_t off = 0;
  float *colbuf = reinterpret_cast<float *>(smem + off);
  off += (size_t(pd.ncols_max + 1) * pd.img_pitch * 4 + 15) & ~size_t(15);  // + one pseudo column of sentinels
  uint32_t *flags0 = reinterpret_cast<uint32_t *>(smem + off);  // census words, double-buffered by block parity
  const int flags_pitch = ((pd.ncols_max * 4 + 15) & ~15) >> 2;
  off += 2 * size_t(flags_pitch) * 4;
  unsigned char *strips = smem + off;  // merge heads: 2nd..4th of every (group, row), see merge_row_lean
  constexpr bool kCanSeg = ROWS == kLeanRows;  // the whole-cell form never cuts its walks into runs: nothing of this in its code
  const int n_segs = kCanSeg ? pd.n_segs : 0;
  const size_t seg_strip_bytes = size_t(NG) * size_t(pd.RP) * 16;  // segmented: one set of strips per run, pitch = rows rounded to 64
  off += n_segs > 0 ? seg_strip_bytes * size_t(n_segs) : lean_strip_bytes<NG, ROWS>();
  uint16_t *cl_lds = reinterpret_cast<uint16_t *>(smem + off);  // [rows][kListPitch] local columns of this block's windows (slots past W: the pseudo column)

  const int nb = pd.n_blocks;
  const int blk = int(blockIdx.x) % nb;
  const int64_t wg_per_blk = gridDim.x / nb;  // the host launches a multiple of n_blocks workgroups
  const int64_t first_cell = int64_t(blockIdx.x) / nb;
  const int64_t n_items = first_cell < n_cells ? (n_cells - first_cell + wg_per_blk - 1) / wg_per_blk : 0;
  const int row0 = pd.blk_row0[blk];
  const int nrows = pd.blk_nrows[blk];
  const int ncols = pd.blk_ncols[blk];
  const int n_tasks = (ncols + 63) >> 6;
  const uint32_t pitch4 = uint32_t(n_tasks) * 256u;    // bytes per row of the offset table
  const int32_t *tl = pd.tixl + pd.blk_tixl_off[blk];  // [N][64 * tasks] byte offsets of (sample, local column)
  // this workgroup's global tail (tiered image), double-buffered by item parity like the census words
  const size_t tail_half = size_t(max(pd.S - pd.tier_k, 0) + 1) * pd.tail_pitch;
  float *const tail_wg = pd.tail + size_t(blockIdx.x) * 2 * tail_half;
  for (int i = tid; i < nrows * kListPitch; i += int(blockDim.x)) {
    const int r = i / kListPitch, j = i % kListPitch;
    cl_lds[i] = (j < pd.W) ? pd.cols_local[size_t(row0 + r) * pd.W + j] : uint16_t(pd.ncols_max);
  }
  // the pseudo column: every slot a head could start from reads as a loser (top walks start at slot 1, bottom walks at
  // slot S); it is never written again
  for (int i = tid; i < pd.img_pitch; i += int(blockDim.x))
    colbuf[size_t(pd.ncols_max) * pd.img_pitch + i] = __uint_as_float(i <= 2 ? kRawMin : kRawMax);

  // Roles by SIMD: one merging wave per SIMD first.
  const int n_merge = pd.n_merge;
  const int n_prod = nwaves - n_merge;
  __shared__ int s_simd[kWholeThreads / 64];
  uint32_t hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  const int my_simd = int((hwid >> 4) & 3u);
  const int tg_par = int((hwid >> 16) & 1u);
  if (lane == 0) s_simd[wave] = my_simd;
  __syncthreads();
  int rank = 0;
  {
    int occ_me = 0;
    for (int w = 0; w < wave; ++w) occ_me += (s_simd[w] == my_simd);
    const int key_me = (occ_me * 4 + ((my_simd - 2 * tg_par) & 3)) * nwaves + wave;
    for (int w = 0; w < nwaves; ++w) {
      const int sw = s_simd[w];
      int occ = 0;
      for (int u = 0; u < w; ++u) occ += (s_simd[u] == sw);
      const int key = (occ * 4 + ((sw - 2 * tg_par) & 3)) * nwaves + w;
      rank += (key < key_me);
    }
  }
  rank = __builtin_amdgcn_readfirstlane(rank);
  const bool producer = rank >= n_merge;  // wave-uniform
  if (producer) __builtin_amdgcn_s_setprio(0);
  else __builtin_amdgcn_s_setprio(3);
  const int pw = rank - n_merge;      // producer index
  // merging waves: row of the block this lane merges.  Segmented plans: merging wave `rank` walks run rank / nm_rows of
  // the rows (rank % nm_rows) * 64 .. + 63, with the run's own strips.
  const int nm_rows = n_segs > 0 ? max(n_merge / n_segs, 1) : n_merge;
  const int seg = n_segs > 0 ? rank / nm_rows : 0;
  const int mrow = (n_segs > 0 ? rank % nm_rows : rank) * 64 + lane;

  // Two loops, one per role, with the same two barriers per item ("image free / keys sorted", "image written"):
  // s_barrier counts wave arrivals, so waves may reach it from different code.  Written as one loop, the sorted keys
  // (up to 100 registers) were live across the merging waves' code as well and the kernel spilled.
  if (producer) {
    for (int64_t s = 0; s <= n_items; ++s) {
      int v[TPW][N];  // sorted samples (float bits) of this wave's tasks
      uint32_t *flags_p = flags0 + int(s & 1) * flags_pitch;  // census of block s (double-buffered by parity)
      unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
      const bool clocked = HDP_DBG(pd, 32) && lane == 0 && pw == 0;
      const bool clocked_w = HDP_DBG(pd, 1024) && lane == 0;
      if (clocked || clocked_w) c0 = __builtin_readcyclecounter();
      // S as a value made inside the loop: everything derived from it (which slots are padding, where they are
      // written) is then recomputed per item with scalar instructions instead of being hoisted into registers
      int S_rt = pd.S;
      asm volatile("" : "+s"(S_rt));
      if (s < n_items) {
        const int64_t cell_p = first_cell + s * wg_per_blk;
        const char *xc = reinterpret_cast<const char *>(x + cell_p * pd.xp);
#pragma unroll
        for (int k = 0; k < TPW; ++k) {
          const int task = pw + k * n_prod;  // wave-uniform
          if (task < n_tasks) {
            const int lc = task * 64 + lane;
            const uint32_t lane_b = uint32_t(lc) * 4u;
            // byte offsets first (coalesced, L2-resident table), then the samples: lanes = adjacent columns =
            // adjacent time steps, so a load instruction fetches 256 contiguous bytes on a regular calendar.
            // The table has N rows: rows past S repeat row S - 1 and are overwritten with the sentinel key below.
            if (pd.lane_stride != 0) {  // wave-uniform
              // Regular columns (every sample of a column is `lane_stride` bytes behind the one before: any calendar
              // without ragged or padded columns -- the plan checks): ONE offset per lane from the table, the sample
              // index goes into the load's scalar base.  100 table loads, their wait and 100 vector adds less per task.
              int off0 = ld_saddr(tl, lane_b);
              wait_vm0();
              landed(off0);
              const char *base = xc;
#pragma unroll
              for (int i = 0; i < N; ++i) {
                v[k][i] = ld_saddr(base, uint32_t(off0));
                // rows past S repeat sample S - 1 (they are overwritten with the sentinel key below)
                if (i < lane_first_pad_slot(N) - 1 || i + 1 < S_rt) base += pd.lane_stride;
                asm volatile("" : "+s"(base));  // a running scalar base, not N loop-invariant pairs
              }
            } else {
              uint32_t toff = lane_b;
#pragma unroll
              for (int i = 0; i < N; ++i) {
                v[k][i] = ld_saddr(tl, toff);
                toff += pitch4;
                asm volatile("" : "+v"(toff));  // one running offset: the N row offsets are loop invariants the compiler
                                                 // would otherwise keep in N registers across the whole item loop
              }
              wait_vm0();
#pragma unroll
              for (int i = 0; i < N; ++i) ld_saddr_inplace(xc, v[k][i]);
            }
            wait_vm0();
#pragma unroll
            for (int i = 0; i < N; ++i) landed(v[k][i]);
            if (clocked && k == 0) c1 = __builtin_readcyclecounter();
            // census of NaN / +-inf (numba's special cases), only when one is present in the wave
            // (exponent all ones <=> bits << 1 >= 0xff000000: one v_max3_u32 per two samples)
            uint32_t emax = 0;
#pragma unroll
            for (int i = 0; i + 1 < N; i += 2) {
              const uint32_t a = uint32_t(v[k][i]), b = uint32_t(v[k][i + 1]);
              emax = max(emax, max(a + a, b + b));
            }
            if (N & 1) emax = max(emax, uint32_t(v[k][N - 1]) * 2u);
            // (slots past S hold a copy of the last sample here)
            const bool special = emax >= 0xff000000u;
            uint32_t cnt = 0;  // nan << 20 | +inf << 10 | -inf
            if (__ballot(special) != 0) {
#pragma unroll
              for (int i = 0; i < N; ++i) {
                const bool real = i < lane_first_pad_slot(N) || i < S_rt;
                float a = __int_as_float(v[k][i]);
                if (real && a != a) { cnt += 1u << 20; a = 0.0f; }
                if (real && a == INFINITY) cnt += 1u << 10;
                if (real && a == -INFINITY) cnt += 1u;
                v[k][i] = __float_as_int(a);
              }
            }
            if (lc < ncols)
              flags_p[lc] = ((cnt >> 20) ? 0x80000000u : 0u) | (((cnt >> 10) & 0x3ffu) << 15) | (cnt & 0x3ffu);
            // the image holds the samples themselves; slots past S become -inf and sort to the tail (wave-uniform
            // select; they are written onto the trailing sentinel's slot, which is stored last)
#pragma unroll
            for (int i = lane_first_pad_slot(N); i < N; ++i) v[k][i] = (i >= S_rt) ? int(0xff800000u) : v[k][i];
#if !(defined(HDP_LANE_ABL) && (HDP_LANE_ABL & 2))
            sort_lane_desc<N>(v[k]);
#endif
          }
        }
      }
      if (clocked || clocked_w) c2 = __builtin_readcyclecounter();
      if (clocked_w) atomicAdd(&pd.clk[8 + rank], c2 - c0);
      __syncthreads();  // image free (merge s-1 done), keys of block s sorted
      if (clocked) c3 = __builtin_readcyclecounter();
      if (s < n_items) {
#pragma unroll
        for (int k = 0; k < TPW; ++k) {
          const int task = pw + k * n_prod;
          const int lc = task * 64 + lane;
          if (task < n_tasks && lc < ncols) {
            // lane stride img_pitch (odd) words: a store instruction hits 64 different banks
            float *col = colbuf + lc * pd.img_pitch + 1;
            // branch-free: slots past the column's last sample all land on its trailing sentinel
            if constexpr (!TIER) {  // whole column in LDS
#pragma unroll
              for (int i = 0; i < N; ++i) {
                // slots below the first possible padding slot always hold a sample: constant LDS offsets (a run-time
                // min() on every slot gave N loop-invariant addresses, hoisted into N registers)
                if (i < lane_first_pad_slot(N)) col[i] = __int_as_float(v[k][i]);
                else col[min(i, S_rt)] = __int_as_float(v[k][i]);
              }
              col[S_rt] = __uint_as_float(kRawMin);  // loses every descending walk
            } else {
              // tiered: the top kTierK samples to LDS, the marker behind them, the rest to the workgroup's global tail
              // [sample - kTierK][column] (a store instruction writes 64 adjacent columns); slots past S land on a
              // spare row
              const float *tbase = tail_wg + size_t(s & 1) * tail_half;  // wave-uniform
              uint32_t toff = uint32_t(lc) * 4u;                          // one running byte offset (see the loads)
              const uint32_t trow = uint32_t(pd.tail_pitch) * 4u;
              // the tail's global stores go out FIRST: their latency then runs under the LDS writes below instead of
              // after them (both sit between the item's two barriers, where the merging waves wait)
#pragma unroll
              for (int i = kTierK; i < N; ++i) {
                // slots past S (padding) all land on the spare row behind the last sample's
                const uint32_t o = (i < lane_first_pad_slot(N)) ? toff : uint32_t(lc) * 4u + uint32_t(min(i, S_rt) - kTierK) * trow;
                asm volatile("global_store_dword %0, %1, %2" ::"v"(o), "v"(v[k][i]), "s"(tbase) : "memory");
                toff += trow;
                asm volatile("" : "+v"(toff));
              }
#pragma unroll
              for (int i = 0; i < kTierK && i < N; ++i) col[i] = __int_as_float(v[k][i]);
              col[kTierK] = __uint_as_float(0x7ff00000u | uint32_t(lc));
              col[kTierK + 1] = __uint_as_float(kRawMin);  // what the heads of window slots past W read
              // the tail stores are asm (the compiler does not count them) and a merging wave may read them back from L2:
              // they must have landed before the barrier that opens the item's merge.  (Round 4 measured two alternatives,
              // both without gain: waiting after that barrier while the rare tail branch polls a counter -- the poll loop
              // re-shaped the merge step's code, +8 % -- and barriers that fence LDS only, so that nobody drains its
              // global stores at them -- 12.4 ms per 131 072 cells either way.)
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            col[-1] = __uint_as_float(kRawMax);    // loses every ascending walk
          }
        }
      }
      if (clocked) c4 = __builtin_readcyclecounter();
      __syncthreads();  // image of block s ready
      if (clocked) {  // first producer wave: loads, census + sort, wait for the merge, image write
        atomicAdd(&pd.clk[4], c1 - c0);
        atomicAdd(&pd.clk[5], c2 - c1);
        atomicAdd(&pd.clk[6], c3 - c2);
        atomicAdd(&pd.clk[7], c4 - c3);
      }
    }
  } else {
    const TgtLanes tl_top = load_tgt_lanes<true>(pd, lane), tl_bot = load_tgt_lanes<false>(pd, lane);
    // segmented plans: this wave's run and its emission table
    ThrSeg sg{};
    const int2 *sg_tgt = nullptr;
    TgtLanes tl_seg{-1, 0, 0, 0.0, 0.0, false};
    if constexpr (kCanSeg) {
      if (n_segs > 0) {
        const ThrSeg *sp = pd.segs + seg;  // wave-uniform: scalar loads
        sg.top = int(ldk64(sp)); sg.pivot_pos = int(ldk64(sp) >> 32);
        sg.limit = int(ldk64(reinterpret_cast<const char *>(sp) + 8)); sg.tgt_off = int(ldk64(reinterpret_cast<const char *>(sp) + 8) >> 32);
        sg.nt = int(ldk64(reinterpret_cast<const char *>(sp) + 16)); sg.steps = int(ldk64(reinterpret_cast<const char *>(sp) + 16) >> 32);
        sg_tgt = (sg.top ? pd.tgt_top : pd.tgt_bot) + sg.tgt_off;
        tl_seg = load_tgt_lanes_seg(pd, sg_tgt, sg.nt, lane);
      }
    }
    for (int64_t s = 0; s <= n_items; ++s) {
      const uint32_t *flags_m = flags0 + int((s + 1) & 1) * flags_pitch;  // census of block s - 1
      unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
      const bool clocked = HDP_DBG(pd, 32) && lane == 0 && rank == 0;
      const bool clocked_w = HDP_DBG(pd, 1024) && lane == 0;
      if (clocked || clocked_w) c0 = __builtin_readcyclecounter();
      if (kCanSeg && n_segs > 0) {
        if constexpr (kCanSeg) {
          // Every lane of a segmented merging wave runs a merge (lanes past the block's last row repeat that row without
          // storing): the entry ranks are reduced over all 64 lanes.
          if (s >= 1 && mrow < nm_rows * 64) {
            const int64_t cell = first_cell + (s - 1) * wg_per_blk;
            const int rr = min(mrow, nrows - 1);
            const uint16_t *cl = cl_lds + rr * kListPitch;
            double *orow = out + cell * pd.n_doy * int64_t(pd.P) + row0 + rr;  // [cell][P][n_doy]
            RowFlags rf{0, 0};
            uint32_t nan_or = 0;
#pragma unroll
            for (int j = 0; j < kHG * NG; ++j) {
              const uint32_t f = (j < pd.W) ? flags_m[cl[j]] : 0u;
              nan_or |= f;
              rf.n_pos += (f >> 15) & 0x7fff;
              rf.n_neg += f & 0x7fff;
            }
            if (nan_or >> 31) rf.n_pos = -1;
            unsigned char *st = strips + size_t(seg) * seg_strip_bytes;
            if (sg.top)
              merge_row_seg<true, NG, ROWS>(pd, sg, sg_tgt, reinterpret_cast<const unsigned char *>(colbuf), st, cl, mrow, rf, orow, mrow < nrows, tl_seg);
            else
              merge_row_seg<false, NG, ROWS>(pd, sg, sg_tgt, reinterpret_cast<const unsigned char *>(colbuf), st, cl, mrow, rf, orow, mrow < nrows, tl_seg);
          }
        }
      } else if (s >= 1 && mrow < nrows) {
        const int64_t cell = first_cell + (s - 1) * wg_per_blk;
        const int row = row0 + mrow;
        const uint16_t *cl = cl_lds + mrow * kListPitch;
        double *orow = out + cell * pd.n_doy * int64_t(pd.P) + row;  // [cell][P][n_doy]
#if !(defined(HDP_LANE_ABL) && (HDP_LANE_ABL & 1))
        // ranks >= 4 are the second merging wave of their SIMD (roles above: one merging wave per SIMD first)
          merge_both_lean<NG, TIER, ROWS>(pd, reinterpret_cast<const unsigned char *>(colbuf), strips,
                              tail_wg + size_t((s + 1) & 1) * tail_half, flags_m, cl, mrow, orow, tl_top, tl_bot,
                              n_merge > 4 ? int(rank >= 4) : -1);
#endif
      }
      if (clocked || clocked_w) c1 = __builtin_readcyclecounter();
      if (clocked_w) {
        atomicAdd(&pd.clk[8 + rank], c1 - c0);
        if (rank == 0) atomicAdd(&pd.clk[3], 1ull);
        atomicAdd(&pd.clk[20 + rank], (unsigned long long)my_simd);
      }
      __syncthreads();  // merge of block s - 1 done: image free
      if (clocked) c2 = __builtin_readcyclecounter();
      __syncthreads();  // image of block s ready
      if (clocked) {  // first merging wave: merge, wait for the producers, wait for the image
        c3 = __builtin_readcyclecounter();
        atomicAdd(&pd.clk[0], c1 - c0);
        atomicAdd(&pd.clk[1], c2 - c1);
        atomicAdd(&pd.clk[2], c3 - c2);
        atomicAdd(&pd.clk[3], 1ull);
      }
    }
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
template <int EPL, bool SELECT>
static int launch_thr_epl_sel(const ThrDev &pd, size_t lds, const float *x, int64_t n_cells, double *out,
                              hipStream_t stream) {
  auto kern = thresholds_kernel<EPL, SELECT>;
  HDP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)n_cells), dim3(kThrThreads), lds, stream, pd, x, n_cells, out);
  HDP_HIP_TRY(hipGetLastError());
  return HDP_OK;
}

template <int EPL>
static int launch_thr_epl(const ThrDev &pd, size_t lds, const float *x, int64_t n_cells, double *out,
                          hipStream_t stream) {
  return pd.select ? launch_thr_epl_sel<EPL, true>(pd, lds, x, n_cells, out, stream)
                   : launch_thr_epl_sel<EPL, false>(pd, lds, x, n_cells, out, stream);
}

// persistent grid shared by the pipelined and the lane-per-column kernel: as many workgroups as the device keeps
// resident, a multiple of n_blocks (workgroup w works on block w % n_blocks for cells w / n_blocks + k * (grid / n_blocks))
// hdp_threshold_plan_reserve: walk the launch path, allocate what a launch of that size would, launch nothing
static thread_local bool g_reserve_only = false;

template <class Kern>
static int launch_thr_persistent(Kern kern, const ThrDev &pd, size_t lds, const float *x, int64_t n_cells, double *out,
                                 int64_t grid_override, int threads, hipStream_t stream, DevBuf *tail_buf = nullptr) {
  HDP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    HDP_HIP_TRY(hipGetDevice(&dev));
    HDP_HIP_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  }
  int per_cu = 0;
  HDP_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kern),
                                                           threads, lds));
  if (per_cu < 1) per_cu = 1;
  const int64_t nb = pd.n_blocks;
  int64_t resident = int64_t(per_cu) * n_cu;
  if (grid_override > 0) resident = grid_override;
  int64_t grid = std::max<int64_t>(1, std::min<int64_t>(resident / nb, n_cells)) * nb;
  ThrDev pdl = pd;
  if (tail_buf) {  // tiered image: per-workgroup global tail, two parities of (S - K + 1) rows of tail_pitch floats
    const size_t need = size_t(grid) * 2 * size_t(std::max(pd.S - pd.tier_k, 0) + 1) * pd.tail_pitch * 4;
    if (tail_buf->bytes < need) {
      HDP_HIP_TRY(hipStreamSynchronize(stream));
      const hipError_t e = tail_buf->alloc(need);
      if (e != hipSuccess) return set_error(HDP_ENOMEM, "allocating %zu bytes of column-tail scratch failed: %s", need, hipGetErrorString(e));
    }
    pdl.tail = tail_buf->as<float>();
  }
  if (g_reserve_only) return HDP_OK;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(threads), lds, stream, pdl, x, n_cells, out);
  HDP_HIP_TRY(hipGetLastError());
  return HDP_OK;
}

template <int N>
static int launch_thr_lane(const ThrDev &pd, size_t lds, const float *x, int64_t n_cells, double *out,
                           int64_t grid_override, hipStream_t stream, DevBuf *tail_buf) {
  // merging waves + the producers the widest block needs: at S = 100 that is 2 + 3 waves, and two such workgroups
  // per CU leave every wave the 160 registers a 100-key column takes
  const int tpw = lane_tasks_per_wave<N>();
  const int n_tasks = (pd.ncols_max + 63) / 64;
  const bool whole = pd.n_blocks == 1 && pd.RP > kLeanRows;  // the plan put every row of a cell into one workgroup
  const int threads = std::min<int>(whole ? kWholeThreads : kThrThreads, 64 * (pd.n_merge + (n_tasks + tpw - 1) / tpw));
  // head groups of three: windows of up to 9 columns run with three groups, up to 15 with five (their lists are padded
  // with the pseudo column).  Whole-cell form: tiered image for columns of more than 64 samples (the plan makes every such
  // whole-cell plan tiered), whole columns in LDS up to 64.
  switch (pd.W <= 9 ? 3 : (pd.W <= 15 ? 5 : 0)) {
    case 3:
      if constexpr (N > 64) {
        if (whole) {
          HDP_REQUIRE(pd.tier_k < pd.S, HDP_EUNSUP, "whole-cell plan without a tiered image for S = %d", pd.S);
          return launch_thr_persistent(thresholds_lane_kernel<N, 3, true, kWholeRows>, pd, lds, x, n_cells, out, grid_override, threads, stream, tail_buf);
        }
      } else {
        if (whole)
          return launch_thr_persistent(thresholds_lane_kernel<N, 3, false, kWholeRows>, pd, lds, x, n_cells, out, grid_override, threads, stream, nullptr);
      }
      return launch_thr_persistent(thresholds_lane_kernel<N, 3, false, kLeanRows>, pd, lds, x, n_cells, out, grid_override, threads, stream, nullptr);
    case 5:
      if constexpr (N > 64) {
        if (whole) {
          HDP_REQUIRE(pd.tier_k < pd.S, HDP_EUNSUP, "whole-cell plan without a tiered image for S = %d", pd.S);
          return launch_thr_persistent(thresholds_lane_kernel<N, 5, true, kWholeRows>, pd, lds, x, n_cells, out, grid_override, threads, stream, tail_buf);
        }
      } else {
        if (whole)
          return launch_thr_persistent(thresholds_lane_kernel<N, 5, false, kWholeRows>, pd, lds, x, n_cells, out, grid_override, threads, stream, nullptr);
      }
      return launch_thr_persistent(thresholds_lane_kernel<N, 5, false, kLeanRows>, pd, lds, x, n_cells, out, grid_override, threads, stream, nullptr);
    default: return set_error(HDP_EUNSUP, "lane-per-column kernel: unsupported window width");
  }
}

// register slots per column the lane-per-column kernel is instantiated for (0: S too large)
static int lane_slots_for(int64_t S) {
  static const int kN[] = {16, 32, 64, 80, 100};  // register slots per column the kernel is instantiated for (80: so that
                                                      // the 100-slot kernel's first 80 slots always hold a sample -- constant addresses)
  for (int n : kN)
    if (S <= n) return n;
  return 0;
}
static int lane_tasks_per_wave_rt(int N) { return N >= 64 ? 1 : (N >= 32 ? 2 : 4); }

// Which kernel a launch of this plan runs: the plan's choice under the HDP_THR_* selectors that were in the
// environment WHEN THE PLAN WAS CREATED (tests create one plan per variant to compare them on the same input).
struct ThrVariant {
  bool lane, select;
};
static ThrVariant thr_variant(const hdp_threshold_plan *plan, int debug) {
  ThrVariant v;
  // HDP_THR_LANE=0 (or the older HDP_THR_PIPE=0) selects the one-workgroup-per-cell kernel for A/B
  const bool lane_allowed = plan->opt_pipe != 0 && !(debug & 8);
  v.lane = plan->lane && lane_allowed && plan->opt_lane != 0;
  // rank selection pays once the merge is deep (a lane per (row, rank) instead of a lane per row);
  // HDP_THR_SELECT=0/1 forces the choice for tests
  const bool wg_per_cell = !v.lane;
  v.select = wg_per_cell && (plan->W <= 16) && (plan->steps_top + plan->steps_bot >= 512);
  if (plan->opt_select >= 0) v.select = wg_per_cell && (plan->W <= 16) && plan->opt_select != 0;
  if (plan->select_only) v.select = true;  // the plan's LDS image has no room for merge heads
  return v;
}

extern "C" const char *hdp_threshold_plan_describe(const hdp_threshold_plan *plan) {
  static thread_local char buf[256];
  if (!plan) return "";
  const ThrVariant v = thr_variant(plan, 0);
  if (v.lane)
    snprintf(buf, sizeof buf,
             "thresholds_lane_kernel<N=%d,NG=%d%s> (one lane per column: register merge-exchange sort; %d merging waves; "
             "%d rows x %d blocks, %zu B LDS%s)",
             plan->lane_n, plan->W <= 9 ? 3 : 5,
             plan->lane_tier_k < plan->S ? ",tiered,whole-cell" : (plan->n_blocks == 1 && plan->RP > hdp::kLeanRows ? ",whole-cell" : (plan->lane_n_segs > 0 ? ",segmented" : "")), plan->lane_n_merge,
             plan->rows_per_block, plan->n_blocks, plan->lane_lds_bytes,
             plan->lane_tier_k < plan->S ? "; top 60 samples of a column in LDS, the rest in a global tail" : "");
  else
    snprintf(buf, sizeof buf,
             "thresholds_kernel<EPL=%d,%s> (one workgroup per cell: LDS columns, wave sort, %s; <= %d rows x %d blocks%s, "
             "%zu B LDS)",
             plan->epl, v.select ? "select" : "merge", v.select ? "rank selection per (row, rank)" : "W-way merge per row",
             plan->rows_per_block, plan->n_blocks, plan->select_only ? " over a ring of resident columns" : "",
             plan->lds_bytes);
  return buf;
}

int launch_thresholds(const hdp_threshold_plan *plan, const float *x_dev, int64_t n_cells,
                      double *out_dev, hipStream_t stream, int64_t x_pitch) {
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
  pd.blk_sort_off = plan->blk_sort_off.as<int32_t>();
  pd.sort_slots = plan->sort_slots.as<int32_t>();
  pd.n_doy = (int)plan->n_doy;
  pd.S = (int)plan->S;
  pd.W = (int)plan->W;
  pd.P = (int)plan->P;
  pd.T = (int)plan->T;
  pd.xp = x_pitch > 0 ? x_pitch : plan->T;
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
  pd.n_merge = plan->n_merge;
  pd.n_segs = 0;
  pd.segs = nullptr;
  pd.lane_stride = 0;
  pd.tixl = plan->tixl.as<int32_t>();
  pd.blk_tixl_off = plan->blk_tixl_off.as<int32_t>();
  pd.tier_k = plan->lane_tier_k;
  pd.img_pitch = plan->lane_img_pitch;
  pd.tail_pitch = ((plan->ncols_max + 63) / 64) * 64;
  pd.tail = nullptr;
  pd.grid_override = plan->opt_grid;
#ifdef HDP_DEBUG_ABLATIONS
  pd.debug = getenv("HDP_THR_DEBUG") ? atoi(getenv("HDP_THR_DEBUG")) : 0;
#else
  pd.debug = 0;
#endif
  pd.clk = nullptr;
  if (pd.debug & (8 | 32 | 1024 | 2048)) {
    if (plan->clk.bytes == 0) {
      HDP_HIP_TRY(plan->clk.alloc(32 * sizeof(unsigned long long)));
      HDP_HIP_TRY(hipMemset(plan->clk.p, 0, 32 * sizeof(unsigned long long)));
    }
    pd.clk = plan->clk.as<unsigned long long>();
  }
  const ThrVariant var = thr_variant(plan, pd.debug);
  pd.select = var.select;
  // hdp_threshold_plan_reserve walks this path with sentinel pointers: only the lane kernel's launcher owns scratch (and
  // checks g_reserve_only); any other variant -- e.g. a debug build that turned the lane kernel off -- has nothing to reserve
  if (g_reserve_only && !var.lane) return HDP_OK;
  if (var.lane) {
    pd.n_merge = plan->lane_n_merge;
    pd.n_segs = plan->lane_n_segs;
    pd.lane_stride = plan->lane_stride;
    pd.segs = plan->lane_segs.as<hdp::ThrSeg>();
    switch (plan->lane_n) {
      case 16: return launch_thr_lane<16>(pd, plan->lane_lds_bytes, x_dev, n_cells, out_dev, plan->opt_grid, stream, &plan->lane_tail);
      case 32: return launch_thr_lane<32>(pd, plan->lane_lds_bytes, x_dev, n_cells, out_dev, plan->opt_grid, stream, &plan->lane_tail);
      case 64: return launch_thr_lane<64>(pd, plan->lane_lds_bytes, x_dev, n_cells, out_dev, plan->opt_grid, stream, &plan->lane_tail);
      case 80: return launch_thr_lane<80>(pd, plan->lane_lds_bytes, x_dev, n_cells, out_dev, plan->opt_grid, stream, &plan->lane_tail);
      case 100: return launch_thr_lane<100>(pd, plan->lane_lds_bytes, x_dev, n_cells, out_dev, plan->opt_grid, stream, &plan->lane_tail);
      default: return set_error(HDP_EUNSUP, "lane-per-column kernel: no instantiation for %d slots", plan->lane_n);
    }
  }
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

// Time-major input [T][pitch] (CMIP order): chunks of cells are transposed into series-major staging buffers on the
// plan's copy stream while the kernel works on the previous chunk (the thresholds kernels are latency-bound and leave
// HBM mostly idle, so the copy hides behind them).  Two staging chunks, fork/join by events on the caller's stream.
int launch_thresholds_tm(const hdp_threshold_plan *plan, const float *x_tm_dev, int64_t pitch, int64_t n_cells,
                         double *out_dev, hipStream_t stream) {
  if (n_cells == 0) return HDP_OK;
  const int64_t T = plan->T;
  static const long long chunk_env = env_option("HDP_TM_CHUNK_CELLS", 0);  // A/B only
  const int64_t chunk = std::min<int64_t>(n_cells, chunk_env > 0 ? chunk_env : std::max<int64_t>(256, std::min<int64_t>(8192, (int64_t(5) << 30) / (T * 4))));
  if (!plan->tm_stream) {
    // The copy stream gets the LOWEST priority (round 4; it had the highest): a kernel-trace of the pipeline
    // (tools/dbg/tm_time.py) shows a chunk's kernel and the next chunk's copy released at the same moment, the copy's
    // workgroups filling every CU first, and the kernel taking its own time PLUS the copy's (1.24 against 0.69 ms per 8 192
    // cells next to a 0.95 ms copy) -- time-slicing, not overlap.  With the kernel's workgroups placed first and chunks of
    // 8 192 cells the pass is 9.6 instead of 10.3 ms per 65 536 cells (series-major 5.1); a copy in turns on one stream
    // costs the same 10.3.
    int prio_lo = 0, prio_hi = 0;
    HDP_HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    // events first, the stream last: `tm_stream != nullptr` then means "everything exists" -- a failure half way leaves
    // tm_stream null, the next call starts over (an event that already exists is kept), the destructor frees the rest
    if (!plan->tm_fork) HDP_HIP_TRY(hipEventCreateWithFlags(&plan->tm_fork, hipEventDisableTiming));
    for (int i = 0; i < 2; ++i) {
      if (!plan->tm_copied[i]) HDP_HIP_TRY(hipEventCreateWithFlags(&plan->tm_copied[i], hipEventDisableTiming));
      if (!plan->tm_used[i]) HDP_HIP_TRY(hipEventCreateWithFlags(&plan->tm_used[i], hipEventDisableTiming));
    }
    HDP_HIP_TRY(hipStreamCreateWithPriority(&plan->tm_stream, hipStreamNonBlocking, prio_lo));
  }
  // staging rows are padded to a multiple of 128 bytes: with T * 4 = 146 000 (80 past a line) every 128-byte store of the
  // copy straddled two lines and the copy ran at 3.6 TB/s; aligned (a 96-year record happens to be) it runs at 5.0
  const int64_t Tp = (T + 31) & ~int64_t(31);
  const size_t need = 2 * size_t(chunk) * Tp * 4;
  if (plan->tm_stage.bytes < need) {
    HDP_HIP_TRY(hipStreamSynchronize(stream));
    HDP_HIP_TRY(hipStreamSynchronize(plan->tm_stream));
    const hipError_t e = plan->tm_stage.alloc(need);
    if (e != hipSuccess)
      return set_error(HDP_ENOMEM, "allocating %zu bytes of time-major staging failed: %s", need, hipGetErrorString(e));
  }
  if (g_reserve_only) {  // staging, stream and events exist now; the tiered image's tail for one chunk (nothing launches)
    if (plan->lane && plan->lane_tier_k < plan->S) return launch_thresholds(plan, plan->tm_stage.as<float>(), chunk, out_dev, stream, Tp);
    return HDP_OK;
  }
  HDP_HIP_TRY(hipEventRecord(plan->tm_fork, stream));
  HDP_HIP_TRY(hipStreamWaitEvent(plan->tm_stream, plan->tm_fork, 0));
  int64_t b = 0;
  for (int64_t c0 = 0; c0 < n_cells; c0 += chunk, ++b) {
    const int64_t nc = std::min(chunk, n_cells - c0);
    const int half = int(b & 1);
    float *stage = plan->tm_stage.as<float>() + size_t(half) * size_t(chunk) * Tp;
    if (b >= 2) HDP_HIP_TRY(hipStreamWaitEvent(plan->tm_stream, plan->tm_used[half], 0));
    int rc = launch_transpose(x_tm_dev + c0, pitch, T, nc, stage, plan->tm_stream, false, Tp);
    if (rc != HDP_OK) return rc;
    HDP_HIP_TRY(hipEventRecord(plan->tm_copied[half], plan->tm_stream));
    HDP_HIP_TRY(hipStreamWaitEvent(stream, plan->tm_copied[half], 0));
    rc = launch_thresholds(plan, stage, nc, out_dev + c0 * plan->n_doy * plan->P, stream, Tp);
    if (rc != HDP_OK) return rc;
    HDP_HIP_TRY(hipEventRecord(plan->tm_used[half], stream));
  }
  return HDP_OK;
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
  HDP_REQUIRE(P <= 65535, HDP_EUNSUP, "too many quantiles");
  HDP_REQUIRE(W <= 252, HDP_EUNSUP, "window of %lld day-of-year columns exceeds 252", (long long)W);
  for (int64_t i = 0; i < n_doy * S; ++i)
    HDP_REQUIRE(time_index[i] >= -T && time_index[i] < T, HDP_EINVAL,
                "time_index[%lld]=%lld outside [-T, T)", (long long)i, (long long)time_index[i]);
  for (int64_t i = 0; i < n_doy * W; ++i)
    HDP_REQUIRE(cols[i] >= 0 && cols[i] < n_doy, HDP_EINVAL, "cols[%lld]=%d outside [0, n_doy)",
                (long long)i, cols[i]);

  auto *pl = new hdp_threshold_plan();
  pl->n_doy = n_doy; pl->S = S; pl->W = W; pl->P = P; pl->T = T;
  pl->opt_pipe = (int32_t)hdp::env_option("HDP_THR_PIPE", -1);
  pl->opt_select = (int32_t)hdp::env_option("HDP_THR_SELECT", -1);
  pl->opt_lane = (int32_t)hdp::env_option("HDP_THR_LANE", -1);
  pl->opt_grid = hdp::env_option("HDP_THR_GRID", 0);
  const int opt_rows = (int)hdp::env_option("HDP_THR_ROWS", 0);
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
  struct Tgt { int rank, slot; };  // slot = quantile index | EmitKind << 16
  std::vector<Tgt> top, bot;
  const int64_t n = pl->n;
  struct Mid { int64_t p, klo, khi, from_top, from_bot; };
  std::vector<Mid> mids;  // the interpolated quantiles: each may be walked to from either end of the window
  for (int64_t p = 0; p < P; ++p) {
    int64_t klo, khi;
    int rc = hdp::quantile_param(q[p], n, &qp[p], &klo, &khi);
    if (rc != HDP_OK) {
      delete pl;
      return set_error(HDP_EQUANT, "Quantiles must be in the range [0, 1]");
    }
    if (qp[p].mode == hdp::Q_MAX) {
      top.push_back({0, int(p) | (hdp::E_MAX << 16)});
    } else if (qp[p].mode == hdp::Q_MIN) {
      bot.push_back({0, int(p) | (hdp::E_MIN << 16)});
    } else {
      // lower = ascending[klo], upper = ascending[khi], khi = klo + 1 (or klo at the clamp): the
      // quantile is emitted at the later of the two ranks in merge order
      mids.push_back({p, klo, khi, (n - 1 - klo) + 1, khi + 1});
    }
  }
  {
    // Which end each quantile is walked from.  The descending merge runs down to the deepest rank given to it, the
    // ascending one up to its deepest, and a row pays for both walks: the lowest quantiles go to the bottom walk, the rest
    // to the top walk, at the split that makes the SUM of the two depths smallest ("each to its nearer end" is one of the
    // candidates and wins ties; ten quantiles around the median cost 817 steps from the top alone, 682 + 682 split at 0.5).
    std::vector<size_t> order(mids.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return mids[a].klo < mids[b].klo; });
    const int64_t base_top = top.empty() ? 0 : 1, base_bot = bot.empty() ? 0 : 1;
    auto cost_of = [&](size_t n_bot) {  // the n_bot lowest quantiles from the bottom
      int64_t st = base_top, sb = base_bot;
      for (size_t i = 0; i < order.size(); ++i) {
        const Mid &m = mids[order[i]];
        if (i < n_bot) sb = std::max(sb, m.from_bot);
        else st = std::max(st, m.from_top);
      }
      return st + sb;
    };
    size_t nearer = 0;  // the split "each to its nearer end" makes (from_top <= from_bot goes to the top)
    while (nearer < order.size() && mids[order[nearer]].from_top > mids[order[nearer]].from_bot) ++nearer;
    size_t best = nearer;
    for (size_t k = 0; k <= order.size(); ++k)
      if (cost_of(k) < cost_of(best)) best = k;
    for (size_t i = 0; i < order.size(); ++i) {
      const Mid &m = mids[order[i]];
      const int same = (m.khi == m.klo) ? 1 : 0;
      if (i >= best)
        top.push_back({int(n - 1 - m.klo), int(m.p) | ((same ? hdp::E_SAME : hdp::E_TOP_PAIR) << 16)});
      else
        bot.push_back({int(m.khi), int(m.p) | ((same ? hdp::E_SAME : hdp::E_BOT_PAIR) << 16)});
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
  // Plans that can only ever run the rank selection (more than 128 samples per column, deep ranks, W <= 16, and
  // HDP_THR_SELECT not 0 when the plan is made) need no merge heads in LDS: two more columns per block at S = 1000.
  pl->select_only = S > 128 && W <= 16 && (pl->steps_top + pl->steps_bot >= 512) && pl->opt_select != 0;
  auto lds_for = [&](int rows, int ncols) -> size_t {
    const int RP = (rows + 63) & ~63;
    size_t b = (size_t(ncols) * spad * 4 + 15) & ~size_t(15);
    b += 2 * ((size_t(ncols) * 4 + 15) & ~size_t(15));  // census words (x2)
    if (!pl->select_only) {
      b += size_t(pl->Wp) * RP * 4;  // heads
      b += size_t(pl->Wp) * RP * 4;  // position | slot payloads
    }
    b += (size_t(rows) * W * 2 + 15) & ~size_t(15);  // window column lists
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
  // Whole-cell form of the lane-per-column kernel: with the tiered image (top kTierK samples of a column in LDS) all
  // n_doy columns and the heads of all n_doy rows fit one CU's LDS, so one 12-wave workgroup handles a whole cell:
  // no halo columns (every column is sorted once per cell, not once per block) and 365 instead of 244 merge chains
  // in flight per CU -- the chains are latency-bound, their number is the kernel's throughput.
  bool whole = false;
  size_t whole_lds = 0;
  {
    const int ngw0 = W <= 9 ? 3 : (W <= 15 ? 5 : 0);  // the lane kernel's head groups of three: instantiated for 3 and 5
    // tiered (more than 64 samples per column: top-side quantiles only, bottom walks would start in the global tail) or
    // with whole columns in LDS (up to 64 samples: any quantiles)
    const bool tiered = S > 64;
    // (a walk deeper than ~8 samples per column and tier slot would spend its steps fetching from the global tail)
    const bool cand = S >= 3 && S <= 100 && (!tiered || (pl->steps_bot == 0 && pl->steps_top <= 8 * hdp::kTierK)) && (ngw0 == 3 || ngw0 == 5) &&
                      pl->opt_lane != 0 && pl->opt_pipe != 0 && opt_rows <= 0 && hdp::env_option("HDP_THR_WHOLE", 1) != 0;
    if (cand && n_doy <= hdp::kWholeRows && n_doy > hdp::kLeanRows) {
      int ip = tiered ? hdp::kTierK + 3 : spad;
      if ((ip & 1) == 0) ++ip;
      size_t b = (size_t(n_doy + 1) * ip * 4 + 15) & ~size_t(15);
      b += 2 * ((size_t(n_doy) * 4 + 15) & ~size_t(15));
      b += size_t(ngw0) * hdp::kWholeRows * 16;
      b += (size_t(n_doy) * hdp::kListPitch * 2 + 15) & ~size_t(15);
      const int n_slots = hdp::lane_slots_for(S);
      const int waves = int((n_doy + 63) / 64) + (int((n_doy + 63) / 64) + hdp::lane_tasks_per_wave_rt(n_slots) - 1) / hdp::lane_tasks_per_wave_rt(n_slots);
      if (b <= kMaxLds && waves * 64 <= hdp::kWholeThreads) {
        whole = true;
        whole_lds = b;
      }
    }
  }
  int rows = whole ? (int)n_doy : opt_rows;
  if (rows <= 0) {
    // largest row count whose worst block fits `cap` bytes of LDS
    auto fit = [&](size_t cap) -> int {
      for (int r = (int)std::min<int64_t>(n_doy, hdp::kThrThreads); r >= 1; --r) {
        int cm_;
        if (max_lds_for_rows(r, &cm_) <= cap) return r;
      }
      return 0;
    };
    // two 512-thread workgroups per CU (<= 78 KB each), else whatever fits one CU; measured at C3:
    // 122-row blocks with two merging waves beat 61-row blocks (fewer halo columns per row)
    int r = fit(78 * 1024);
    if (r < std::min<int64_t>(n_doy, 16)) r = fit(kMaxLds);
    if (r > 0) {
      const int nb = int((n_doy + r - 1) / r);
      rows = int((n_doy + nb - 1) / nb);  // balance the blocks
    }
  }
  const bool rows_forced = opt_rows > 0;
  rows = (int)std::min<int64_t>(rows, std::min<int64_t>(n_doy, whole ? hdp::kWholeThreads : hdp::kThrThreads));
  int cm = 0;
  if (whole) {
    std::vector<int> all;
    cols_of_block(0, rows, all);
    cm = (int)all.size();
  } else if (rows <= 0 || max_lds_for_rows(rows, &cm) > kMaxLds) {
    delete pl;
    return set_error(HDP_EUNSUP, "window of %lld x %lld samples does not fit the 160 KiB LDS",
                     (long long)W, (long long)S);
  }
  // lane-per-column kernel: every producer wave sorts (and holds) up to lane_tasks_per_wave tasks of 64 columns
  const int ngw = W <= 9 ? 3 : (W <= 15 ? 5 : 0);  // head groups of three of the lane kernel: instantiated for 3 and 5
  // (S >= 3: the pseudo column that pads the window lists must lose from slot 1 going down AND from slot S going up)
  const int lane_n = ((ngw == 3 || ngw == 5) && S >= 3) ? hdp::lane_slots_for(S) : 0;
  auto lane_ok = [&](int r, int ncols_max) -> bool {
    if (!lane_n) return false;
    const int nm = (r + 63) / 64, np = hdp::kThrThreads / 64 - nm;
    if (np < 1) return false;
    return (ncols_max + 63) / 64 <= np * hdp::lane_tasks_per_wave_rt(lane_n);
  };
  bool lane = whole || lane_ok(rows, cm);
  if (lane_n && !lane && !rows_forced) {
    for (int r = rows - 1; r >= std::max(1, rows / 2); --r) {
      const int nb = int((n_doy + r - 1) / r);
      const int rb = int((n_doy + nb - 1) / nb);
      int c2 = 0;
      if (max_lds_for_rows(rb, &c2) <= kMaxLds && lane_ok(rb, c2)) {
        rows = rb; cm = c2; lane = true;
        break;
      }
    }
  }
  pl->lane = lane;
  pl->lane_n = lane_n;
  std::vector<hdp::ThrSeg> lane_segs_host;  // segmented blocked form: the runs chosen below
  if (lane) {
    // Tiered image: the merge is a chain of dependent steps per row, so the kernel's throughput is the number of rows
    // resident per CU over the step latency, and the rows are bounded by LDS.  With more than 64 samples per column
    // and top-side quantiles only, a column keeps its top kTierK samples in LDS (a window's deepest rank rarely
    // draws more than that from one column) and the rest in a global tail: three workgroups per CU instead of two.
    // (only the whole-cell form gains from it: three 5-wave workgroups of the blocked form do not fit a CU's
    // register file side by side, measured)
    pl->lane_tier_k = (whole && S > 64) ? hdp::kTierK : (int32_t)S;
    int ip = pl->lane_tier_k < S ? pl->lane_tier_k + 3 : spad;  // sentinel, samples, marker, sentinel
    if ((ip & 1) == 0) ++ip;
    pl->lane_img_pitch = ip;
    // the lane kernel's own LDS layout: image, census (x2), lean head strips (24 bytes per group and row, pitch
    // kLeanRows), window column lists
    size_t b = (size_t(cm + 1) * ip * 4 + 15) & ~size_t(15);
    b += 2 * ((size_t(cm) * 4 + 15) & ~size_t(15));
    b += size_t(ngw) * hdp::kLeanRows * 16;
    b += (size_t(rows) * hdp::kListPitch * 2 + 15) & ~size_t(15);
    pl->lane_lds_bytes = whole ? whole_lds : b;
    if (!whole && (b > kMaxLds || rows > hdp::kLeanRows)) pl->lane = false;
    // Segmented form of the blocked kernel (round 4; the round-3 "dual" form was its two-run case without pivots): the
    // requested ranks of each direction are cut into runs at the gaps between them; every run gets its own merging waves
    // and head strips and enters the merge at a pivot a few standard deviations short of its first rank (merge_row_seg), so
    // a row's chains are short and run side by side.  Blocks shrink until the workgroup (runs x ceil(rows / 64) merging
    // waves + producers) still has 8 waves and two workgroups share a CU.  HDP_THR_DUAL=0: never; 1: whenever it fits.
    const long long seg_opt = hdp::env_option("HDP_THR_DUAL", -1);
    if (pl->lane && !whole && !rows_forced && seg_opt != 0) {
      constexpr int kMaxSegs = 6, kSplitGap = 64, kMinEntry = 64;
      struct Run { int top; size_t first, count; };   // targets [first, first + count) of `top` or `bot`
      std::vector<Run> runs;
      auto cut = [&](const std::vector<Tgt> &v, int is_top) {
        for (size_t i = 0; i < v.size(); ++i) {
          if (i == 0 || v[i].rank - v[i - 1].rank > kSplitGap) runs.push_back({is_top, i, 1});
          else runs.back().count += 1;
        }
      };
      cut(top, 1);
      cut(bot, 0);
      auto gap_before = [&](size_t k) {  // ranks a run made of runs k - 1 and k would span (same direction), else "infinite"
        if (k == 0 || runs[k].top != runs[k - 1].top) return 1 << 30;
        const std::vector<Tgt> &v = runs[k].top ? top : bot;
        return v[runs[k].first + runs[k].count - 1].rank - v[runs[k - 1].first].rank;
      };
      while ((int)runs.size() > kMaxSegs) {  // too many runs for the workgroup: join the two neighbours that make the shortest run
        size_t best = 0;
        int bg = 1 << 30;
        for (size_t k = 1; k < runs.size(); ++k)
          if (gap_before(k) < bg) { bg = gap_before(k); best = k; }
        if (bg == (1 << 30)) break;
        runs[best - 1].count += runs[best].count;
        runs.erase(runs.begin() + best);
      }
      std::vector<hdp::ThrSeg> segs;
      bool any_pivot = false;
      for (const Run &r : runs) {
        const std::vector<Tgt> &v = r.top ? top : bot;
        hdp::ThrSeg sg{};
        sg.top = r.top;
        sg.tgt_off = (int32_t)r.first;
        sg.nt = (int32_t)r.count;
        sg.steps = v[r.first + r.count - 1].rank + 1;
        const int R1 = v[r.first].rank;
        sg.limit = R1 - 1;
        sg.pivot_pos = 0;
        if (R1 >= kMinEntry) {
          // the p-th sample of one column ranks about W * p in the window; its count C is binomial-like: leave four
          // standard deviations plus two samples per column of head room below the run's first rank
          // C = samples before the mean of the W columns' p-th samples: binomial-like count noise R1 (1 - R1 / n) plus the
          // pivot's own quantile-level noise, (W - 1) S sqrt(p (S - p) / S^3 / W) ranks
          const double pl0 = double(R1) / double(W), Sd = double(S);
          const double sd_count = std::sqrt(double(R1) * (1.0 - double(R1) / double(n)));
          const double sd_pivot = double(W - 1) * Sd * std::sqrt(std::max(pl0 * (Sd - pl0), 0.0) / (Sd * Sd * Sd) / double(W));
          const double sd = std::sqrt(sd_count * sd_count + sd_pivot * sd_pivot);
          const int pp = int((double(R1) - 3.5 * sd - double(W)) / double(W));
          if (pp >= 2) sg.pivot_pos = (int32_t)std::min<int64_t>(pp, S - 1);
        }
        any_pivot |= sg.pivot_pos > 0;
        segs.push_back(sg);
      }
      const bool both_long = pl->steps_top >= 96 && pl->steps_bot >= 96;   // the old dual condition
      if ((int)segs.size() <= kMaxSegs && !segs.empty() && (seg_opt > 0 || any_pivot || (both_long && segs.size() >= 2))) {
        const int tpw = hdp::lane_tasks_per_wave_rt(lane_n);
        const int ns = (int)segs.size();
        for (int r = std::min(rows, (int)hdp::kLeanRows); r >= 16; --r) {
          const int nb = int((n_doy + r - 1) / r);
          const int rb = int((n_doy + nb - 1) / nb);  // balanced blocks
          int c2 = 0;
          if (max_lds_for_rows(rb, &c2) > kMaxLds) continue;
          const int nm2 = ns * ((rb + 63) / 64);
          const int n_tasks = (c2 + 63) / 64;
          if (nm2 + (n_tasks + tpw - 1) / tpw > hdp::kThrThreads / 64) continue;
          size_t bd = (size_t(c2 + 1) * ip * 4 + 15) & ~size_t(15);
          bd += 2 * ((size_t(c2) * 4 + 15) & ~size_t(15));
          bd += size_t(ns) * size_t(ngw) * size_t((rb + 63) & ~63) * 16;
          bd += (size_t(rb) * hdp::kListPitch * 2 + 15) & ~size_t(15);
          if (bd > (kMaxLds + 1024) / 2 - 512 && !(ns > 2 && bd <= kMaxLds)) continue;  // two workgroups per CU (one when many runs)
          rows = rb;
          cm = c2;
          pl->lane_n_segs = ns;
          lane_segs_host = segs;
          pl->lane_lds_bytes = bd;
          break;
        }
      }
    }
  }
  pl->n_merge = (rows + 63) / 64;
  pl->lane_n_merge = pl->lane_n_segs > 0 ? pl->lane_n_segs * pl->n_merge : pl->n_merge;
  pl->rows_per_block = rows;
  pl->RP = (rows + 63) & ~63;
  pl->ncols_max = cm;
  pl->lds_bytes = lds_for(rows, cm);
  pl->n_blocks = int((n_doy + rows - 1) / rows);

  // per-block tables
  std::vector<int32_t> row0s, nrows, ncols, loff, llen, coff, cdoy, soff, sslots;
  std::vector<int2> list;
  std::vector<uint16_t> cl(size_t(n_doy) * W);
  std::vector<int> set, local(n_doy);
  if (pl->select_only) {
    // Ring schedule: the LDS image is `cm` column slots that persist across the blocks of a cell.  A block loads
    // and sorts only the columns its rows need that are not resident yet, into the slots of columns no later row
    // needs soonest (Belady); adjacent windows share all but one column, so after the first block a block of R rows
    // brings in about R columns instead of R + W - 1.
    const int C = cm, n_waves = hdp::kThrThreads / 64;
    std::vector<std::vector<int>> rows_of(n_doy);
    for (int r = 0; r < n_doy; ++r)
      for (int64_t j = 0; j < W; ++j) {
        std::vector<int> &v = rows_of[cols[int64_t(r) * W + j]];
        if (v.empty() || v.back() != r) v.push_back(r);
      }
    std::vector<int> slot_of(n_doy, -1), col_in(C, -1);
    auto next_use = [&](int c, int from_row) {
      const std::vector<int> &v = rows_of[c];
      auto it = std::lower_bound(v.begin(), v.end(), from_row);
      return it == v.end() ? (1 << 30) : *it;
    };
    auto needed_for = [&](int r0, int nr, std::vector<int> &need, std::vector<int> &fresh) {
      need.clear();
      fresh.clear();
      std::vector<char> seen(n_doy, 0);
      for (int r = r0; r < r0 + nr; ++r)
        for (int64_t j = 0; j < W; ++j) {
          const int c = cols[int64_t(r) * W + j];
          if (seen[c]) continue;
          seen[c] = 1;
          need.push_back(c);
          if (slot_of[c] < 0) fresh.push_back(c);
        }
    };
    int r0 = 0, n_steps = 0;
    std::vector<int> need, fresh;
    while (r0 < n_doy) {
      // as many rows as fit; among the three largest fitting counts prefer one whose new columns fill the sorting waves
      int nr = (int)std::min<int64_t>(rows, n_doy - r0);
      for (;; --nr) {
        needed_for(r0, nr, need, fresh);
        if ((int)need.size() <= C || nr == 1) break;
      }
      int best = nr;
      for (int cand = nr; cand >= std::max(1, nr - 2); --cand) {
        needed_for(r0, cand, need, fresh);
        if ((int)need.size() <= C && (int)fresh.size() % n_waves == 0) { best = cand; break; }
      }
      nr = best;
      needed_for(r0, nr, need, fresh);
      // slots: free ones first, then the resident columns whose next use lies farthest ahead (never one needed now)
      std::vector<char> needed_now(n_doy, 0);
      for (int c : need) needed_now[c] = 1;
      std::vector<int> victims;
      for (int sl = 0; sl < C; ++sl)
        if (col_in[sl] < 0 || !needed_now[col_in[sl]]) victims.push_back(sl);
      std::sort(victims.begin(), victims.end(), [&](int a, int b) {
        const int ua = col_in[a] < 0 ? (1 << 30) + 1 : next_use(col_in[a], r0 + nr);
        const int ub = col_in[b] < 0 ? (1 << 30) + 1 : next_use(col_in[b], r0 + nr);
        return ua != ub ? ua > ub : a < b;
      });
      soff.push_back((int32_t)sslots.size());
      const size_t start = list.size();
      for (size_t i = 0; i < fresh.size(); ++i) {
        const int sl = victims[i], c = fresh[i];
        if (col_in[sl] >= 0) slot_of[col_in[sl]] = -1;
        col_in[sl] = c;
        slot_of[c] = sl;
        sslots.push_back(sl);
        for (int64_t e2 = 0; e2 < S; ++e2) {
          int64_t t = time_index[int64_t(c) * S + e2];
          if (t < 0) t += T;  // NumPy negative indexing: -1 is the last time step
          list.push_back(make_int2((int)t, sl * spad + 1 + (int)e2));
        }
      }
      std::stable_sort(list.begin() + start, list.end(), [](const int2 &a, const int2 &b) { return a.x < b.x; });
      for (int r = r0; r < r0 + nr; ++r)
        for (int64_t j = 0; j < W; ++j) cl[size_t(r) * W + j] = (uint16_t)slot_of[cols[int64_t(r) * W + j]];
      coff.push_back((int)cdoy.size());
      row0s.push_back(r0);
      nrows.push_back(nr);
      ncols.push_back((int)fresh.size());
      loff.push_back((int)start);
      llen.push_back((int)(list.size() - start));
      r0 += nr;
      ++n_steps;
    }
    pl->n_blocks = n_steps;
  }
  for (int b = 0; !pl->select_only && b < pl->n_blocks; ++b) {
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
    coff.push_back((int)cdoy.size());
    for (size_t i = 0; i < set.size(); ++i) cdoy.push_back(set[i]);
    row0s.push_back(r0);
    nrows.push_back(nr);
    ncols.push_back((int)set.size());
    loff.push_back((int)start);
    llen.push_back((int)(list.size() - start));
    soff.push_back((int32_t)sslots.size());
    for (size_t i = 0; i < set.size(); ++i) sslots.push_back((int32_t)i);
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
  if (sslots.empty()) sslots.push_back(0);
  up(pl->blk_sort_off, soff.data(), soff.size() * 4);
  up(pl->sort_slots, sslots.data(), sslots.size() * 4);
  up(pl->qparam, qp.data(), qp.size() * sizeof(hdp::QuantileParam));
  up(pl->tgt_top, ttop.data(), ttop.size() * sizeof(int2));
  up(pl->tgt_bot, tbot.data(), tbot.size() * sizeof(int2));
  if (!lane_segs_host.empty()) up(pl->lane_segs, lane_segs_host.data(), lane_segs_host.size() * sizeof(hdp::ThrSeg));
  if (pl->lane) {
    // per block [N][64 * tasks]: BYTE offset of sample s of local column c (pad lanes repeat the last column)
    std::vector<int32_t> tl, tloff;
    for (int b = 0; b < pl->n_blocks; ++b) {
      const int nc = ncols[b], pitch = ((nc + 63) / 64) * 64;
      tloff.push_back((int32_t)tl.size());
      const size_t base = tl.size();
      tl.resize(base + size_t(lane_n) * pitch, 0);
      for (int c = 0; c < pitch; ++c) {
        const int doy = cdoy[size_t(coff[b]) + std::min(c, nc - 1)];
        for (int64_t e2 = 0; e2 < lane_n; ++e2) {  // rows past S repeat the last sample (the kernel discards them)
          int64_t t = time_index[int64_t(doy) * S + std::min<int64_t>(e2, S - 1)];
          if (t < 0) t += T;  // NumPy negative indexing: -1 is the last time step
          tl[base + size_t(e2) * pitch + c] = (int32_t)(t * 4);
        }
      }
    }
    {  // regular columns: sample e2 of every column lies e2 * stride bytes behind its sample 0 (rows past S repeat S - 1)
      int64_t stride = S >= 2 ? int64_t(tl[size_t(tl.size() ? 1 : 0) * (((ncols[0] + 63) / 64) * 64)]) - int64_t(tl[0]) : 0;
      bool reg = S >= 2 && stride > 0 && stride < (int64_t(1) << 30);
      for (int b = 0; reg && b < pl->n_blocks; ++b) {
        const int pitch = ((ncols[b] + 63) / 64) * 64;
        const int32_t *t0 = tl.data() + tloff[b];
        for (int64_t e2 = 0; reg && e2 < lane_n; ++e2)
          for (int c = 0; c < pitch; ++c)
            if (int64_t(t0[size_t(e2) * pitch + c]) != int64_t(t0[c]) + std::min<int64_t>(e2, S - 1) * stride) { reg = false; break; }
      }
      pl->lane_stride = (reg && hdp::env_option("HDP_THR_STRIDE", 1) != 0) ? (int32_t)stride : 0;
    }
    if (T >= (int64_t(1) << 29)) pl->lane = false;  // byte offsets must fit 31 bits
    up(pl->tixl, tl.data(), tl.size() * 4);
    up(pl->blk_tixl_off, tloff.data(), tloff.size() * 4);
  }
  if (e != hipSuccess) {
    delete pl;
    return set_error(HDP_EHIP, "uploading threshold plan tables failed: %s", hipGetErrorString(e));
  }
  *plan_out = pl;
  return HDP_OK;
}

extern "C" int hdp_threshold_plan_reserve(hdp_threshold_plan *plan, int64_t n_cells, int time_major) {
  HDP_REQUIRE(hdp::device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_REQUIRE(plan && n_cells >= 0, HDP_EINVAL, "bad arguments");
  if (n_cells == 0) return HDP_OK;
  // only the lane-per-column kernel's tiered image and the time-major path own scratch; the other kernels allocate nothing
  hdp::g_reserve_only = true;
  double *const no_out = reinterpret_cast<double *>(uintptr_t(16));   // never dereferenced: nothing is launched
  const float *const no_x = reinterpret_cast<const float *>(uintptr_t(16));
  int rc = HDP_OK;
  if (time_major) rc = hdp::launch_thresholds_tm(plan, no_x, n_cells, n_cells, no_out, hdp::default_stream());
  else if (plan->lane && plan->lane_tier_k < plan->S) rc = hdp::launch_thresholds(plan, no_x, n_cells, no_out, hdp::default_stream());
  hdp::g_reserve_only = false;
  return rc;
}

extern "C" int hdp_threshold_plan_destroy(hdp_threshold_plan *plan) {
  if (plan && plan->clk.bytes) {  // HDP_THR_DEBUG=8: per-phase clocks of the lead wave, summed over blocks
    unsigned long long c[32] = {0};
    if (hipMemcpy(c, plan->clk.p, sizeof c, hipMemcpyDeviceToHost) == hipSuccess && c[27])
      fprintf(stderr, "[hdp thresholds segmented] walks=%llu  mean entry rank: lowest lane %.1f, highest lane %.1f; mean limit %.1f\n",
              c[27], double(c[24]) / c[27], double(c[25]) / c[27], double(c[26]) / c[27]);
    if (c[3] && c[8]) {
      fprintf(stderr, "[hdp thresholds lane] items=%llu  busy ticks/item by wave rank (mergers first, then producers):", c[3]);
      for (int r = 0; r < 12; ++r) fprintf(stderr, " %.0f", double(c[8 + r]) / c[3]);
      fprintf(stderr, "  | mean SIMD of merging ranks:");
      for (int r = 0; r < 6; ++r) fprintf(stderr, " %.2f", double(c[20 + r]) / c[3]);
      fprintf(stderr, "\n");
    } else if (c[3] && (c[4] | c[5]))  // HDP_THR_DEBUG=32: the lane kernel's first merging wave and first producer
      fprintf(stderr,
              "[hdp thresholds lane] items=%llu  ticks/item: merge=%.0f wait_producers=%.0f wait_image=%.0f | "
              "producer: loads=%.0f census+sort=%.0f wait_merge=%.0f image_write=%.0f\n",
              c[3], double(c[0]) / c[3], double(c[1]) / c[3], double(c[2]) / c[3], double(c[4]) / c[3],
              double(c[5]) / c[3], double(c[6]) / c[3], double(c[7]) / c[3]);
    else if (c[3])
      fprintf(stderr, "[hdp thresholds] blocks=%llu  ticks/block: load=%.0f sort=%.0f merge=%.0f\n", c[3],
              double(c[0]) / c[3], double(c[1]) / c[3], double(c[2]) / c[3]);
  }
  delete plan;
  return HDP_OK;
}
