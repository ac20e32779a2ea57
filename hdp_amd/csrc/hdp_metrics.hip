// Heatwave detection and seasonal aggregation on gfx950.
//
// Replaces the Numba function compute_heatwave_metrics (reference hdp/metric.py:304-341)
// and the P x D Python loops of apply_ufunc(vectorize=True) around it (metric.py:357-366):
// for every series, percentile and definition: exceedance series -> heatwave ids ->
// HWF / HWN / HWD / HWA per season.
//
// One wavefront per (series, group of 64 (percentile, definition) pairs):
//   stage A  the wave reads 64 consecutive days per load (coalesced along time) and, for
//            each percentile, compares them with that day's threshold (staged in LDS as
//            f32 rounded toward -inf, which preserves `f32 > f64` exactly); __ballot
//            turns the 64 comparisons into one 64-bit exceedance word in LDS.  The measure
//            is read from HBM exactly once for all P x D combinations.
//   stage B  lane l owns pair (p, d) and walks its percentile's words run by run
//            (ctz on the word / its complement), applying the reference's run/gap state
//            machine (metric.py:39-58) once per hot RUN rather than per day, and
//            attributing labelled runs to seasons on the fly.  No id series is stored:
//            ids only matter through "same id as the previous labelled run in this season".
//   output   int16, four seasons packed per 8-byte store, layout [4][P][D][series][Ypitch].
//
// HBM-bound by design (no MFMA): algorithmic bytes per series =
//   4*T (measure) + 8*n_doy*P (thresholds) + 2*4*Y*P*D (metrics).
#include "hdp_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace hdp {

#ifdef HDP_DEBUG_ABLATIONS
#define HDP_MDBG(md, mask) ((md).debug & (mask))
#else
#define HDP_MDBG(md, mask) 0
#endif

struct MetDev {
  const uint16_t *doy_map;  // [T rounded up to 64]
  const int32_t *defs;      // [D][3]
  const int2 *seasons;      // [2][Y]
  int T, n_doy, D, Y, P, Ypitch, n_groups, np_max, n_doy_pad;
  int64_t xp;  // elements from one series to the next (T, or the padded pitch of the time-major staging)
  int seas_bytes, thr_bytes, wave_bytes;  // LDS carve, all multiples of 16
  int dmax;                                // max over definitions of max(min_duration, 1)
  int debug;                               // timing ablations only (HDP_METRICS_DEBUG): 1 = no stage B, 2 = no stage A
  unsigned long long *bits_g;              // split path: exceedance words [cell][P][words_pad] in HBM
  int words_pad;                           // 64-day words per (cell, percentile) row, multiple of kCW
  int n_words;                             // words of a row that carry data
  int year_words;                          // 1: year-aligned words (exceed_years_kernel), see there; 0: word w = days [64 w, 64 w + 64)
  long long out_cells, cell_off;           // series count of the whole output / offset of this launch's first series
  const int32_t *def_perm;                 // packed state machines: position in `defs` -> definition index of the output (or null)
};

constexpr int kMetWaves = 4;
constexpr int kChunkWords = 32;  // 2048 days of exceedance bits per (wave, percentile) in LDS

// largest float <= d (so that  x > d  <=>  x > result  for every float x)
__device__ __forceinline__ float f64_to_f32_down(double d) {
  float r = (float)d;  // round to nearest
  if ((double)r > d) {
    uint32_t b = __float_as_uint(r);
    if (r > 0.0f) b -= 1;
    else if (r < 0.0f) b += 1;
    else b = 0x80000001u;  // d in (-min_subnormal, 0): step below zero
    r = __uint_as_float(b);
  }
  return r;
}

struct LaneState {
  int open, s_open, e_prev;
  int in_hw, subs, id;
  int si, hwf, hwn, hwd, cur, last_id;
  unsigned long long acc_f, acc_n, acc_d, acc_a;
};

__device__ __forceinline__ void finalize_season(LaneState &st, int Y, int16_t *out_f, int16_t *out_n,
                                                int16_t *out_d, int16_t *out_a) {
  const unsigned hwa = st.hwn ? (unsigned)st.hwf / (unsigned)st.hwn : 0u;  // trunc(mean) == HWF // HWN
  const int sh = 16 * (st.si & 3);
  st.acc_f |= (unsigned long long)(st.hwf & 0xffff) << sh;
  st.acc_n |= (unsigned long long)(st.hwn & 0xffff) << sh;
  st.acc_d |= (unsigned long long)(st.hwd & 0xffff) << sh;
  st.acc_a |= (unsigned long long)(hwa & 0xffff) << sh;
  if ((st.si & 3) == 3 || st.si == Y - 1) {
    const int o = st.si & ~3;
    *reinterpret_cast<unsigned long long *>(out_f + o) = st.acc_f;
    *reinterpret_cast<unsigned long long *>(out_n + o) = st.acc_n;
    *reinterpret_cast<unsigned long long *>(out_d + o) = st.acc_d;
    *reinterpret_cast<unsigned long long *>(out_a + o) = st.acc_a;
    st.acc_f = st.acc_n = st.acc_d = st.acc_a = 0;
  }
  st.hwf = st.hwn = st.hwd = st.cur = 0;
  st.last_id = 0;
  st.si += 1;
}

// one hot run [s, e) has ended: reference state machine (metric.py:44-58) + season sums
__device__ __forceinline__ void process_run(LaneState &st, int s, int e, int min_dur, int max_subs,
                                            const int2 *seas, int Y, int16_t *out_f, int16_t *out_n,
                                            int16_t *out_d, int16_t *out_a) {
  const int len = e - s;
  bool label = false;
  if (!st.in_hw) {
    if (len >= min_dur) { st.id += 1; st.in_hw = 1; label = true; }
  } else if (st.subs < max_subs) {
    st.subs += 1;
    label = true;
  } else {
    if (len >= min_dur) { st.id += 1; label = true; }
    else st.in_hw = 0;
    st.subs = 0;
  }
  if (!label) return;
  while (st.si < Y && seas[st.si].y <= s) finalize_season(st, Y, out_f, out_n, out_d, out_a);
  while (st.si < Y) {
    const int2 ab = seas[st.si];
    if (ab.x >= e) break;
    const int lo = max(s, ab.x), hi = min(e, ab.y);
    const int days = hi - lo;
    st.hwf += days;
    if (st.id != st.last_id) { st.hwn += 1; st.cur = days; st.last_id = st.id; }
    else st.cur += days;
    st.hwd = max(st.hwd, st.cur);
    if (ab.y <= e) finalize_season(st, Y, out_f, out_n, out_d, out_a);
    else break;
  }
}

__global__ __launch_bounds__(kMetWaves * 64) void metrics_kernel_general(
    MetDev md, const float *__restrict__ x, const double *__restrict__ thr, int64_t n_thr_cells,
    const uint8_t *__restrict__ is_south, int64_t n_cells, int16_t *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  // LDS: seasons [2][Y] int2 | per wave: thr32 [np_max][n_doy_pad] f32, bits [np_max][chunk] u64
  int2 *seas_all = reinterpret_cast<int2 *>(smem);
  unsigned char *wbase = smem + md.seas_bytes + size_t(wave) * md.wave_bytes;
  float *thr32 = reinterpret_cast<float *>(wbase);
  unsigned long long *bits = reinterpret_cast<unsigned long long *>(wbase + md.thr_bytes);

  for (int i = tid; i < 2 * md.Y; i += blockDim.x) seas_all[i] = md.seasons[i];
  __syncthreads();

  const int64_t task = int64_t(blockIdx.x) * kMetWaves + wave;
  if (task >= n_cells * md.n_groups) return;  // no barriers below
  const int64_t cell = task / md.n_groups;
  const int group = int(task % md.n_groups);

  const int PD = md.P * md.D;
  const int c0 = group * 64;
  const int combo = c0 + lane;
  const bool valid = combo < PD;
  const int p_lo = c0 / md.D;
  const int p_hi = min(md.P - 1, (min(PD, c0 + 64) - 1) / md.D);
  const int np = p_hi - p_lo + 1;
  const int my_p = valid ? combo / md.D : p_lo;
  const int my_d = valid ? combo % md.D : 0;
  const int pi = my_p - p_lo;

  // thresholds of this series -> LDS, f32 rounded toward -inf
  {
    const double *tc = thr + (cell % n_thr_cells) * int64_t(md.n_doy) * md.P;
    for (int d0 = 0; d0 < md.n_doy; d0 += 64) {
      const int doy = d0 + lane;
      if (doy < md.n_doy)
        for (int q = 0; q < np; ++q)
          thr32[q * md.n_doy_pad + doy] = f64_to_f32_down(tc[int64_t(p_lo + q) * md.n_doy + doy]);
    }
  }

  const int2 *seas = seas_all + (is_south[cell] ? md.Y : 0);
  const int min_dur = md.defs[my_d * 3 + 0];
  const int max_break = md.defs[my_d * 3 + 1];
  const int max_subs = md.defs[my_d * 3 + 2];
  const int64_t row = ((int64_t(my_p) * md.D + my_d) * md.out_cells + md.cell_off + cell) * md.Ypitch;
  const int64_t plane = int64_t(md.P) * md.D * md.out_cells * md.Ypitch;
  int16_t *out_f = out + row;
  int16_t *out_n = out_f + plane;
  int16_t *out_d = out_n + plane;
  int16_t *out_a = out_d + plane;

  LaneState st;
  st.open = 0; st.s_open = 0; st.e_prev = -(1 << 30);
  st.in_hw = 0; st.subs = 0; st.id = 0;
  st.si = 0; st.hwf = st.hwn = st.hwd = st.cur = 0; st.last_id = 0;
  st.acc_f = st.acc_n = st.acc_d = st.acc_a = 0;

  const float *xc = x + cell * md.xp;
  const int n_words = md.n_words;

  for (int w0 = 0; w0 < n_words; w0 += kChunkWords) {
    const int nw = min(kChunkWords, n_words - w0);
    // ---- stage A: exceedance words for this chunk --------------------------------------
    for (int w = 0; w < nw; ++w) {
      const int t = (w0 + w) * 64 + lane;
      const bool in = t < md.T;
      const float xv = in ? xc[t] : 0.0f;
      const int dv = md.doy_map[t];  // padded to a multiple of 64
      for (int q = 0; q < np; ++q) {
        const bool hot = in && (xv > thr32[q * md.n_doy_pad + dv]);
        const unsigned long long m = __ballot(hot);
        if (lane == 0) bits[q * kChunkWords + w] = m;
      }
    }
    // same wave wrote and reads `bits`; LDS ops of one wave complete in order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- stage B: run extraction + state machine -----------------------------------------
    if (valid) {
      for (int w = 0; w < nw; ++w) {
        const unsigned long long word = bits[pi * kChunkWords + w];
        const int t0 = (w0 + w) * 64;
        int pos = 0;
        while (pos < 64) {
          if (!st.open) {
            const unsigned long long rem = word >> pos;
            if (rem == 0) break;
            pos += __ffsll((long long)rem) - 1;
            st.s_open = t0 + pos;
            st.open = 1;
            if (st.s_open - st.e_prev > max_break) st.in_hw = 0;  // metric.py:48-49
          }
          const unsigned long long remz = (~word) >> pos;
          if (remz == 0) break;  // run continues into the next word
          pos += __ffsll((long long)remz) - 1;
          const int e = t0 + pos;
          st.open = 0;
          process_run(st, st.s_open, e, min_dur, max_subs, seas, md.Y, out_f, out_n, out_d, out_a);
          st.e_prev = e;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (valid) {
    if (st.open) process_run(st, st.s_open, md.T, min_dur, max_subs, seas, md.Y, out_f, out_n, out_d, out_a);
    while (st.si < md.Y) finalize_season(st, md.Y, out_f, out_n, out_d, out_a);
  }
}

// ---- fast path: seasons far enough apart that every lane can close a season at the same time ----
//
// All lanes of a wave walk the same series, so season boundaries are wave-uniform in time.  When
// consecutive seasons of a hemisphere are at least dmax + 64 days apart (dmax = largest
// min_duration; the reference's May-Oct / Nov-Apr tables are ~210 days apart) a season can be
// finalised for ALL lanes at the first 32-day word starting >= season_end + dmax:
//   * a hot run still open there that began inside the season is already longer than every
//     min_duration, so the reference WILL label it (metric.py:44-58: long runs are labelled in
//     every branch) and its id is known (current id, +1 unless it is a sub-event): its in-season
//     days are credited now, without committing the state machine, which still runs at the run's end;
//   * every run processed before that point ends before the next season starts, every run processed
//     after it either was open (credited above) or lies beyond the season.
// Season bounds therefore live in SGPRs, the four results are packed 16 seasons per 32-byte
// sector-aligned store, and the per-run path is branch-free.
constexpr int kCW = 32;  // 64-day exceedance words per chunk: lane w of a VGPR holds word w
constexpr int kQB = 4;   // percentiles per stage-A batch
constexpr int kRow = kCW + 1;  // u64 pitch of one percentile's words in LDS (odd: rows on different banks)

struct ULane {
  int open, s_open, e_prev;
  int in_hw, subs, id;
  int hwf, hwn, hwd, cur, last_id;
};

__device__ __forceinline__ void credit(ULane &st, int days, int run_id) {
  // days > 0 of a labelled run with id run_id fall inside the current season
  const bool first = run_id != st.last_id;
  st.hwf += days;
  st.hwn += first ? 1 : 0;
  st.cur = first ? days : st.cur + days;
  st.last_id = run_id;
  st.hwd = max(st.hwd, st.cur);
}

// branch-free form of the reference state machine for one finished run [s, e)
__device__ __forceinline__ void run_closed(ULane &st, int s, int e, int min_dur, int max_subs, int sa, int sb) {
  const bool ge = (e - s) >= min_dur;
  const bool sub = st.in_hw && (st.subs < max_subs);
  const bool label = sub || ge;
  st.subs = sub ? st.subs + 1 : (st.in_hw ? 0 : st.subs);
  st.id += (ge && !sub) ? 1 : 0;
  st.in_hw = label ? 1 : 0;
  const int days = min(e, sb) - max(s, sa);
  if (label && days > 0) credit(st, days, st.id);
}

__device__ __forceinline__ void push16(uint32_t (&a)[8], uint32_t v) {
#pragma unroll
  for (int i = 0; i < 7; ++i) a[i] = __builtin_amdgcn_alignbit(a[i + 1], a[i], 16);
  a[7] = (a[7] >> 16) | (v << 16);
}

__device__ __forceinline__ void store32(int16_t *dst, const uint32_t (&a)[8]) {
  uint4 *p = reinterpret_cast<uint4 *>(dst);
  p[0] = make_uint4(a[0], a[1], a[2], a[3]);
  p[1] = make_uint4(a[4], a[5], a[6], a[7]);
}

// ---- split path, kernel 1: exceedance words of every percentile -> HBM -------------------------
// One 256-thread workgroup per series: the thresholds (f32 rounded toward -inf) are staged in LDS
// once per series and shared by the four waves, each of which converts every fourth 2048-day chunk.
// With only ~15 KB of LDS per workgroup the CU holds ~40 waves, so this streaming kernel hides HBM
// latency by occupancy; the state-machine kernel that follows then needs no thresholds at all.
// CW = 64-day words per wave and chunk (lane w of an accumulator holds word w): 32, or 16 for records of at most 64 words
// (T <= 4096), where 32-word chunks would leave two of the four waves without work.
#ifdef HDP_CROSSCHECK_KERNELS  // round-1 kernels kept as cross-checks: `make EXTRA=-DHDP_CROSSCHECK_KERNELS`
template <int CW>
__global__ __launch_bounds__(256) void exceed_kernel(MetDev md, const float *__restrict__ x,
                                                     const double *__restrict__ thr, int64_t n_thr_cells,
                                                     int64_t n_cells) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // thresholds as f32 rounded toward -inf, percentile-minor: [n_doy][PQ], PQ = P rounded up to kQB, so the kQB
  // thresholds a lane needs for one day come with ONE 16-byte LDS read (address = day row, no per-percentile add)
  float *thr32 = reinterpret_cast<float *>(smem);
  const int PQ = (md.P + kQB - 1) / kQB * kQB;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t cell = blockIdx.x;
  {
    const double *tc = thr + (cell % n_thr_cells) * int64_t(md.n_doy) * md.P;
    for (int doy = threadIdx.x; doy < md.n_doy; doy += 256)
      for (int q = 0; q < PQ; ++q)
        thr32[doy * PQ + q] = f64_to_f32_down(tc[int64_t(min(q, md.P - 1)) * md.n_doy + doy]);
  }
  __syncthreads();
  const float *xc = x + cell * md.xp;
  const int n_words = md.n_words;
  const int Tp = n_words * 64;
  unsigned long long *brow = md.bits_g + cell * md.P * int64_t(md.words_pad);
  static_assert(kQB == 4, "one float4 of thresholds per day");
  for (int w0 = wave * CW; w0 < n_words; w0 += 4 * CW) {
    float xr[CW];
    uint32_t dr[CW / 2];
#pragma unroll
    for (int w = 0; w < CW; ++w) {
      const int t = (w0 + w) * 64 + lane;
      xr[w] = (t < md.T) ? xc[t] : -INFINITY;
    }
#pragma unroll
    for (int w = 0; w < CW; w += 2) {
      const int t = (w0 + w) * 64 + lane;
      const uint32_t a = (t < Tp) ? md.doy_map[t] : 0u;
      const uint32_t b = (t + 64 < Tp) ? md.doy_map[t + 64] : 0u;
      dr[w / 2] = a | (b << 16);
    }
    for (int q0 = 0; q0 < md.P; q0 += kQB) {
      uint32_t lo[kQB], hi[kQB];
#pragma unroll
      for (int j = 0; j < kQB; ++j) lo[j] = hi[j] = 0;
      const float4 *t4 = reinterpret_cast<const float4 *>(thr32 + q0);  // row pitch PQ / 4 float4s
      const int pq4 = PQ / 4;
      float4 tc = t4[(dr[0] & 0xffffu) * pq4], tn = tc;
#pragma unroll
      for (int w = 0; w < CW; ++w) {
        if (w + 1 < CW) {
          const int dn = ((w + 1) & 1) ? (dr[(w + 1) / 2] >> 16) : (dr[(w + 1) / 2] & 0xffffu);
          tn = t4[dn * pq4];
        }
        const float xv = xr[w];
        const float tj[kQB] = {tc.x, tc.y, tc.z, tc.w};
#pragma unroll
        for (int j = 0; j < kQB; ++j) {
          const unsigned long long m = __ballot(xv > tj[j]);
          asm volatile("s_nop 1\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
                       : "+v"(lo[j]), "+v"(hi[j])
                       : "s"((uint32_t)m), "s"((uint32_t)(m >> 32)), "i"(w));
        }
        tc = tn;
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int j = 0; j < kQB; ++j)
        if (q0 + j < md.P && lane < CW)
          brow[int64_t(q0 + j) * md.words_pad + w0 + lane] = ((unsigned long long)hi[j] << 32) | lo[j];
    }
  }
}

// ---- the same kernel with the percentiles in PAIRS --------------------------------------------------------------
// exceed_kernel handles four percentiles per pass, so P = 10 costs twelve compare + lane-write slots per word, and it
// re-derives every threshold address from a packed day-of-year index (three instructions per word and pass): 51 vector
// instructions per 64-day word.  Here a pass handles two percentiles (no dead slots for any even P; one for odd P), the
// staged thresholds are [n_doy][2 * ceil(P/2)] so a lane's pair is ONE ds_read_b64 (row pitch 40 bytes at P = 10:
// consecutive day-of-year rows fall on distinct bank pairs), and the lane's 32 LDS row addresses of a chunk sit in
// registers and simply advance by 8 bytes per pass: 2 compares + 4 lane writes per word and pass, one add per word and
// pass, i.e. 35 instead of 51 vector instructions per word at P = 10.  Same words, same scratch layout.
#endif  // HDP_CROSSCHECK_KERNELS

template <int CW>
__global__ __launch_bounds__(256) void exceed_pairs_kernel(MetDev md, const float *__restrict__ x,
                                                           const double *__restrict__ thr, int64_t n_thr_cells,
                                                           int64_t n_cells) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *thr32 = reinterpret_cast<float *>(smem);  // [n_doy][PP], f32 rounded toward -inf
  const int NP2 = (md.P + 1) >> 1, PP = 2 * NP2;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t cell = blockIdx.x;
  {
    const double *tc = thr + (cell % n_thr_cells) * int64_t(md.n_doy) * md.P;
    for (int doy = threadIdx.x; doy < md.n_doy; doy += 256)
      for (int q = 0; q < PP; ++q)
        thr32[doy * PP + q] = f64_to_f32_down(tc[int64_t(min(q, md.P - 1)) * md.n_doy + doy]);
  }
  __syncthreads();
  const float *xc = x + cell * md.xp;
  const int n_words = md.n_words;
  const int Tp = n_words * 64;
  unsigned long long *brow = md.bits_g + cell * md.P * int64_t(md.words_pad);
  const uint32_t row_bytes = uint32_t(PP) * 4u;
  for (int w0 = wave * CW; w0 < n_words; w0 += 4 * CW) {
    float xr[CW];
    uint32_t ad[CW];  // LDS byte address of this lane's threshold row for word w (advances 8 bytes per pass)
#pragma unroll
    for (int w = 0; w < CW; ++w) {
      const int t = (w0 + w) * 64 + lane;
      xr[w] = (t < md.T) ? xc[t] : -INFINITY;
      ad[w] = ((t < Tp) ? uint32_t(md.doy_map[t]) : 0u) * row_bytes;
    }
    for (int g = 0; g < NP2; ++g) {
      uint32_t lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0;
      float2 tc = *reinterpret_cast<const float2 *>(smem + ad[0]), tn = tc;
#pragma unroll
      for (int w = 0; w < CW; ++w) {
        if (w + 1 < CW) tn = *reinterpret_cast<const float2 *>(smem + ad[w + 1]);
        const float xv = xr[w];
        const unsigned long long m0 = __ballot(xv > tc.x);
        const unsigned long long m1 = __ballot(xv > tc.y);
        asm volatile("s_nop 1\n\tv_writelane_b32 %0, %4, %8\n\tv_writelane_b32 %1, %5, %8\n\t"
                     "v_writelane_b32 %2, %6, %8\n\tv_writelane_b32 %3, %7, %8"
                     : "+v"(lo0), "+v"(hi0), "+v"(lo1), "+v"(hi1)
                     : "s"((uint32_t)m0), "s"((uint32_t)(m0 >> 32)), "s"((uint32_t)m1), "s"((uint32_t)(m1 >> 32)), "i"(w));
        ad[w] += 8u;  // next pass: the next pair of this row
        tc = tn;
        __builtin_amdgcn_sched_barrier(0);
      }
      if (lane < CW) {
        brow[int64_t(2 * g) * md.words_pad + w0 + lane] = ((unsigned long long)hi0 << 32) | lo0;
        if (2 * g + 1 < md.P) brow[int64_t(2 * g + 1) * md.words_pad + w0 + lane] = ((unsigned long long)hi1 << 32) | lo1;
      }
    }
  }
}

// ---- exceedance words for REGULAR calendars (doy_map[t] == t mod n_doy, 320 < n_doy <= 384): year-aligned words ---------
// The scratch row of a (series, percentile) holds kYearSpans = 6 words per year: word 6 y + j covers the 64 days that start
// at day y n_doy + 64 j.  The last word of a year therefore runs 384 - n_doy days into the next year (19 for a 365-day
// calendar): those bits repeat the first bits of the next word, and a consumer treats only the first n_doy - 320 bits of
// it as positions a run can start or end at (metrics_kernel_cells16 does; a run that reaches them continues in the next
// word).  What this buys: lane L of span j of ANY year is day-of-year (64 j + L) mod n_doy, so the thresholds a lane ever
// needs are 6 x NP values that stay in its registers for the whole record (f32 rounded toward -inf, as in exceed_kernel),
// and a span's ballot IS a word of the scratch: no LDS, no day-of-year table, no lane collection.  One wave per series
// walks the record in time order (the next year's six loads in flight); the ballots of two adjacent spans leave with one
// s_store_dwordx4 through the scalar data cache (s_dcache_wb before the wave ends; tools/ubench/sstore.hip: correct,
// 1.3 ns per ballot and CU), the zero words up to the row pitch the same way -- a row never mixes scalar and vector
// writes.  Vector instructions per series and year: 6 loads + 6 NP compares (exceed_pairs_kernel: ~340 at NP = 10), which
// is what the state-machine kernel running beside this one is short of.
typedef unsigned long long hdp_u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void sstore128(unsigned long long *base, uint32_t byte_off, hdp_u64x2 v) {
  asm volatile("s_store_dwordx4 %0, %1, %2" ::"s"(v), "s"(base), "s"(byte_off) : "memory");
}

constexpr int kYearSpans = 6;  // words (spans of 64 days) per year: n_doy in (320, 384]
constexpr int kYearsPG = 10;   // percentiles per launch (6 x NP threshold registers per lane)
#ifndef HDP_YEARS_AHEAD
#define HDP_YEARS_AHEAD 5
#endif
constexpr int kYearsAhead = HDP_YEARS_AHEAD;  // years of loads in flight per wave
constexpr int kYearsMinRecord = 24;          // shorter records keep the day-aligned kernels

template <int NP>  // percentiles of this launch: [p0, p0 + NP)
__global__ __launch_bounds__(64) void exceed_years_kernel(MetDev md, const float *__restrict__ x,
                                                          const double *__restrict__ thr, int64_t n_thr_cells,
                                                          int64_t n_cells, int p0) {
  constexpr int NJ = kYearSpans;
  static_assert(NJ % 2 == 0, "spans leave in pairs");
  const int lane = threadIdx.x;
  const int64_t cell = blockIdx.x;
  const int n_doy = md.n_doy, T = md.T;
  float th[NJ][NP];
  {
    const double *tc = thr + (cell % n_thr_cells) * int64_t(n_doy) * md.P + int64_t(p0) * n_doy;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      int doy = 64 * j + lane;
      if (doy >= n_doy) doy -= n_doy;  // the last span runs into the next year
#pragma unroll
      for (int q = 0; q < NP; ++q) th[j][q] = f64_to_f32_down(tc[int64_t(q) * n_doy + doy]);
    }
  }
  const float *xc = x + cell * md.xp;
  unsigned long long *rows[NP];  // wave-uniform row bases
#pragma unroll
  for (int q = 0; q < NP; ++q) rows[q] = md.bits_g + (cell * md.P + p0 + q) * int64_t(md.words_pad);
  const int n_years = md.n_words / NJ;
  auto load_year = [&](int y, float (&buf)[NJ]) {  // a year whose six spans lie inside the record
    const float *xy = xc + y * n_doy + lane;
#pragma unroll
    for (int j = 0; j < NJ; ++j) buf[j] = xy[64 * j];
  };
  uint32_t woff = 0;  // byte offset of the year's first word in a row
  auto year_words = [&](const float (&v)[NJ]) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {  // row by row: the year's six words of a row are 48 adjacent bytes
#pragma unroll
      for (int j = 0; j < NJ; j += 2) {
        hdp_u64x2 m;
        m.x = __ballot(v[j] > th[j][q]);
        m.y = __ballot(v[j + 1] > th[j + 1][q]);
        sstore128(rows[q], woff + uint32_t(j) * 8u, m);
      }
    }
    woff += uint32_t(NJ) * 8u;
  };
  // kYearsAhead + 1 register sets, as many years per trip: the loads of the next kYearsAhead years are in flight while a
  // year is compared (few resident waves with deep queues: the state machines beside this kernel want the registers)
  constexpr int NB = kYearsAhead + 1;
  // leading years whose last span (384 days from the year's start) ends inside the record: unmasked loads
  const int n_whole = min(n_years, T >= 64 * NJ ? (T - 64 * NJ) / n_doy + 1 : 0);
  float yb[NB][NJ];
#pragma unroll
  for (int k = 0; k < kYearsAhead; ++k)
    if (k < n_whole) load_year(k, yb[k]);
  int y = 0;
  // steady state: every year of a trip issues the loads of the year kYearsAhead later, unconditionally (a conditional
  // load would make the compiler wait for ALL loads before each compare: it assumes the path with nothing in flight)
  for (; y + NB - 1 + kYearsAhead < n_whole; y += NB) {
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      load_year(y + k + kYearsAhead, yb[(k + kYearsAhead) % NB]);
      year_words(yb[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < NB + kYearsAhead; ++k) {  // drain: at most NB - 1 + kYearsAhead years
    if (y + k < n_whole) {
      if (y + k + kYearsAhead < n_whole) load_year(y + k + kYearsAhead, yb[(k + kYearsAhead) % NB]);
      year_words(yb[k % NB]);
    }
  }
  for (y = n_whole; y < n_years; ++y) {  // the record ends inside these (at most two) years
    float v[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int t = y * n_doy + 64 * j + lane;
      v[j] = (t < T) ? xc[t] : -INFINITY;  // past the record: no exceedance (metric.py:27 pads with zeros)
    }
    year_words(v);
  }
  // zero words up to the row pitch (the state machines look one word ahead); n_words and words_pad are even
  const uint32_t end_off = uint32_t(md.words_pad) * 8u;
  hdp_u64x2 zero;
  zero.x = 0ull;
  zero.y = 0ull;
  for (uint32_t o = woff; o < end_off; o += 16u) {
#pragma unroll
    for (int q = 0; q < NP; ++q) sstore128(rows[q], o, zero);
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
}

#ifdef HDP_CROSSCHECK_KERNELS
template <bool SPLIT>
__global__ __launch_bounds__(kMetWaves * 64, 6) void metrics_kernel_uniform(
    MetDev md, const float *__restrict__ x, const double *__restrict__ thr, int64_t n_thr_cells,
    const uint8_t *__restrict__ is_south, int64_t n_cells, int16_t *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform

  // LDS per wave: thr32 [np_max][n_doy_pad] f32 | bits [np_max][kCW] u64
  unsigned char *wbase = smem + size_t(wave) * md.wave_bytes;
  float *thr32 = reinterpret_cast<float *>(wbase);
  unsigned long long *bits64 = reinterpret_cast<unsigned long long *>(wbase + md.thr_bytes);
  const uint32_t *bits32 = reinterpret_cast<const uint32_t *>(bits64);

  const int64_t task = int64_t(blockIdx.x) * kMetWaves + wave;
  if (task >= n_cells * md.n_groups) return;  // no workgroup barriers in this kernel
  const int64_t cell = task / md.n_groups;
  const int group = int(task % md.n_groups);

  const int PD = md.P * md.D;
  const int c0 = group * 64;
  const int combo = c0 + lane;
  const bool valid = combo < PD;
  const int p_lo = c0 / md.D;
  const int p_hi = min(md.P - 1, (min(PD, c0 + 64) - 1) / md.D);
  const int np = p_hi - p_lo + 1;
  const int my_p = valid ? combo / md.D : p_lo;
  const int my_d = valid ? combo % md.D : 0;
  const int pi = my_p - p_lo;

  if constexpr (!SPLIT) {  // thresholds of this series -> LDS as f32 rounded toward -inf
    const double *tc = thr + (cell % n_thr_cells) * int64_t(md.n_doy) * md.P;
    for (int d0 = 0; d0 < md.n_doy; d0 += 64) {
      const int doy = d0 + lane;
      if (doy < md.n_doy)
        for (int q = 0; q < np; ++q)
          thr32[q * md.n_doy_pad + doy] = f64_to_f32_down(tc[int64_t(p_lo + q) * md.n_doy + doy]);
    }
  }

  const int Y = md.Y;
  const int hemi = __builtin_amdgcn_readfirstlane((int)is_south[cell]);
  const int2 *seas = md.seasons + (hemi ? Y : 0);  // wave-uniform reads
  int si = 0;
  int sa = 0x7fffffff - 1024, sb = 0x7fffffff - 1024;
  if (Y > 0) {
    sa = __builtin_amdgcn_readfirstlane(seas[0].x);
    sb = __builtin_amdgcn_readfirstlane(seas[0].y);
  }

  const int min_dur = md.defs[my_d * 3 + 0];
  const int max_break = md.defs[my_d * 3 + 1];
  const int max_subs = md.defs[my_d * 3 + 2];
  const int dmax = md.dmax;
  const int64_t row = ((int64_t(my_p) * md.D + my_d) * md.out_cells + md.cell_off + cell) * md.Ypitch;
  const int64_t plane = int64_t(md.P) * md.D * md.out_cells * md.Ypitch;
  int16_t *orow = out + row;

  ULane st;
  st.open = 0; st.s_open = 0; st.e_prev = -(1 << 30);
  st.in_hw = 0; st.subs = 0; st.id = 0;
  st.hwf = st.hwn = st.hwd = st.cur = 0; st.last_id = 0;
  uint32_t af[8], an[8], ad[8], aa[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) af[i] = an[i] = ad[i] = aa[i] = 0;

  // close season `si` (wave-uniform) for every lane
#define HDP_FINALIZE(CREDIT_OPEN)                                                                   \
  do {                                                                                              \
    if ((CREDIT_OPEN) && st.open && st.s_open < sb) {                                               \
      const bool sub_ = st.in_hw && (st.subs < max_subs);                                           \
      credit(st, sb - max(st.s_open, sa), st.id + (sub_ ? 0 : 1));                                  \
    }                                                                                               \
    const uint32_t hwa_ = st.hwn ? (uint32_t)st.hwf / (uint32_t)st.hwn : 0u; /* == HWF // HWN */    \
    push16(af, (uint32_t)st.hwf & 0xffffu);                                                         \
    push16(an, (uint32_t)st.hwn & 0xffffu);                                                         \
    push16(ad, (uint32_t)st.hwd & 0xffffu);                                                         \
    push16(aa, hwa_ & 0xffffu);                                                                     \
    if ((si & 15) == 15 || si == Y - 1) {                                                           \
      for (int k_ = (si & 15); k_ < 15; ++k_) { /* partial last group: shift the tail down */       \
        push16(af, 0); push16(an, 0); push16(ad, 0); push16(aa, 0);                                 \
      }                                                                                             \
      if (valid) {                                                                                  \
        int16_t *o_ = orow + (si & ~15);                                                            \
        store32(o_, af);                                                                            \
        store32(o_ + plane, an);                                                                    \
        store32(o_ + 2 * plane, ad);                                                                \
        store32(o_ + 3 * plane, aa);                                                                \
      }                                                                                             \
    }                                                                                               \
    st.hwf = st.hwn = st.hwd = st.cur = 0;                                                          \
    st.last_id = 0;                                                                                 \
    si += 1;                                                                                        \
    if (si < Y) {                                                                                   \
      sa = __builtin_amdgcn_readfirstlane(seas[si].x);                                              \
      sb = __builtin_amdgcn_readfirstlane(seas[si].y);                                              \
    } else {                                                                                        \
      sa = sb = 0x7fffffff - 1024;                                                                  \
    }                                                                                               \
  } while (0)

  const float *xc = x + cell * md.xp;
  const int n_words = md.n_words;

  // run-skip shortcut: usable for 2 <= min_duration <= 32 (the look-ahead is one 32-day word)
  const int my_skip = (min_dur >= 2 && min_dur <= 32) ? min_dur : 1;
  const int skip_max = min(dmax, 32);

  // The chunk's measure values (and their threshold rows) live in registers: all kCW loads of a
  // chunk are issued back to back right after stage A has consumed the previous ones, so they are
  // in flight during stage B and HBM latency is off the critical path.
  const int Tp = n_words * 64;
  const unsigned long long *brow = SPLIT ? md.bits_g + (cell * md.P + p_lo) * int64_t(md.words_pad) : nullptr;
  float xr[kCW];
  uint32_t dr[kCW / 2];  // two 16-bit threshold-row indices per register
  auto load_chunk = [&](int w0) {
#pragma unroll
    for (int w = 0; w < kCW; ++w) {
      const int t = (w0 + w) * 64 + lane;
      xr[w] = (t < md.T) ? xc[t] : -INFINITY;  // -inf is never hot
    }
#pragma unroll
    for (int w = 0; w < kCW; w += 2) {
      const int t = (w0 + w) * 64 + lane;
      const uint32_t a = (t < Tp) ? md.doy_map[t] : 0u;  // table padded to a multiple of 64
      const uint32_t b = (t + 64 < Tp) ? md.doy_map[t + 64] : 0u;
      dr[w / 2] = a | (b << 16);
    }
  };
  if constexpr (!SPLIT) load_chunk(0);

  for (int w0 = 0; w0 < n_words; w0 += kCW) {
    const int nw = min(kCW, n_words - w0);
    const bool last_chunk = (w0 + kCW >= n_words);
    if constexpr (SPLIT) {
      // exceedance words were produced by exceed_kernel: two percentile rows per 512-byte load
      for (int q = (lane >> 5); q < np; q += 2)
        bits64[q * kRow + (lane & 31)] = brow[int64_t(q) * md.words_pad + w0 + (lane & 31)];
    }
    // ---- stage A: exceedance words, kQB percentiles at a time; lane w of lo/hi = word w ----------
    for (int q0 = 0; q0 < np && !SPLIT && !HDP_MDBG(md, 2); q0 += kQB) {
      uint32_t lo[kQB], hi[kQB];
      int roff[kQB];  // LDS row of each percentile of the batch (clamped: the tail repeats the last row)
#pragma unroll
      for (int j = 0; j < kQB; ++j) {
        lo[j] = hi[j] = 0;
        roff[j] = min(q0 + j, np - 1) * md.n_doy_pad;
      }
      // software-pipelined over the 32 words: the threshold reads of word w+1 are in flight while
      // word w is compared, so LDS latency is paid once per batch, not once per word
      float tc[kQB], tn[kQB];
      {
        const float *tp = thr32 + (dr[0] & 0xffffu);
#pragma unroll
        for (int j = 0; j < kQB; ++j) tc[j] = tp[roff[j]];
      }
#pragma unroll
      for (int w = 0; w < kCW; ++w) {
        if (w + 1 < kCW) {
          const int dn = ((w + 1) & 1) ? (dr[(w + 1) / 2] >> 16) : (dr[(w + 1) / 2] & 0xffffu);
          const float *tp = thr32 + dn;
#pragma unroll
          for (int j = 0; j < kQB; ++j) tn[j] = tp[roff[j]];
        }
        const float xv = xr[w];
#pragma unroll
        for (int j = 0; j < kQB; ++j) {
          const unsigned long long m = __ballot(xv > tc[j]);
          // lane w of lo/hi keeps word w.  (No writelane builtin in this clang; the two wait
          // states are what hipcc itself pads between a VALU SGPR write and v_writelane's read.)
          asm volatile("s_nop 1\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
                       : "+v"(lo[j]), "+v"(hi[j])
                       : "s"((uint32_t)m), "s"((uint32_t)(m >> 32)), "i"(w));
        }
#pragma unroll
        for (int j = 0; j < kQB; ++j) tc[j] = tn[j];
        __builtin_amdgcn_sched_barrier(0);  // keep the ballots of different words apart (SGPR pressure)
      }
#pragma unroll
      for (int j = 0; j < kQB; ++j)
        if (q0 + j < np && lane < kCW) bits64[(q0 + j) * kRow + lane] = ((unsigned long long)hi[j] << 32) | lo[j];
    }
    if constexpr (!SPLIT) {
      if (w0 + kCW < n_words) load_chunk(w0 + kCW);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- stage B: 32-day half-words, run by run ---------------------------------------------------
    for (int hw = 0; hw < 2 * nw && !HDP_MDBG(md, 1); ++hw) {
      const int t0 = w0 * 64 + hw * 32;
      while (si < Y && sb + dmax <= t0) HDP_FINALIZE(true);  // wave-uniform
      const uint32_t word = valid ? bits32[pi * (2 * kRow) + hw] : 0u;
      // Runs shorter than min_duration are no-ops while no heatwave is active (they are neither
      // labelled nor change the state, metric.py:44-58), so in that state the scan jumps straight
      // to the next run of >= min_duration days: `longs` has bit i set iff days i..i+m-1 are all
      // hot (looking into the next word; unknown future bits count as hot, which only disables the
      // shortcut).  While a heatwave is active every run is examined.
      uint32_t nxt = 0xffffffffu;
      if (hw + 1 < 2 * nw) nxt = valid ? bits32[pi * (2 * kRow) + hw + 1] : 0u;
      else if (last_chunk) nxt = 0u;  // beyond the record: not hot
      uint32_t longs = word;
      for (int k = 1; k < skip_max; ++k)
        longs &= (k < my_skip) ? __builtin_amdgcn_alignbit(nxt, word, k) : 0xffffffffu;
      const bool work = st.open ? (word != 0xffffffffu) : ((st.in_hw ? word : longs) != 0u);
      if (__ballot(work) == 0) continue;
      int pos = 0;
      while (true) {
        if (!st.open) {
          const uint32_t r = (st.in_hw ? word : longs) >> pos;
          if (r == 0) break;
          pos += __builtin_ctz(r);
          st.s_open = t0 + pos;
          st.open = 1;
          if (st.s_open - st.e_prev > max_break) st.in_hw = 0;  // metric.py:48-49
        }
        const uint32_t rz = (~word) >> pos;
        if (rz == 0) break;  // the run continues into the next word
        pos += __builtin_ctz(rz);
        const int e = t0 + pos;
        st.open = 0;
        run_closed(st, st.s_open, e, min_dur, max_subs, sa, sb);
        st.e_prev = e;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (st.open) {  // a run reaching the end of the record closes at T (metric.py:27: zero padding)
    run_closed(st, st.s_open, md.T, min_dur, max_subs, sa, sb);
    st.open = 0;
  }
  while (si < Y) HDP_FINALIZE(false);
#undef HDP_FINALIZE
}
#endif  // HDP_CROSSCHECK_KERNELS


// ---- split path, kernel 2': lane = series, one wave = 64 series x one percentile x up to 6 definitions ----
// The six definitions of a percentile see the same runs, so a lane extracts each run of its series once
// and steps the reference state machine (metric.py:39-58) of every definition with it; a run is skipped
// only while no definition has a heatwave active and it is shorter than every min_duration (a no-op for
// all of them).  Lock step is across 64 series of one percentile instead of across the (percentile,
// definition) pairs of one series, the per-word overhead is shared by 64 series, and the results of a
// season leave as 2-byte values that are contiguous across lanes: no per-lane packing registers.
// Output = the device layout, series-minor: out [4][P][D][Y][n_total] int16.
constexpr int kSlotPitch = 80;  // bytes per lane of the word slots in LDS
struct CLane {  // per-definition state of a lane
  int in_hw, subs, id;
  int hwf, hwn, hwd, cur, last_id;
};

// At most 5 waves per SIMD (the kernel is VALU-bound and no faster with 7): the exceedance kernel of the
// next batch runs beside it and needs wave slots of its own, or it becomes the slower stage of the pipeline.
template <int DG>
__global__ __launch_bounds__(kMetWaves * 64) __attribute__((amdgpu_waves_per_eu(1, 5))) void metrics_kernel_cells(MetDev md, const uint8_t *__restrict__ is_south,
                                                                    int64_t n_cells, int16_t *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_dpass = (md.D + DG - 1) / DG;
  const int64_t n_grp = (n_cells + 63) >> 6;
  const int64_t task = int64_t(blockIdx.x) * kMetWaves + wave;
  if (task >= n_grp * md.P * n_dpass) return;  // no workgroup barriers in this kernel
  const int d0 = int(task % n_dpass) * DG;
  const int p = int((task / n_dpass) % md.P);
  const int64_t cell = (task / (int64_t(n_dpass) * md.P)) * 64 + lane;
  const bool valid = cell < n_cells;
  const int my_hemi = valid ? int(is_south[cell]) : 2;
  const unsigned long long *brow = md.bits_g + ((valid ? cell : 0) * md.P + p) * int64_t(md.words_pad);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // lane-private 64-byte slot (pitch 80 bytes: lanes spread over the banks)
  unsigned char *slot = smem + (size_t(wave) * 64 + lane) * kSlotPitch;
  uint4 *slot4 = reinterpret_cast<uint4 *>(slot);
  const unsigned long long *slot8 = reinterpret_cast<const unsigned long long *>(slot);

  // definition parameters of this pass (wave-uniform); slots past D never label and are not stored
  int min_dur[DG], max_break[DG], max_subs[DG];
  int mmin = 0x7fffffff;
#pragma unroll
  for (int k = 0; k < DG; ++k) {
    const bool real = d0 + k < md.D;
    min_dur[k] = real ? md.defs[(d0 + k) * 3 + 0] : 0x3fffffff;
    max_break[k] = real ? md.defs[(d0 + k) * 3 + 1] : 0;
    max_subs[k] = real ? md.defs[(d0 + k) * 3 + 2] : 0;
    mmin = min(mmin, max(min_dur[k], 1));
  }
  const int skip = min(mmin, 64);  // look-ahead of the run-skip shortcut is one 64-day word
  const int Y = md.Y, dmax = md.dmax;
  const int n_words = md.n_words;
  const int64_t n_total = md.out_cells;          // series of the whole output; this launch starts at md.cell_off
  const int64_t plane = int64_t(Y) * n_total;    // one (metric, percentile, definition) plane

  for (int h = 0; h < 2; ++h) {  // lanes of one hemisphere at a time: season bounds stay wave-uniform
    const bool act = my_hemi == h;
    if (__ballot(act) == 0) continue;
    const int2 *seas = md.seasons + (h ? Y : 0);
    int si = 0;
    int sa = 0x7fffffff - 1024, sb = 0x7fffffff - 1024;
    if (Y > 0) {
      sa = __builtin_amdgcn_readfirstlane(seas[0].x);
      sb = __builtin_amdgcn_readfirstlane(seas[0].y);
    }
    CLane st[DG];
#pragma unroll
    for (int k = 0; k < DG; ++k) {
      st[k].in_hw = st[k].subs = st[k].id = 0;
      st[k].hwf = st[k].hwn = st[k].hwd = st[k].cur = st[k].last_id = 0;
    }
    int open = 0, s_open = 0, e_prev = -(1 << 30);

    auto credit_k = [&](CLane &c, int days, int run_id) {
      const bool first = run_id != c.last_id;
      c.hwf += days;
      c.hwn += first ? 1 : 0;
      c.cur = first ? days : c.cur + days;
      c.last_id = run_id;
      c.hwd = max(c.hwd, c.cur);
    };
    // one finished run [s, e): reference state machine + season credit, for every definition of the pass
    // `may_credit` is wave-uniform: false for words that end before the current season starts (all lanes
    // of a pass share the season table, and 55-60 % of a year is off-season), where the season sums
    // cannot change and only the state machines are stepped
    auto close_run = [&](int s, int e, bool may_credit) {
      const int len = e - s;
      if (may_credit) {
        const int days = min(e, sb) - max(s, sa);
#pragma unroll
        for (int k = 0; k < DG; ++k) {
          CLane &c = st[k];
          const bool ge = len >= min_dur[k];
          const bool sub = c.in_hw && (c.subs < max_subs[k]);
          const bool label = sub || ge;
          c.subs = sub ? c.subs + 1 : (c.in_hw ? 0 : c.subs);
          c.id += (ge && !sub) ? 1 : 0;
          c.in_hw = label ? 1 : 0;
          if (label && days > 0) credit_k(c, days, c.id);
        }
      } else {
#pragma unroll
        for (int k = 0; k < DG; ++k) {
          CLane &c = st[k];
          const bool ge = len >= min_dur[k];
          const bool sub = c.in_hw && (c.subs < max_subs[k]);
          c.subs = sub ? c.subs + 1 : (c.in_hw ? 0 : c.subs);
          c.id += (ge && !sub) ? 1 : 0;
          c.in_hw = (sub || ge) ? 1 : 0;
        }
      }
    };
    // close season si (wave-uniform) for every lane and definition
    auto finalize = [&](bool credit_open) {
#pragma unroll
      for (int k = 0; k < DG; ++k) {
        CLane &c = st[k];
        if (credit_open && open && s_open < sb) {
          // a run still open dmax days past the season's end is labelled in every branch of the reference
          const bool sub = c.in_hw && (c.subs < max_subs[k]);
          credit_k(c, sb - max(s_open, sa), c.id + (sub ? 0 : 1));
        }
        if (act && d0 + k < md.D) {
          const unsigned hwa = c.hwn ? (unsigned)c.hwf / (unsigned)c.hwn : 0u;  // == HWF // HWN
          int16_t *o = out + ((int64_t(p) * md.D + d0 + k) * Y + si) * n_total + md.cell_off + cell;
          const int64_t mstride = int64_t(md.P) * md.D * plane;
          o[0] = (int16_t)c.hwf;
          o[mstride] = (int16_t)c.hwn;
          o[2 * mstride] = (int16_t)c.hwd;
          o[3 * mstride] = (int16_t)hwa;
        }
        c.hwf = c.hwn = c.hwd = c.cur = 0;
        c.last_id = 0;
      }
      si += 1;
      if (si < Y) {
        sa = __builtin_amdgcn_readfirstlane(seas[si].x);
        sb = __builtin_amdgcn_readfirstlane(seas[si].y);
      } else {
        sa = sb = 0x7fffffff - 1024;
      }
    };

    // The lane's words come in blocks of eight (one 64-byte line per lane and block, four 16-byte loads that
    // hit the same line): block b + 1 is in flight while block b, parked in a lane-private LDS slot, is walked
    // word by word.  Word-at-a-time loads re-fetched every line up to eight times (6.7x the bytes).
    const uint4 *bline = reinterpret_cast<const uint4 *>(brow);
    uint4 pf[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) pf[q] = act ? bline[q] : make_uint4(0, 0, 0, 0);
    for (int w = 0; w < n_words; ++w) {
      const int t0 = w * 64;
      if ((w & 7) == 0) {  // wave-uniform: park the fetched block, request the next one
#pragma unroll
        for (int q = 0; q < 4; ++q) slot4[q] = pf[q];
        const bool more = act && (w + 8 < md.words_pad);
#pragma unroll
        for (int q = 0; q < 4; ++q) pf[q] = more ? bline[(w >> 1) + 4 + q] : make_uint4(0, 0, 0, 0);
      }
      while (si < Y && sb + dmax <= t0) finalize(true);  // wave-uniform
      const unsigned long long word = slot8[w & 7];
      // the word after it: next in the slot, or the first word of the block in flight; beyond the record: not hot
      const unsigned long long nxt =
          (w & 7) != 7 ? slot8[(w & 7) + 1] : (((unsigned long long)pf[0].y << 32) | pf[0].x);
      // `longs`: bit i set iff days i..i+skip-1 are all hot (looking into the next word)
      unsigned long long longs = word;
      for (int k = 1; k < skip; ++k) longs &= (word >> k) | (nxt << (64 - k));
      int any_hw = 0;
#pragma unroll
      for (int k = 0; k < DG; ++k) any_hw |= st[k].in_hw;
      const bool work = open ? (word != ~0ull) : ((any_hw ? word : longs) != 0ull);
      if (__ballot(work) == 0) continue;
      const bool may_credit = t0 + 64 > sa;  // wave-uniform; the current season is never one already closed
      int pos = 0;  // < 64 whenever it is used as a shift
      while (true) {
        if (!open) {
          const unsigned long long r = (any_hw ? word : longs) >> pos;
          if (r == 0) break;
          pos += __builtin_ctzll(r);
          s_open = t0 + pos;
          open = 1;
#pragma unroll
          for (int k = 0; k < DG; ++k)
            if (s_open - e_prev > max_break[k]) st[k].in_hw = 0;  // metric.py:48-49
        }
        const unsigned long long rz = (~word) >> pos;
        if (rz == 0) break;  // the run continues into the next word
        pos += __builtin_ctzll(rz);
        const int e = t0 + pos;
        open = 0;
        close_run(s_open, e, may_credit);
        e_prev = e;
        any_hw = 0;
#pragma unroll
        for (int k = 0; k < DG; ++k) any_hw |= st[k].in_hw;
      }
    }
    if (open) {  // a run reaching the end of the record closes at T (metric.py:27: zero padding)
      close_run(s_open, md.T, true);
      open = 0;
    }
    while (si < Y) finalize(false);
  }
}

// ---- the same kernel with the definitions' state as packed 16-bit pairs (two definitions per register) -----------
// Eligible when every min_duration and max_break is <= 16383 and T <= 65535 (host-checked): run lengths and gaps
// are clamped to 32767 (exact against operands <= 16383), sub-event counts cannot reach 32767, ids matter only through
// equality within a season (mod 65536 is exact below 65536 heatwaves) and season sums are < 32768 by the plan's own
// check.  Predicates are 0 / 0xffff half-words made with v_pk_sub_i16 + v_pk_ashrrev_i16 and combined with and / or /
// bfi, so one instruction steps two state machines: about 7 instead of 12 vector instructions per definition and run.
typedef short pk16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk16 pk_of(uint32_t v) { return __builtin_bit_cast(pk16, v); }
__device__ __forceinline__ uint32_t pk_bits(pk16 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t pk_dup(int v) { return uint32_t(v) * 0x10001u; }       // v in [0, 65535] in both halves
__device__ __forceinline__ uint32_t pk_lt(uint32_t a, uint32_t b) {                        // halves in [0, 32767]: a < b ? 0xffff : 0
  return pk_bits((pk_of(a) - pk_of(b)) >> (short)15);
}
__device__ __forceinline__ uint32_t pk_nz(uint32_t x) {                                    // any 16-bit half: != 0 ? 0xffff : 0
  // min(x, 1) as an opaque instruction: written in C the compiler recognises "x != 0 ? -1 : 0" per half and
  // scalarises it into two compares, two selects and a permute (5 instructions instead of 2)
  uint32_t t;
  asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(t) : "v"(x));
  return pk_bits((pk16)(0 - pk_of(t)));
}
__device__ __forceinline__ uint32_t pk_neg(uint32_t a) { return pk_bits(pk_of(a) >> (short)15); }  // a < 0 ? 0xffff : 0 per half
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return pk_bits(pk_of(a) + pk_of(b)); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return pk_bits(pk_of(a) - pk_of(b)); }
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) {
  return pk_bits(__builtin_elementwise_max(pk_of(a), pk_of(b)));
}
__device__ __forceinline__ uint32_t bsel(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }

struct CPair {  // two definitions' state, one per 16-bit half
  uint32_t hw, subs;  // hw: 0xffff while a heatwave is active; subs: sub-events so far MINUS max_subs (negative: room left)
  // fresh: 0xffff while the current heatwave has not been credited in the current season yet.  The reference numbers
  // its heatwaves (id) and counts one when a credited run's id differs from the last credited one (last_id, reset with
  // the season): "differs" is exactly "a heatwave started, or the season changed, since the last credit" -- one flag
  // instead of two counters, one instruction instead of four for `first` (round 4).
  uint32_t fresh;
  uint32_t hwf, hwn, hwd, cur;
};

// waves per SIMD the register allocation aims at (min, max).  (1, 5) until the `fresh` flag freed two registers; with 65 a
// sixth wave fits: 17.06 against 17.41 ms per 131 072 series over three A/B pairs on one box ((1, 4) 17.6, (1, 7) 17.5)
#ifndef HDP_C16_WAVES
#define HDP_C16_WAVES 1, 6
#endif
template <int NP, int NS>  // NP pairs of definitions per lane, the first NS of them "simple" (see below)
__global__ __launch_bounds__(kMetWaves * 64) __attribute__((amdgpu_waves_per_eu(HDP_C16_WAVES))) void metrics_kernel_cells16(
    MetDev md, const uint8_t *__restrict__ is_south, int64_t n_cells, int16_t *__restrict__ out, int d0) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t n_grp = (n_cells + 63) >> 6;
  const int64_t task = int64_t(blockIdx.x) * kMetWaves + wave;  // (64 series, percentile); definitions [d0, d0 + DG)
  if (task >= n_grp * md.P) return;  // no workgroup barriers in this kernel
  const int p = int(task % md.P);
  const int64_t cell = (task / md.P) * 64 + lane;
  const bool valid = cell < n_cells;
  const int my_hemi = valid ? int(is_south[cell]) : 2;
  const unsigned long long *brow = md.bits_g + ((valid ? cell : 0) * md.P + p) * int64_t(md.words_pad);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *slot = smem + (size_t(wave) * 64 + lane) * kSlotPitch;
  uint4 *slot4 = reinterpret_cast<uint4 *>(slot);
  const unsigned long long *slot8 = reinterpret_cast<const unsigned long long *>(slot);

  // definition parameters of this pass, packed (wave-uniform); slots past D never label and are not stored
  uint32_t min_dur[NP], max_break[NP], max_subs[NP], neg_subs[NP];
  int mmin = 0x7fffffff;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    uint32_t a = 0, b = 0, c = 0;
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
      const int d = d0 + 2 * k + hlf;
      const bool real = d < md.D;
      const int md_ = real ? md.defs[d * 3 + 0] : 32767;
      const int mb_ = real ? md.defs[d * 3 + 1] : 0;
      const int ms_ = real ? min(md.defs[d * 3 + 2], 32767) : 0;
      a |= uint32_t(md_) << (16 * hlf);
      b |= uint32_t(mb_) << (16 * hlf);
      c |= uint32_t(max(ms_, 0)) << (16 * hlf);
      if (real) mmin = min(mmin, max(md_, 1));
    }
    min_dur[k] = a; max_break[k] = b; max_subs[k] = c;
    neg_subs[k] = pk_sub(0u, c);  // sub-events are counted from -max_subs up: room left is the sign bit
  }
  // A definition with max_break = 0 ends its heatwave at every gap (metric.py:48-49: a gap is >= 1 day), so it never has a
  // sub-event, every labelled run is a heatwave of its own and no state crosses a gap: a pair of two such definitions
  // keeps only its three season sums and costs 6 instead of 22 instructions per run.  The host orders such definitions
  // first and instantiates the kernel with NS = the number of leading pairs made of them (a compile-time property: as
  // a run-time flag per pair it cost more scalar work than it saved).
  const int skip = min(mmin, 64);  // look-ahead of the run-skip shortcut is one 64-day word
  const int Y = md.Y, dmax = md.dmax;
  const int n_words = md.n_words;
  // year-aligned words (exceed_years_kernel): the last word of a year has len5 positions, its other bits repeat the next word's
  const int len5 = md.year_words ? md.n_doy - 64 * (kYearSpans - 1) : 64;
  const int64_t n_total = md.out_cells;
  const int64_t plane = int64_t(Y) * n_total;

  for (int h = 0; h < 2; ++h) {  // lanes of one hemisphere at a time: season bounds stay wave-uniform
    const bool act = my_hemi == h;
    if (__ballot(act) == 0) continue;
    const int2 *seas = md.seasons + (h ? Y : 0);
    int si = 0;
    int sa = 0x7fffffff - 1024, sb = 0x7fffffff - 1024;
    if (Y > 0) {
      sa = __builtin_amdgcn_readfirstlane(seas[0].x);
      sb = __builtin_amdgcn_readfirstlane(seas[0].y);
    }
    CPair st[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) st[k] = CPair{0, neg_subs[k], 0xffffffffu, 0, 0, 0, 0};
    int open = 0, s_open = 0, e_prev = -(1 << 30);

    // `lab`: 0xffff in the halves whose definition labels the run; days > 0 of it fall inside the current season
    // `fresh_now`: the halves whose credited run belongs to a heatwave not yet counted in this season
    auto credit_k = [&](CPair &c, uint32_t lab, uint32_t fresh_now, int days) {
      const uint32_t dd = pk_dup(days) & lab;
      const uint32_t first = lab & fresh_now;
      c.hwf = pk_add(c.hwf, dd);
      c.hwn = pk_sub(c.hwn, first);  // first is -1 per half: += 1
      c.cur = pk_add(c.cur & ~first, dd);
      c.fresh = c.fresh & ~lab;
      c.hwd = pk_max(c.hwd, c.cur);
    };
    // one finished run [s, e): reference state machine + season credit, for every definition of the pass
    auto close_run = [&](int s, int e, bool may_credit) {
      const uint32_t len = pk_dup(min(e - s, 32767));
      if (!may_credit) {  // wave-uniform: the word ends before the current season starts -- state only, simple pairs nothing
#pragma unroll
        for (int k = NS; k < NP; ++k) {
          CPair &c = st[k];
          const uint32_t ge = ~pk_lt(len, min_dur[k]);
          const uint32_t sub = c.hw & pk_neg(c.subs);  // room for another sub-event
          c.subs = bsel(sub, pk_add(c.subs, 0x00010001u), bsel(c.hw, neg_subs[k], c.subs));
          c.fresh |= ge & ~sub;
          c.hw = sub | ge;
        }
        return;
      }
      const int days = min(e, sb) - max(s, sa);
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        CPair &c = st[k];
        const uint32_t ge = ~pk_lt(len, min_dur[k]);
        if (k < NS) {  // hw, subs, fresh, cur of such a pair are never touched (hw stays 0)
          if (days > 0) {
            const uint32_t dd = pk_dup(days) & ge;
            c.hwf = pk_add(c.hwf, dd);
            c.hwn = pk_sub(c.hwn, ge);  // ge is -1 per half: += 1
            c.hwd = pk_max(c.hwd, dd);
          }
          continue;
        }
        const uint32_t sub = c.hw & pk_neg(c.subs);  // room for another sub-event
        const uint32_t label = sub | ge;
        c.subs = bsel(sub, pk_add(c.subs, 0x00010001u), bsel(c.hw, neg_subs[k], c.subs));
        c.fresh |= ge & ~sub;  // a new heatwave starts
        c.hw = label;
        if (days > 0) credit_k(c, label, c.fresh, days);
      }
    };
    // close season si (wave-uniform) for every lane and definition
    auto finalize = [&](bool credit_open) {
      const bool pre = credit_open && open && s_open < sb;
      const int pre_days = sb - max(s_open, sa);
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        CPair &c = st[k];
        if (pre && k < NS) {
          const uint32_t dd = pk_dup(pre_days);
          c.hwf = pk_add(c.hwf, dd);
          c.hwn = pk_add(c.hwn, 0x00010001u);
          c.hwd = pk_max(c.hwd, dd);
        } else if (pre) {
          // a run still open dmax days past the season's end is labelled in every branch of the reference
          const uint32_t sub = c.hw & pk_neg(c.subs);  // room for another sub-event
          credit_k(c, 0xffffffffu, c.fresh | ~sub, pre_days);  // a heatwave of its own unless it continues as a sub-event
        }
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
          const int d = d0 + 2 * k + hlf;
          if (act && d < md.D) {
            const unsigned hwf = (c.hwf >> (16 * hlf)) & 0xffffu, hwn = (c.hwn >> (16 * hlf)) & 0xffffu;
            const unsigned hwd = (c.hwd >> (16 * hlf)) & 0xffffu;
            const unsigned hwa = hwn ? hwf / hwn : 0u;  // == HWF // HWN
            const int dout = md.def_perm ? md.def_perm[d] : d;
            int16_t *o = out + ((int64_t(p) * md.D + dout) * Y + si) * n_total + md.cell_off + cell;
            const int64_t mstride = int64_t(md.P) * md.D * plane;
            o[0] = (int16_t)hwf;
            o[mstride] = (int16_t)hwn;
            o[2 * mstride] = (int16_t)hwd;
            o[3 * mstride] = (int16_t)hwa;
          }
        }
        c.hwf = c.hwn = c.hwd = c.cur = 0;
        c.fresh = 0xffffffffu;  // the next season counts a heatwave that runs on into it again
      }
      si += 1;
      if (si < Y) {
        sa = __builtin_amdgcn_readfirstlane(seas[si].x);
        sb = __builtin_amdgcn_readfirstlane(seas[si].y);
      } else {
        sa = sb = 0x7fffffff - 1024;
      }
    };

    const uint4 *bline = reinterpret_cast<const uint4 *>(brow);
    uint4 pf[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) pf[q] = act ? bline[q] : make_uint4(0, 0, 0, 0);
    int t_next = 0, jy = 0;
    for (int w = 0; w < n_words; ++w) {
      // geometry of word w (wave-uniform): first day t0, L positions a run can start or end at
      const int t0 = t_next;
      const bool short_word = md.year_words && jy == kYearSpans - 1;
      const int L = short_word ? len5 : 64;
      t_next += L;
      jy = (jy == kYearSpans - 1) ? 0 : jy + 1;
      const unsigned long long vmask = ~0ull >> (64 - L);
      if ((w & 7) == 0) {  // wave-uniform: park the fetched block, request the next one
#pragma unroll
        for (int q = 0; q < 4; ++q) slot4[q] = pf[q];
        const bool more = act && (w + 8 < md.words_pad);
#pragma unroll
        for (int q = 0; q < 4; ++q) pf[q] = more ? bline[(w >> 1) + 4 + q] : make_uint4(0, 0, 0, 0);
      }
      while (si < Y && sb + dmax <= t0) finalize(true);  // wave-uniform
      const unsigned long long word = slot8[w & 7];
      // the day after bit 63 of a short word is bit 64 - L of the next word (bits L.. repeat its first 64 - L bits)
      const unsigned long long nxt =
          ((w & 7) != 7 ? slot8[(w & 7) + 1] : (((unsigned long long)pf[0].y << 32) | pf[0].x)) >> (64 - L);
      unsigned long long longs = word;
      for (int k = 1; k < skip; ++k) longs &= (word >> k) | (nxt << (64 - k));
      longs &= vmask;                                   // runs of >= skip days that START in this word
      const unsigned long long starts = word & vmask;   // hot days at positions of this word
      const unsigned long long ends = ~word & vmask;    // cool days at positions of this word
      uint32_t any_hw = 0;
#pragma unroll
      for (int k = 0; k < NP; ++k) any_hw |= st[k].hw;
      const bool work = open ? (ends != 0ull) : ((any_hw ? starts : longs) != 0ull);
      if (__ballot(work) == 0) continue;
      const bool may_credit = t0 + 64 > sa;  // wave-uniform; the current season is never one already closed
      int pos = 0;  // < 64 whenever it is used as a shift
      while (true) {
        if (!open) {
          const unsigned long long r = (any_hw ? starts : longs) >> pos;
          if (r == 0) break;
          pos += __builtin_ctzll(r);
          s_open = t0 + pos;
          open = 1;
          const uint32_t gap = pk_dup(min(s_open - e_prev, 32767));  // >= 1
#pragma unroll
          for (int k = 0; k < NP; ++k)
            if (k >= NS) st[k].hw &= ~pk_lt(max_break[k], gap);  // metric.py:48-49
        }
        const unsigned long long rz = ends >> pos;
        if (rz == 0) break;  // the run continues into the next word
        pos += __builtin_ctzll(rz);
        const int e = t0 + pos;
        open = 0;
        close_run(s_open, e, may_credit);
        e_prev = e;
        any_hw = 0;
#pragma unroll
        for (int k = 0; k < NP; ++k) any_hw |= st[k].hw;
      }
    }
    if (open) {  // a run reaching the end of the record closes at T (metric.py:27: zero padding)
      close_run(s_open, md.T, true);
      open = 0;
    }
    while (si < Y) finalize(false);
  }
}


// Row layout of the (percentile, definition)-per-lane kernels, [planes][nc][Ypitch], -> device layout
// [planes][Y][n_total] at series offset cell_off.  One workgroup per (plane, 64 series), through LDS.
__global__ __launch_bounds__(256) void metrics_rows_to_cells_kernel(const int16_t *__restrict__ src, int Y, int64_t nc,
                                                                   int Ypitch, int16_t *__restrict__ dst,
                                                                   int64_t n_total, int64_t cell_off) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int16_t *tile = reinterpret_cast<int16_t *>(smem);  // [64][Ypitch + 2]
  const int TP = Ypitch + 2;
  const int64_t c0 = int64_t(blockIdx.x) * 64;
  const int64_t pl = blockIdx.y;
  const int ncell = (int)min<int64_t>(64, nc - c0);
  const int16_t *s = src + (pl * nc + c0) * Ypitch;
  for (int i = threadIdx.x; i < ncell * Ypitch; i += 256) tile[(i / Ypitch) * TP + i % Ypitch] = s[i];
  __syncthreads();
  int16_t *d = dst + pl * int64_t(Y) * n_total + cell_off + c0;
  for (int i = threadIdx.x; i < Y * 64; i += 256) {
    const int y = i >> 6, c = i & 63;
    if (c < ncell) d[int64_t(y) * n_total + c] = tile[c * TP + y];
  }
}

// device layout [4][P][D][Y][n] -> reference block layout [P][D][n][4][Y]
__global__ void metrics_repack_kernel(const int16_t *__restrict__ src, int64_t PD, int64_t n, int64_t Y,
                                      int16_t *__restrict__ dst) {
  const int64_t total = PD * n * 4 * Y;
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < total;
       i += int64_t(gridDim.x) * blockDim.x) {
    const int64_t y = i % Y;
    const int64_t m = (i / Y) % 4;
    const int64_t c = (i / (4 * Y)) % n;
    const int64_t pd = i / (4 * Y * n);
    dst[i] = src[((m * PD + pd) * Y + y) * n + c];
  }
}

// device layout [4][P][D][Y][n] int16 -> metric-major planes [4][P][D][n][Y] int64: the four output variables of
// compute_individual_metrics (metric.py:418-431: int64, dims (percentile, definition, cells..., time)) as they
// are handed to xarray, so the host side neither widens nor regroups the result
__global__ void metrics_planes_i64_kernel(const int16_t *__restrict__ src, int64_t MPD, int64_t n, int64_t Y,
                                          long long *__restrict__ dst) {
  const int64_t total = MPD * n * Y;
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < total;
       i += int64_t(gridDim.x) * blockDim.x) {
    const int64_t y = i % Y;
    const int64_t c = (i / Y) % n;
    const int64_t mpd = i / (Y * n);
    dst[i] = (long long)src[(mpd * Y + y) * n + c];
  }
}

// The same widening on the all-gathered shards of a grid (hdp_metrics_f32_planes_i64_sharded): src is
// [world][MPD][Y][n_mem * shard] int16 -- rank r's columns m * shard + c are member m of grid cells r * shard + c --
// dst the planes [nr][n_mem * n_total][Y] int64 of rows [r0, r0 + nr), series = member * n_total + grid cell.
__global__ void metrics_planes_i64_gathered_kernel(const int16_t *__restrict__ src, int64_t MPD, int64_t Y, int64_t shard,
                                                   int64_t n_mem, int64_t n_total, int64_t r0, int64_t nr,
                                                   long long *__restrict__ dst) {
  const int64_t series = n_mem * n_total, pad = n_mem * shard;
  const int64_t total = nr * series * Y;
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < total;
       i += int64_t(gridDim.x) * blockDim.x) {
    const int64_t y = i % Y;
    const int64_t s = (i / Y) % series;
    const int64_t rr = i / (Y * series);
    const int64_t m = s / n_total, g = s % n_total;
    const int64_t r = g / shard, c = g % shard;
    dst[i] = (long long)src[((r * MPD + r0 + rr) * Y + y) * pad + m * shard + c];
  }
}

// ---- unit-level mirrors of the njit helpers --------------------------------------------------

// metric.py:280-301
__global__ void indicate_hot_days_kernel(const float *__restrict__ x, int64_t n_series, int64_t T,
                                         const double *__restrict__ thr, int64_t n_doy,
                                         const int64_t *__restrict__ doy_map, uint8_t *__restrict__ hot) {
  const int64_t total = n_series * T;
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < total;
       i += int64_t(gridDim.x) * blockDim.x) {
    const int64_t s = i / T, t = i % T;
    hot[i] = ((double)x[i] > thr[s * n_doy + doy_map[t]]) ? 1 : 0;
  }
}

// metric.py:11-60, one lane per series, streaming over hot runs
__global__ void index_heatwaves_kernel(const uint8_t *__restrict__ hot, int64_t n_series, int64_t T,
                                       int64_t min_dur, int64_t max_break, int64_t max_subs,
                                       int64_t *__restrict__ ids) {
  const int64_t s = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (s >= n_series) return;
  const uint8_t *h = hot + s * T;
  int64_t *o = ids + s * T;
  bool in_hw = false;
  int64_t subs = 0, cur = 0, e_prev = -(int64_t(1) << 40);
  int64_t t = 0;
  while (t < T) {
    if (!h[t]) { o[t] = 0; ++t; continue; }
    const int64_t start = t;
    while (t < T && h[t]) ++t;
    const int64_t len = t - start;
    if (start - e_prev > max_break) in_hw = false;
    bool label = false;
    if (!in_hw) {
      if (len >= min_dur) { ++cur; in_hw = true; label = true; }
    } else if (subs < max_subs) {
      ++subs; label = true;
    } else {
      if (len >= min_dur) { ++cur; label = true; }
      else in_hw = false;
      subs = 0;
    }
    for (int64_t u = start; u < t; ++u) o[u] = label ? cur : 0;
    e_prev = t;
  }
}

// metric.py:63-172 for arbitrary (possibly overlapping) ranges and arbitrary id values,
// one lane per (series, season).  O(len^2), unit-test sizes only.
__global__ void season_metrics_kernel(const int64_t *__restrict__ ids, int64_t n_series, int64_t T,
                                      const int64_t *__restrict__ ranges, int64_t Y,
                                      int64_t *__restrict__ out, double *__restrict__ hwa) {
  const int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= n_series * Y) return;
  const int64_t s = i / Y, y = i % Y;
  const int64_t a = ranges[2 * y], b = ranges[2 * y + 1];
  const int64_t *v = ids + s * T;
  int64_t n_unique = 0, n_nonzero_unique = 0, hwf = 0;
  int64_t vmin = 0;
  bool have = false;
  for (int64_t t = a; t < b; ++t) {
    const int64_t val = v[t];
    if (val > 0) ++hwf;  // metric.py:101 counts ids > 0
    bool first = true;
    for (int64_t u = a; u < t; ++u)
      if (v[u] == val) { first = false; break; }
    if (first) {
      ++n_unique;
      if (val != 0) ++n_nonzero_unique;
      if (!have || val < vmin) { vmin = val; have = true; }
    }
  }
  // metric.py:124-136: with two or more unique values the smallest is dropped
  const bool drop_min = n_unique >= 2;
  int64_t longest = 0, total = 0, kept = 0;
  for (int64_t t = a; t < b; ++t) {
    const int64_t val = v[t];
    bool first = true;
    for (int64_t u = a; u < t; ++u)
      if (v[u] == val) { first = false; break; }
    if (!first) continue;
    if (drop_min && val == vmin) continue;
    ++kept;
    if (val == 0) continue;  // length stays 0
    int64_t cnt = 0;
    for (int64_t u = a; u < b; ++u) cnt += (v[u] == val);
    total += cnt;
    if (cnt > longest) longest = cnt;
  }
  const double mean = kept ? (double)total / (double)kept : __longlong_as_double(0x7ff8000000000000LL);
  int64_t *o = out + s * 4 * Y;
  o[0 * Y + y] = hwf;
  o[1 * Y + y] = n_nonzero_unique;
  o[2 * Y + y] = longest;
  o[3 * Y + y] = (int64_t)mean;
  hwa[s * Y + y] = mean;
}

// ---- heat index (hdp/measure.py:61-94), element-wise, HBM-bound ------------------------------
// float64 arithmetic in the order the reference writes it (Numba types the float32 arguments
// against float64 literals), one float32 product (rel_humid*temp) as in the reference's last term.
__device__ __forceinline__ float heat_index_f(float temp, float rel_humid) {
  const double t = (double)temp, r = (double)rel_humid;
  double hi = 0.5 * (t + 61.0 + ((t - 68.0) * 1.2) + (r * 0.094));
  if (hi > 80.0) {
    hi = -42.379;
    hi += 2.04901523 * t;
    hi += 10.14333127 * r;
    hi += -0.22475541 * t * r;
    hi += -0.00683783 * (t * t);
    hi += -0.05481717 * (r * r);
    hi += 0.00122874 * (t * t) * r;
    hi += 0.00085282 * t * (r * r);
    const double rt = (double)(rel_humid * temp);  // float32 product, as typed in the reference
    hi += -0.00000199 * (rt * rt);
    if (rel_humid < 13.0f && 80.0f <= temp && temp <= 112.0f) {
      hi -= ((13.0 - r) / 4.0) * __dsqrt_rn(fabs(17.0 - fabs(t - 95.0)) / 17.0);
    } else if (rel_humid > 85.0f && 80.0f <= temp && temp <= 87.0f) {
      hi += ((r - 85.0) / 10.0) * ((87.0 - t) / 5.0);
    }
  }
  return (float)hi;
}

template <bool CELSIUS>
__global__ __launch_bounds__(256) void heat_index_kernel(const float *__restrict__ temp,
                                                         const float *__restrict__ rh, int64_t n,
                                                         float *__restrict__ out) {
  const int64_t n4 = n >> 2;
  const float4 *t4 = reinterpret_cast<const float4 *>(temp);
  const float4 *r4 = reinterpret_cast<const float4 *>(rh);
  float4 *o4 = reinterpret_cast<float4 *>(out);
  auto one = [](float t, float r) -> float {
    if (CELSIUS) t = (t * 1.8f) + 32.0f;         // celsius_to_fahrenheit, float32 (measure.py:54)
    float h = heat_index_f(t, r);
    if (CELSIUS) h = (h - 32.0f) / 1.8f;          // fahrenheit_to_celsius, float32 (measure.py:37)
    return h;
  };
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n4; i += int64_t(gridDim.x) * blockDim.x) {
    const float4 t = t4[i], r = r4[i];
    o4[i] = make_float4(one(t.x, r.x), one(t.y, r.y), one(t.z, r.z), one(t.w, r.w));
  }
  for (int64_t i = (n4 << 2) + blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n;
       i += int64_t(gridDim.x) * blockDim.x)
    out[i] = one(temp[i], rh[i]);
}

int launch_heat_index(const float *temp_dev, const float *rh_dev, int64_t n, float *out_dev, bool celsius,
                      hipStream_t stream) {
  if (n == 0) return HDP_OK;
  HDP_REQUIRE((reinterpret_cast<uintptr_t>(temp_dev) | reinterpret_cast<uintptr_t>(rh_dev) |
               reinterpret_cast<uintptr_t>(out_dev)) % 16 == 0,
              HDP_EINVAL, "heat index buffers must be 16-byte aligned");
  int64_t g = ((n >> 2) + 255) / 256;
  g = std::max<int64_t>(1, std::min<int64_t>(g, 256 * 16));
  if (celsius)
    hipLaunchKernelGGL(heat_index_kernel<true>, dim3((unsigned)g), dim3(256), 0, stream, temp_dev, rh_dev, n, out_dev);
  else
    hipLaunchKernelGGL(heat_index_kernel<false>, dim3((unsigned)g), dim3(256), 0, stream, temp_dev, rh_dev, n, out_dev);
  HDP_HIP_TRY(hipGetLastError());
  return HDP_OK;
}

// ---- weighted mean of every row: the figure deck's compute_weighted_spatial_mean (hdp/graphics/figure.py:14-15,
// da.weighted(cos(deg2rad(lat))).mean(dim=["lat", "lon"])) on the series-minor device layout, where one row is one
// (metric, percentile, definition, season) over all series.  out[r] = sum_c w[c] v[r][c] / sum_c w[c] over the
// non-NaN values (xarray's weighted mean skips NaNs in the data and in the sum of weights alike); float64
// accumulation in a fixed order (thread-strided partial sums, wave butterfly, four waves through LDS), so the
// result is reproducible; HBM-bound, the row is read once.
template <typename TIn>
__device__ __forceinline__ bool wm_valid(TIn) { return true; }
template <>
__device__ __forceinline__ bool wm_valid<double>(double x) { return x == x; }
template <>
__device__ __forceinline__ bool wm_valid<float>(float x) { return x == x; }

// Wave butterfly + four waves through LDS; every thread returns the workgroup's sum (fixed order).
__device__ __forceinline__ double wm_block_sum(double x, double *part /* [4] */) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
  __syncthreads();  // `part` may still be read from the previous reduction
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = x;
  __syncthreads();
  return (part[0] + part[1]) + (part[2] + part[3]);
}

template <typename TIn>
__global__ __launch_bounds__(256) void weighted_row_mean_kernel(const TIn *__restrict__ v, int64_t n,
                                                                const double *__restrict__ w,
                                                                double *__restrict__ out) {
  __shared__ double part[4];
  const TIn *row = v + int64_t(blockIdx.x) * n;
  double sx = 0.0, sw = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const TIn x = row[i];
    if (wm_valid<TIn>(x)) {
      sx += w[i] * double(x);
      sw += w[i];
    }
  }
  const double tx = wm_block_sum(sx, part);
  const double tw = wm_block_sum(sw, part);
  if (threadIdx.x == 0) out[blockIdx.x] = tw != 0.0 ? tx / tw : __longlong_as_double(0x7ff8000000000000LL);
}

// int16 rows (the metrics' device layout), kWmRows rows per workgroup: the float64 weights -- four times the bytes
// of the values they multiply -- are fetched once for kWmRows rows, eight values per 16-byte load; no NaNs, so the
// sum of weights is the same for every row.  Needs n % 8 == 0 and 16-byte aligned rows (host-checked).
constexpr int kWmRows = 4;
__global__ __launch_bounds__(256) void weighted_rows_mean_i16x8_kernel(const int16_t *__restrict__ v, int64_t n_rows,
                                                                       int64_t n, const double *__restrict__ w,
                                                                       double *__restrict__ out) {
  __shared__ double part[4];
  const int64_t r0 = int64_t(blockIdx.x) * kWmRows;
  const uint4 *rows[kWmRows];
#pragma unroll
  for (int r = 0; r < kWmRows; ++r) rows[r] = reinterpret_cast<const uint4 *>(v + min(r0 + r, n_rows - 1) * n);
  double sx[kWmRows], sw = 0.0;
#pragma unroll
  for (int r = 0; r < kWmRows; ++r) sx[r] = 0.0;
  const double2 *w2 = reinterpret_cast<const double2 *>(w);
  for (int64_t i = threadIdx.x; i < (n >> 3); i += 256) {
    double wi[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double2 t = w2[(i << 2) + k];
      wi[2 * k] = t.x;
      wi[2 * k + 1] = t.y;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) sw += wi[k];
#pragma unroll
    for (int r = 0; r < kWmRows; ++r) {
      const uint4 q = rows[r][i];
      const uint32_t u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        sx[r] += wi[2 * k] * double(int16_t(u[k] & 0xffffu));
        sx[r] += wi[2 * k + 1] * double(int16_t(u[k] >> 16));
      }
    }
  }
  const double tw = wm_block_sum(sw, part);
#pragma unroll
  for (int r = 0; r < kWmRows; ++r) {
    const double tx = wm_block_sum(sx[r], part);
    if (threadIdx.x == 0 && r0 + r < n_rows)
      out[r0 + r] = tw != 0.0 ? tx / tw : __longlong_as_double(0x7ff8000000000000LL);
  }
}

template <typename TIn>
static int launch_weighted_row_mean_t(const TIn *v_dev, int64_t n_rows, int64_t n, const double *w_dev,
                                      double *out_dev, hipStream_t stream) {
  if (n_rows == 0) return HDP_OK;
  HDP_REQUIRE(n_rows < (int64_t(1) << 31), HDP_EUNSUP, "too many rows for one launch");
  hipLaunchKernelGGL(weighted_row_mean_kernel<TIn>, dim3((unsigned)n_rows), dim3(256), 0, stream, v_dev, n, w_dev,
                     out_dev);
  HDP_HIP_TRY(hipGetLastError());
  return HDP_OK;
}
int launch_weighted_row_mean_i16(const int16_t *v_dev, int64_t n_rows, int64_t n, const double *w_dev,
                                 double *out_dev, hipStream_t stream) {
  if (n_rows == 0) return HDP_OK;
  if ((n & 7) == 0 && ((reinterpret_cast<uintptr_t>(v_dev) | reinterpret_cast<uintptr_t>(w_dev)) & 15) == 0) {
    const int64_t g = (n_rows + kWmRows - 1) / kWmRows;
    HDP_REQUIRE(g < (int64_t(1) << 31), HDP_EUNSUP, "too many rows for one launch");
    hipLaunchKernelGGL(weighted_rows_mean_i16x8_kernel, dim3((unsigned)g), dim3(256), 0, stream, v_dev, n_rows, n,
                       w_dev, out_dev);
    HDP_HIP_TRY(hipGetLastError());
    return HDP_OK;
  }
  return launch_weighted_row_mean_t<int16_t>(v_dev, n_rows, n, w_dev, out_dev, stream);
}
int launch_weighted_row_mean_f64(const double *v_dev, int64_t n_rows, int64_t n, const double *w_dev,
                                 double *out_dev, hipStream_t stream) {
  return launch_weighted_row_mean_t<double>(v_dev, n_rows, n, w_dev, out_dev, stream);
}

// ---- layout: time-major [T][n] (CMIP order) -> series-major [n][T] ---------------------------
// 64 x 64 tiles through LDS (pitch 65: conflict-free both ways); reads are coalesced along the
// cell axis, writes along time.  HBM-bound: 8 bytes of traffic per element.
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ src, int64_t src_pitch,
                                                        int64_t T, int64_t n, float *__restrict__ dst, int64_t dp) {
  // 32 time steps x 64 cells per workgroup: 8.3 KB of LDS, so the copy fits beside the whole-cell thresholds
  // kernel (143 KB of a CU's 160 KB) and runs WHILE it computes; rows of 256 B in, 128 B out
  __shared__ float tile[32][65];
  const int64_t t0 = int64_t(blockIdx.y) * 32, c0 = int64_t(blockIdx.x) * 64;
  const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;  // 64 x 4 threads
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t t = t0 + ly + 4 * i, c = c0 + lx;
    if (t < T && c < n) tile[ly + 4 * i][lx] = src[t * src_pitch + c];
  }
  __syncthreads();
  const int tx = threadIdx.x & 31, cy = threadIdx.x >> 5;  // 32 time steps x 8 cells per pass
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t c = c0 + cy + 8 * i, t = t0 + tx;
    if (t < T && c < n) dst[c * dp + t] = tile[tx][cy + 8 * i];
  }
}

// The same copy with NL loads per thread in flight (round 3).  The copy runs BESIDE the kernel that consumes the
// previous chunk: the whole-cell thresholds kernel leaves a CU 20 KB of LDS and 80 registers per SIMD, the packed state
// machines 32 registers per SIMD -- room for one or two of these workgroups, and a workgroup of the kernel above has only
// 8 KB in flight per load-barrier-store round trip (~1.5 TB/s chip-wide at two per CU: slower than the kernel it was
// meant to hide behind).  Here a thread first issues all its NL loads (4 NL time steps x 64 cells = NL KB per
// workgroup, landing in registers), then feeds them through the same 8.3 KB tile 32 time steps at a time.
template <int NL>
__global__ __launch_bounds__(256) void transpose_mlp_kernel(const float *__restrict__ src, int64_t src_pitch,
                                                            int64_t T, int64_t n, float *__restrict__ dst, int64_t dp) {
  static_assert(NL % 8 == 0, "sub-tiles of 32 time steps");
  __shared__ float tile[32][65];
  const int64_t t0 = int64_t(blockIdx.y) * (4 * NL), c0 = int64_t(blockIdx.x) * 64;
  const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;  // 64 cells x 4 time steps per load instruction
  const bool cok = c0 + lx < n;
  float v[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int64_t t = t0 + ly + 4 * i;
    v[i] = (cok && t < T) ? src[t * src_pitch + c0 + lx] : 0.0f;
  }
  const int tx = threadIdx.x & 31, cy = threadIdx.x >> 5;  // 32 time steps x 8 cells per store instruction
#pragma unroll
  for (int s = 0; s < NL / 8; ++s) {
#pragma unroll
    for (int i = 0; i < 8; ++i) tile[ly + 4 * i][lx] = v[8 * s + i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int64_t c = c0 + cy + 8 * i, t = t0 + 32 * s + tx;
      if (t < T && c < n) dst[c * dp + t] = tile[tx][cy + 8 * i];
    }
    __syncthreads();
  }
}

// `beside_state_machines`: the copy runs next to the packed state machines -- 24 loads in flight per thread; otherwise
// (next to the whole-cell thresholds kernel, or on an idle device) 48.  Round 3 measured the 8-load kernel as the better
// neighbour of the state machines (22.3 against 23.2 ms, 109 795 cells); with the staging rows padded to 128 bytes and
// 331 k cells per call (round 4) the metrics pass takes 62.2 / 57.6 / 57.9 ms with 8 / 24 / 48 loads (series-major 41.7);
// capping the state machines at six or five waves per SIMD to leave the copy registers changes nothing (58.2 - 58.4).
int launch_transpose(const float *src_dev, int64_t src_pitch, int64_t T, int64_t n, float *dst_dev,
                     hipStream_t stream, bool beside_state_machines, int64_t dst_pitch) {
  if (T * n == 0) return HDP_OK;
  const int64_t dp = dst_pitch > 0 ? dst_pitch : T;
  static const int nl_env = (int)env_option("HDP_TM_LOADS", -1);  // loads in flight per thread (0: the 8-load kernel); A/B only
  const int nl = nl_env >= 0 ? nl_env : (beside_state_machines ? 24 : 48);
  const int tsteps = nl >= 48 ? 192 : (nl >= 24 ? 96 : (nl >= 16 ? 64 : 32));
  dim3 grid((unsigned)((n + 63) / 64), (unsigned)((T + tsteps - 1) / tsteps));
  HDP_REQUIRE(grid.y < 65536, HDP_EUNSUP, "time axis too long for the transpose launch");
  if (nl >= 48)
    hipLaunchKernelGGL(transpose_mlp_kernel<48>, grid, dim3(256), 0, stream, src_dev, src_pitch, T, n, dst_dev, dp);
  else if (nl >= 24)
    hipLaunchKernelGGL(transpose_mlp_kernel<24>, grid, dim3(256), 0, stream, src_dev, src_pitch, T, n, dst_dev, dp);
  else if (nl >= 16)
    hipLaunchKernelGGL(transpose_mlp_kernel<16>, grid, dim3(256), 0, stream, src_dev, src_pitch, T, n, dst_dev, dp);
  else
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, stream, src_dev, src_pitch, T, n, dst_dev, dp);
  HDP_HIP_TRY(hipGetLastError());
  return HDP_OK;
}

static unsigned grid_for(int64_t total, int block);

// per-cell transpose of an [A][B] float64 matrix: src [n][A][B] -> dst [n][B][A].  Converts between the
// reference's threshold layout (doy, percentile) and the device layout (percentile, doy).
__global__ void swap_last2_f64_kernel(const double *__restrict__ src, int64_t n, int A, int B,
                                      double *__restrict__ dst) {
  const int64_t total = n * A * B;
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < total;
       i += int64_t(gridDim.x) * blockDim.x) {
    const int64_t c = i / (int64_t(A) * B);
    const int r = int(i % (int64_t(A) * B));
    const int b = r / A, a = r % A;  // i indexes dst: [c][b][a]
    dst[i] = src[(c * A + a) * int64_t(B) + b];
  }
}

int launch_swap_last2_f64(const double *src_dev, int64_t n, int64_t A, int64_t B, double *dst_dev,
                          hipStream_t stream) {
  if (n * A * B == 0) return HDP_OK;
  hipLaunchKernelGGL(swap_last2_f64_kernel, dim3(grid_for(n * A * B, 256)), dim3(256), 0, stream, src_dev, n,
                     (int)A, (int)B, dst_dev);
  HDP_HIP_TRY(hipGetLastError());
  return HDP_OK;
}

// ---- synthetic series (reference generator formula, hdp/utils.py:61-78,41) -------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
  z += 0x9e3779b97f4a7c15ULL;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}

__global__ void generate_kernel(float *__restrict__ x, int64_t n_cells, int64_t T, int64_t cell_offset,
                                const float *__restrict__ lat, uint64_t seed, float noise_scale,
                                float trend_per_day) {
  const int64_t total = n_cells * T;
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < total;
       i += int64_t(gridDim.x) * blockDim.x) {
    const int64_t c = i / T, t = i % T;
    const float la = lat[c];
    const float beta = la < 0.0f ? 90.0f : 270.0f;
    const float phase = (float)((t + (int64_t)beta) % 365) * (6.283185307179586f / 365.0f);
    const uint64_t h = splitmix64(seed ^ (uint64_t)((c + cell_offset) * T + t));
    const float u = (float)(h >> 40) * (1.0f / 16777216.0f);
    x[i] = 20.0f + 2.0f * __sinf(phase) - 10.0f * fabsf(la) / 90.0f + u * noise_scale +
           (float)t * trend_per_day;
  }
}

// ---- launchers -----------------------------------------------------------------------------------
// series per launch pair of the split path (bounded scratch; multiples of n_thr_cells when
// ensemble members share thresholds so that `c % n_thr_cells` stays valid inside a batch)
// Year-aligned exceedance words (exceed_years_kernel -> metrics_kernel_cells16): regular calendar of 321..384 days and the
// packed series-per-lane state machines (the only consumer that knows the format).
bool metrics_year_words(const hdp_metrics_plan *plan) {
  // records of at least kYearsMinRecord years (or when forced, HDP_METRICS_YEARS=2): a wave stages 6 x NP thresholds and
  // needs kYearsAhead years to fill its load queue -- at C2's ten years the day-aligned 16-word kernel is faster
  // (metrics 1.13 against 1.61 ms per step)
  const bool long_enough = plan->T >= int64_t(kYearsMinRecord) * plan->n_doy || plan->opt_years >= 2;
  return plan->regular_calendar && plan->opt_years != 0 && long_enough && plan->n_doy > 64 * (kYearSpans - 1) &&
         plan->n_doy <= 64 * kYearSpans && plan->uniform_seasons && !plan->opt_general && !plan->opt_fused &&
         plan->opt_cells != 0 && plan->defs_fit16 && plan->T <= 65535 && plan->opt_packed != 0;
}
int64_t metrics_row_words(const hdp_metrics_plan *plan) {  // words of a (series, percentile) row that carry data
  return metrics_year_words(plan) ? kYearSpans * ((plan->T + plan->n_doy - 1) / plan->n_doy) : (plan->T + 63) >> 6;
}
int64_t metrics_row_pitch(const hdp_metrics_plan *plan) { return (metrics_row_words(plan) + kCW - 1) / kCW * kCW; }

int64_t metrics_batch_cells(const hdp_metrics_plan *plan, int64_t n_cells, int64_t n_thr_cells) {
  const int64_t words_pad = metrics_row_pitch(plan);
  const int64_t row_bytes = plan->P * words_pad * 8;
  // the scratch is a double buffer of at most 2 x 4 GiB.  Batches stay as large as that allows: one wave
  // of the state-machine kernel runs for about a millisecond, so a launch needs many waves per slot
  // (8192 series per launch cost +20 % in tail effects)
  int64_t batch = std::max<int64_t>(1, (int64_t(4) << 30) / row_bytes);
  // calls that would be one or two batches (an 8-GPU shard of C3: 129 600 series) are cut into four, but not below
  // 32 768 series: the first batch's exceedance kernel has nothing to run beside (20.0 -> 19.5 ms at 129 600 series)
  batch = std::min(batch, std::max<int64_t>(32768, (n_cells + 3) / 4));
  if (plan->opt_batch > 0) batch = plan->opt_batch;
  if (n_thr_cells != n_cells && batch < n_cells)
    batch = std::max<int64_t>(n_thr_cells, batch / n_thr_cells * n_thr_cells);
  return std::min<int64_t>(batch, n_cells);
}

static inline bool one_to_one_pre(int64_t n_thr_cells, int64_t n_cells) { return n_thr_cells == n_cells; }

// streams and events of the split path (created once per plan; also by hdp_metrics_plan_reserve, so that a
// later hdp_metrics_f32_dev creates nothing)
static int ensure_plan_streams(const hdp_metrics_plan *plan) {
  if (plan->aux_stream) return HDP_OK;
  // the exceedance kernels are the memory-bound half of the pair: on a high-priority stream their workgroups are
  // dispatched ahead of the state machines' and stream at full rate while those keep the vector units busy
  int prio_lo = 0, prio_hi = 0;
  HDP_HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
  const int64_t want = env_option("HDP_METRICS_PRIO", 1);
  HDP_HIP_TRY(hipStreamCreateWithPriority(&plan->aux_stream, hipStreamNonBlocking, want > 0 ? prio_hi : (want < 0 ? prio_lo : 0)));
  HDP_HIP_TRY(hipStreamCreateWithFlags(&plan->aux_stream2, hipStreamNonBlocking));
  HDP_HIP_TRY(hipEventCreateWithFlags(&plan->ev_fork, hipEventDisableTiming));
  for (int i = 0; i < 2; ++i) {
    HDP_HIP_TRY(hipEventCreateWithFlags(&plan->ev_exceed[i], hipEventDisableTiming));
    HDP_HIP_TRY(hipEventCreateWithFlags(&plan->ev_state[i], hipEventDisableTiming));
  }
  return HDP_OK;
}

int reserve_metrics_scratch(const hdp_metrics_plan *plan, int64_t n_cells) {
  if (!plan->uniform_seasons || n_cells <= 0) return HDP_OK;
  {
    const int rc = ensure_plan_streams(plan);
    if (rc != HDP_OK) return rc;
  }
  const int64_t words_pad = metrics_row_pitch(plan);
  const size_t need = 2 * size_t(metrics_batch_cells(plan, n_cells, n_cells)) * size_t(plan->P) * words_pad * 8;
  if (plan->bits_scratch.bytes < need) {
    hipError_t e = plan->bits_scratch.alloc(need);
    if (e != hipSuccess)
      return set_error(HDP_ENOMEM, "allocating %zu bytes of exceedance scratch failed: %s", need, hipGetErrorString(e));
  }
  return HDP_OK;
}


// ---- season tables that are not increasing and disjoint (user-supplied ranges; the reference's
// compute_heatwave_metrics, metric.py:304-341, takes any) --------------------------------------------------------
// The streaming kernels above close seasons in order; such tables go through the unit-level kernels instead, one
// (percentile, definition) at a time over batches of series: hot days -> index_heatwaves -> season_metrics for the
// northern and the southern table -> the series' own hemisphere is stored.  Slow (the season kernel is quadratic in
// the season length) and meant for the small inputs such tables come with.
__global__ void hot_days_layout_kernel(const float *__restrict__ x, int64_t x_pitch_t, int64_t x_pitch_c, int64_t n,
                                       int64_t T, const double *__restrict__ thr, int64_t n_thr_cells, int P, int p,
                                       int n_doy, const uint16_t *__restrict__ doy_map, int64_t c0,
                                       uint8_t *__restrict__ hot) {
  const int64_t total = n * T;
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
    const int64_t s = i / T, t = i % T;
    const double th = thr[(((c0 + s) % n_thr_cells) * P + p) * int64_t(n_doy) + doy_map[t]];
    hot[i] = ((double)x[(c0 + s) * x_pitch_c + t * x_pitch_t] > th) ? 1 : 0;
  }
}
__global__ void pick_hemisphere_kernel(const int64_t *__restrict__ res_n, const int64_t *__restrict__ res_s,
                                       const uint8_t *__restrict__ is_south, int64_t n, int64_t Y, int64_t c0,
                                       int64_t n_total, int64_t plane /* P*D*Y*n_total */, int64_t pd_index,
                                       int16_t *__restrict__ out) {
  const int64_t total = n * 4 * Y;
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
    const int64_t s = i / (4 * Y), r = i % (4 * Y), m = r / Y, y = r % Y;
    const int64_t v = (is_south[c0 + s] ? res_s : res_n)[i];
    out[m * plane + (pd_index * Y + y) * n_total + c0 + s] = (int16_t)v;
  }
}

static unsigned grid_for(int64_t total, int block);

int launch_metrics_any_ranges(const hdp_metrics_plan *plan, const float *x_dev, const double *thr_dev,
                              int64_t n_thr_cells, const uint8_t *is_south_dev, int64_t n_cells, int16_t *out_dev,
                              hipStream_t stream, int64_t tm_pitch) {
  const int64_t T = plan->T, Y = plan->Y, P = plan->P, D = plan->D;
  const int64_t batch = std::max<int64_t>(1, std::min<int64_t>(n_cells, (int64_t(1) << 30) / (T * 8)));
  DevBuf hot, ids, resn, ress, hwa;
  HDP_HIP_TRY(hot.alloc(size_t(batch) * T));
  HDP_HIP_TRY(ids.alloc(size_t(batch) * T * 8));
  HDP_HIP_TRY(resn.alloc(size_t(batch) * 4 * Y * 8));
  HDP_HIP_TRY(ress.alloc(size_t(batch) * 4 * Y * 8));
  HDP_HIP_TRY(hwa.alloc(size_t(batch) * Y * 8));
  const int64_t pt = tm_pitch > 0 ? tm_pitch : 1, pc = tm_pitch > 0 ? 1 : T;
  const int64_t plane = P * D * Y * n_cells;
  for (int64_t c0 = 0; c0 < n_cells; c0 += batch) {
    const int64_t nb = std::min(batch, n_cells - c0);
    for (int64_t p = 0; p < P; ++p) {
      hipLaunchKernelGGL(hot_days_layout_kernel, dim3(grid_for(nb * T, 256)), dim3(256), 0, stream, x_dev, pt, pc, nb, T,
                         thr_dev, n_thr_cells, (int)P, (int)p, (int)plan->n_doy, plan->doy_map.as<uint16_t>(), c0,
                         hot.as<uint8_t>());
      HDP_HIP_TRY(hipGetLastError());
      for (int64_t d = 0; d < D; ++d) {
        const int64_t *df = plan->defs_host.data() + 3 * d;
        int rc = launch_index_heatwaves(hot.as<uint8_t>(), nb, T, df[0], df[1], df[2], ids.as<int64_t>(), stream);
        if (rc != HDP_OK) return rc;
        rc = launch_season_metrics(ids.as<int64_t>(), nb, T, plan->ranges64.as<int64_t>(), Y, resn.as<int64_t>(),
                                   hwa.as<double>(), stream);
        if (rc != HDP_OK) return rc;
        rc = launch_season_metrics(ids.as<int64_t>(), nb, T, plan->ranges64.as<int64_t>() + 2 * Y, Y, ress.as<int64_t>(),
                                   hwa.as<double>(), stream);
        if (rc != HDP_OK) return rc;
        hipLaunchKernelGGL(pick_hemisphere_kernel, dim3(grid_for(nb * 4 * Y, 256)), dim3(256), 0, stream,
                           resn.as<int64_t>(), ress.as<int64_t>(), is_south_dev, nb, Y, c0, n_cells, plane, p * D + d,
                           out_dev);
        HDP_HIP_TRY(hipGetLastError());
      }
    }
  }
  HDP_HIP_TRY(hipStreamSynchronize(stream));  // the scratch buffers are released on return
  return HDP_OK;
}

int launch_metrics(const hdp_metrics_plan *plan, const float *x_dev, const double *thr_dev,
                   int64_t n_thr_cells, const uint8_t *is_south_dev, int64_t n_cells, int16_t *out_dev,
                   hipStream_t stream, int64_t tm_pitch) {
  if (n_cells == 0) return HDP_OK;
  if (!plan->ordered_seasons)  // overlapping / unordered user tables: per-series path, any ranges
    return launch_metrics_any_ranges(plan, x_dev, thr_dev, n_thr_cells, is_south_dev, n_cells, out_dev, stream, tm_pitch);
  MetDev md;
  md.doy_map = plan->doy_map.as<uint16_t>();
  md.defs = plan->defs.as<int32_t>();
  md.seasons = plan->seasons.as<int2>();
  md.T = (int)plan->T;
  md.xp = plan->T;
  md.n_doy = (int)plan->n_doy;
  md.D = (int)plan->D;
  md.Y = (int)plan->Y;
  md.P = (int)plan->P;
  md.Ypitch = (int)plan->Ypitch;
  const int PD = md.P * md.D;
  md.n_groups = (PD + 63) / 64;
  md.np_max = std::min(md.P, 63 / md.D + 2);
  md.n_doy_pad = (md.n_doy + 3) & ~3;
  md.dmax = (int)plan->dmax;
#ifdef HDP_DEBUG_ABLATIONS
  md.debug = getenv("HDP_METRICS_DEBUG") ? atoi(getenv("HDP_METRICS_DEBUG")) : 0;
#else
  md.debug = 0;
#endif
  md.bits_g = nullptr;
  md.def_perm = nullptr;
  md.out_cells = n_cells;
  md.cell_off = 0;
  md.year_words = metrics_year_words(plan) ? 1 : 0;
  md.n_words = (int)metrics_row_words(plan);
  md.words_pad = (int)metrics_row_pitch(plan);
  // Paths (same results; tests/test_gpu_parity.py runs them against each other):
  //   split + by_cells  exceed_kernel -> metrics_kernel_cells (default)
  //   split             exceed_kernel -> metrics_kernel_uniform<true>       HDP_METRICS_CELLS=0
  //   fused             metrics_kernel_uniform<false>                       HDP_METRICS_FUSED=1
  //   general           metrics_kernel_general: seasons closed per lane     close/unordered season tables
  // The last three produce rows [4][P][D][series][Ypitch] in a scratch; a transpose brings them to the
  // device layout [4][P][D][Y][series].
  const bool uniform = plan->uniform_seasons && !plan->opt_general;
  const bool split = uniform && !plan->opt_fused;
  const bool by_cells = split && plan->opt_cells != 0;
  // packed 16-bit state machines (two definitions per register) when the definitions and the record allow
  const bool pk_ok = by_cells && plan->defs_fit16 && plan->T <= 65535 && plan->opt_packed != 0;
  const size_t seas_bytes = uniform ? 0 : ((size_t(2) * md.Y * sizeof(int2) + 15) & ~size_t(15));
  const size_t thr_bytes = split ? 0 : ((size_t(md.np_max) * md.n_doy_pad * 4 + 15) & ~size_t(15));
  const size_t per_wave = thr_bytes + size_t(md.np_max) * (uniform ? kRow : kChunkWords) * 8;
  const size_t lds = seas_bytes + kMetWaves * per_wave;
  md.seas_bytes = (int)seas_bytes;
  md.thr_bytes = (int)thr_bytes;
  md.wave_bytes = (int)per_wave;
  if (!by_cells)
    HDP_REQUIRE(lds <= kLdsPerCU - 1024, HDP_EUNSUP,
                "metrics kernel needs %zu bytes of LDS (P=%d, n_doy=%d, Y=%d)", lds, md.P, md.n_doy, md.Y);
  HDP_REQUIRE(int64_t(4) * PD < 65536, HDP_EUNSUP, "too many (percentile, definition) pairs");

  // every path works on batches of series: bounded scratch, and (split) overlap of the two kernels
  const bool one_to_one = (n_thr_cells == n_cells);
  const size_t row_bytes = size_t(md.P) * md.words_pad * 8;           // exceedance words of one series
  const size_t rows_bytes = size_t(4) * PD * md.Ypitch * 2;           // row-layout metrics of one series
  int64_t batch = split ? metrics_batch_cells(plan, n_cells, n_thr_cells) : n_cells;
  // Time-major input [T][tm_pitch] (CMIP order): every batch is transposed into a series-major staging buffer on the
  // stream its exceedance kernel runs on -- the only kernel that reads the measure -- so the copy of batch b + 1 runs
  // beside the state machines of batch b.  Smaller batches bound the staging (2 x batch x T x 4 bytes).
  const bool tm = tm_pitch > 0;
  const int64_t Tp = (int64_t(md.T) + 31) & ~int64_t(31);  // staging rows padded to 128 bytes (launch_thresholds_tm has the measurement)
  HDP_REQUIRE(!tm || split, HDP_EUNSUP, "time-major input needs the split metrics path");
  if (tm) {
    int64_t cap = std::max<int64_t>(64, (int64_t(6) << 30) / (int64_t(md.T) * 4));
    if (!one_to_one_pre(n_thr_cells, n_cells) && cap < n_cells) cap = std::max<int64_t>(n_thr_cells, cap / n_thr_cells * n_thr_cells);
    batch = std::min(batch, cap);
  }
  if (!by_cells) {
    int64_t cap = std::max<int64_t>(1, (int64_t(1) << 30) / (int64_t)rows_bytes);
    if (!one_to_one && cap < n_cells) cap = std::max<int64_t>(n_thr_cells, cap / n_thr_cells * n_thr_cells);
    batch = std::min(batch, cap);
  }
  auto grow = [&](hdp::DevBuf &buf, size_t need, const char *what) -> int {
    if (buf.bytes >= need) return HDP_OK;
    HDP_HIP_TRY(hipStreamSynchronize(stream));  // the old scratch may still be in use
    if (plan->aux_stream) HDP_HIP_TRY(hipStreamSynchronize(plan->aux_stream));
    hipError_t e = buf.alloc(need);
    if (e != hipSuccess)
      return set_error(HDP_ENOMEM, "allocating %zu bytes of %s scratch failed: %s", need, what, hipGetErrorString(e));
    return HDP_OK;
  };
  if (split) {
    const int rc = grow(plan->bits_scratch, 2 * size_t(batch) * row_bytes, "exceedance");
    if (rc != HDP_OK) return rc;
  }
  if (tm) {
    const int rc = grow(plan->tm_stage, 2 * size_t(batch) * size_t(Tp) * 4, "time-major staging");
    if (rc != HDP_OK) return rc;
  }
  if (!by_cells) {
    const int rc = grow(plan->rows_scratch, size_t(batch) * rows_bytes, "metrics row");
    if (rc != HDP_OK) return rc;
  }
  // a failed allocation leaves the buffer empty (DevBuf::alloc): never launch on a null scratch
  HDP_REQUIRE(!split || plan->bits_scratch.p, HDP_ENOMEM, "exceedance scratch is not allocated");
  HDP_REQUIRE(by_cells || plan->rows_scratch.p, HDP_ENOMEM, "metrics row scratch is not allocated");
  md.bits_g = plan->bits_scratch.as<unsigned long long>();
  // regular calendars: year-aligned spans, thresholds in registers, words through the scalar cache (exceed_years_kernel)
  const bool years = md.year_words != 0;
  const bool pairs = plan->opt_pairs != 0;  // exceed_pairs_kernel (two percentiles per pass) instead of exceed_kernel (four)
  const size_t lds_a = pairs ? ((size_t((md.P + 1) / 2 * 2) * md.n_doy * 4 + 15) & ~size_t(15))
                             : ((size_t((md.P + kQB - 1) / kQB * kQB) * md.n_doy * 4 + 15) & ~size_t(15));
  const bool short_record = ((md.T + 63) >> 6) <= 64 || plan->opt_cw == 16;  // at most two 32-word chunks: use 16-word chunks, all four waves
  if (split) {
    HDP_REQUIRE(lds_a <= kLdsPerCU - 1024, HDP_EUNSUP, "too many percentiles for the exceedance kernel");
#ifdef HDP_CROSSCHECK_KERNELS
    const void *ek = pairs ? (short_record ? reinterpret_cast<const void *>(exceed_pairs_kernel<16>)
                                           : reinterpret_cast<const void *>(exceed_pairs_kernel<32>))
                           : (short_record ? reinterpret_cast<const void *>(exceed_kernel<16>)
                                           : reinterpret_cast<const void *>(exceed_kernel<32>));
#else
    const void *ek = short_record ? reinterpret_cast<const void *>(exceed_pairs_kernel<16>)
                                  : reinterpret_cast<const void *>(exceed_pairs_kernel<32>);
#endif
    HDP_HIP_TRY(hipFuncSetAttribute(ek, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a));
  }
#ifdef HDP_CROSSCHECK_KERNELS
  auto kern_rows = split ? metrics_kernel_uniform<true> : (uniform ? metrics_kernel_uniform<false> : metrics_kernel_general);
#else
  auto kern_rows = metrics_kernel_general;  // the plan never asks for the (percentile, definition)-per-lane kernels in this build
  HDP_REQUIRE(by_cells || !uniform, HDP_EUNSUP, "cross-check kernels are not in this build");
#endif
  if (!by_cells)
    HDP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern_rows),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const bool overlap = split && plan->opt_overlap != 0;
  if (overlap) {
    const int rc = ensure_plan_streams(plan);
    if (rc != HDP_OK) return rc;
  }
  // fork: the exceedance kernels run on the plan's stream, behind everything already queued on `stream`
  // The state-machine kernels of odd batches run on a second plan stream, so that the workgroups of batch
  // b + 1 fill the slots batch b frees while it drains (one of its waves runs for milliseconds).
  hipStream_t sx = overlap ? plan->aux_stream : stream;
  if (overlap) {
    HDP_HIP_TRY(hipEventRecord(plan->ev_fork, stream));
    HDP_HIP_TRY(hipStreamWaitEvent(sx, plan->ev_fork, 0));
    HDP_HIP_TRY(hipStreamWaitEvent(plan->aux_stream2, plan->ev_fork, 0));
  }
  int64_t b = 0;
  for (int64_t c0 = 0; c0 < n_cells; c0 += batch, ++b) {
    const int64_t nc = std::min(batch, n_cells - c0);
    const int half = int(b & 1);
    // series c of the batch is series c0 + c of the call: pointers are offset, the (c % n_thr_cells)
    // threshold mapping is kept by offsetting the threshold base when it is 1:1
    HDP_REQUIRE(one_to_one || c0 % n_thr_cells == 0 || batch >= n_cells, HDP_EUNSUP,
                "shared thresholds need batches aligned to the number of threshold cells");
    const double *thr_b = one_to_one ? thr_dev + c0 * int64_t(md.n_doy) * md.P : thr_dev;
    const int64_t ntc_b = one_to_one ? nc : n_thr_cells;
    const float *x_b = tm ? plan->tm_stage.as<float>() + size_t(half) * size_t(batch) * size_t(Tp) : x_dev + c0 * int64_t(md.T);
    MetDev mb = md;
    if (tm) mb.xp = Tp;
    mb.bits_g = md.bits_g + size_t(half) * size_t(batch) * (row_bytes / 8);
    hipStream_t sm = (overlap && half) ? plan->aux_stream2 : stream;  // stream of this batch's state machine
    if (split) {
      // this half of the scratch is free once the state machine of batch b - 2 has read it
      if (overlap && b >= 2) HDP_HIP_TRY(hipStreamWaitEvent(sx, plan->ev_state[half], 0));
      if (tm) {  // staging half `half` was last read by the exceedance kernel of batch b - 2, earlier on this stream
        const int rc = launch_transpose(x_dev + c0, tm_pitch, md.T, nc, const_cast<float *>(x_b), sx, true, Tp);
        if (rc != HDP_OK) return rc;
      }
      if (HDP_MDBG(md, 8)) {  // ablation builds: state machines only, on the exceedance words of the previous call
      } else if (years) {
        HDP_REQUIRE(nc < (int64_t(1) << 31), HDP_EUNSUP, "too many series for one launch");
        for (int q0 = 0; q0 < md.P; q0 += kYearsPG) {  // groups of at most kYearsPG percentiles (threshold registers)
          const dim3 g((unsigned)nc), t(64);
          const size_t ballast = (size_t)plan->opt_years_lds;  // unused LDS per wave: caps the resident waves (A/B knob)
          switch (std::min(kYearsPG, md.P - q0)) {
#define HDP_YEARS_CASE(N) case N: hipLaunchKernelGGL(exceed_years_kernel<N>, g, t, ballast, sx, mb, x_b, thr_b, ntc_b, nc, q0); break;
            HDP_YEARS_CASE(1) HDP_YEARS_CASE(2) HDP_YEARS_CASE(3) HDP_YEARS_CASE(4) HDP_YEARS_CASE(5)
            HDP_YEARS_CASE(6) HDP_YEARS_CASE(7) HDP_YEARS_CASE(8) HDP_YEARS_CASE(9)
            default: hipLaunchKernelGGL(exceed_years_kernel<kYearsPG>, g, t, ballast, sx, mb, x_b, thr_b, ntc_b, nc, q0); break;
#undef HDP_YEARS_CASE
          }
        }
      } else if (pairs && short_record)
        hipLaunchKernelGGL(exceed_pairs_kernel<16>, dim3((unsigned)nc), dim3(256), lds_a, sx, mb, x_b, thr_b, ntc_b, nc);
      else if (pairs)
        hipLaunchKernelGGL(exceed_pairs_kernel<32>, dim3((unsigned)nc), dim3(256), lds_a, sx, mb, x_b, thr_b, ntc_b, nc);
#ifdef HDP_CROSSCHECK_KERNELS
      else if (short_record)
        hipLaunchKernelGGL(exceed_kernel<16>, dim3((unsigned)nc), dim3(256), lds_a, sx, mb, x_b, thr_b, ntc_b, nc);
      else
        hipLaunchKernelGGL(exceed_kernel<32>, dim3((unsigned)nc), dim3(256), lds_a, sx, mb, x_b, thr_b, ntc_b, nc);
#endif
      HDP_HIP_TRY(hipGetLastError());
      if (overlap) {
        HDP_HIP_TRY(hipEventRecord(plan->ev_exceed[half], sx));
        HDP_HIP_TRY(hipStreamWaitEvent(sm, plan->ev_exceed[half], 0));
      }
    }
    if (by_cells && HDP_MDBG(md, 16)) continue;  // ablation builds: exceedance words only
    if (by_cells) {
      mb.out_cells = n_cells;  // the device layout is indexed with the series of the whole call
      mb.cell_off = c0;
      const int dg = pk_ok ? (md.D <= 6 ? ((md.D + 1) & ~1) : 6) : (md.D <= 6 ? md.D : 6);
      const int64_t tasks = ((nc + 63) / 64) * md.P * ((md.D + dg - 1) / dg);
      const int64_t blocks = (tasks + kMetWaves - 1) / kMetWaves;
      HDP_REQUIRE(blocks < (int64_t(1) << 31), HDP_EUNSUP, "too many series for one launch");
      const dim3 g((unsigned)blocks), t(kMetWaves * 64);
      const size_t lds_c = size_t(kMetWaves) * 64 * kSlotPitch;
      if (pk_ok) {
        mb.defs = plan->defs16.as<int32_t>();       // definitions with max_break = 0 first
        mb.def_perm = plan->def_perm.as<int32_t>();
        // one launch per pass of dg definitions, instantiated for the number of leading all-simple pairs of that pass
        const int64_t tasks16 = ((nc + 63) / 64) * md.P;
        const dim3 g16((unsigned)((tasks16 + kMetWaves - 1) / kMetWaves));
        for (int d0 = 0; d0 < md.D; d0 += dg) {
          int ns = 0;
          while (plan->opt_simple && ns < dg / 2 && d0 + 2 * ns < md.D &&
                 plan->defs16_host[3 * (d0 + 2 * ns) + 1] == 0 &&
                 (d0 + 2 * ns + 1 >= md.D || plan->defs16_host[3 * (d0 + 2 * ns + 1) + 1] == 0))
            ++ns;
#define HDP_C16_CASE(NPV, NSV) hipLaunchKernelGGL((metrics_kernel_cells16<NPV, NSV>), g16, t, lds_c, sm, mb, is_south_dev + c0, nc, out_dev, d0)
          switch (dg * 4 + ns) {
            case 2 * 4 + 0: HDP_C16_CASE(1, 0); break;
            case 2 * 4 + 1: HDP_C16_CASE(1, 1); break;
            case 4 * 4 + 0: HDP_C16_CASE(2, 0); break;
            case 4 * 4 + 1: HDP_C16_CASE(2, 1); break;
            case 4 * 4 + 2: HDP_C16_CASE(2, 2); break;
            case 6 * 4 + 0: HDP_C16_CASE(3, 0); break;
            case 6 * 4 + 1: HDP_C16_CASE(3, 1); break;
            case 6 * 4 + 2: HDP_C16_CASE(3, 2); break;
            default: HDP_C16_CASE(3, 3); break;
          }
#undef HDP_C16_CASE
        }
      } else
      switch (dg) {
        case 1: hipLaunchKernelGGL(metrics_kernel_cells<1>, g, t, lds_c, sm, mb, is_south_dev + c0, nc, out_dev); break;
        case 2: hipLaunchKernelGGL(metrics_kernel_cells<2>, g, t, lds_c, sm, mb, is_south_dev + c0, nc, out_dev); break;
        case 3: hipLaunchKernelGGL(metrics_kernel_cells<3>, g, t, lds_c, sm, mb, is_south_dev + c0, nc, out_dev); break;
        case 4: hipLaunchKernelGGL(metrics_kernel_cells<4>, g, t, lds_c, sm, mb, is_south_dev + c0, nc, out_dev); break;
        case 5: hipLaunchKernelGGL(metrics_kernel_cells<5>, g, t, lds_c, sm, mb, is_south_dev + c0, nc, out_dev); break;
        default: hipLaunchKernelGGL(metrics_kernel_cells<6>, g, t, lds_c, sm, mb, is_south_dev + c0, nc, out_dev); break;
      }
      HDP_HIP_TRY(hipGetLastError());
      if (overlap) HDP_HIP_TRY(hipEventRecord(plan->ev_state[half], sm));
      continue;
    }
    // (percentile, definition)-per-lane kernels: rows of this batch into the scratch, then transpose
    mb.out_cells = nc;
    mb.cell_off = 0;
    const int64_t tasks = nc * md.n_groups;
    const int64_t blocks = (tasks + kMetWaves - 1) / kMetWaves;
    HDP_REQUIRE(blocks < (int64_t(1) << 31), HDP_EUNSUP, "too many series for one launch");
    int16_t *rows = plan->rows_scratch.as<int16_t>();
    sm = stream;  // one row scratch: these batches stay in order on the caller's stream
    if (overlap) HDP_HIP_TRY(hipStreamWaitEvent(sm, plan->ev_exceed[half], 0));
    hipLaunchKernelGGL(kern_rows, dim3((unsigned)blocks), dim3(kMetWaves * 64), lds, stream, mb, x_b, thr_b, ntc_b,
                       is_south_dev + c0, nc, rows);
    HDP_HIP_TRY(hipGetLastError());
    if (overlap) HDP_HIP_TRY(hipEventRecord(plan->ev_state[half], stream));
    const size_t tile = size_t(64) * (md.Ypitch + 2) * 2;
    hipLaunchKernelGGL(metrics_rows_to_cells_kernel, dim3((unsigned)((nc + 63) / 64), (unsigned)(4 * PD)), dim3(256),
                       tile, stream, rows, md.Y, nc, md.Ypitch, out_dev, n_cells, c0);
    HDP_HIP_TRY(hipGetLastError());
  }
  if (overlap) {  // join: everything the plan's streams did is ordered before what the caller queues next
    HDP_HIP_TRY(hipStreamWaitEvent(stream, plan->ev_exceed[0], 0));
    if (b > 1) HDP_HIP_TRY(hipStreamWaitEvent(stream, plan->ev_exceed[1], 0));
    if (b > 1 && by_cells) HDP_HIP_TRY(hipStreamWaitEvent(stream, plan->ev_state[1], 0));
  }
  return HDP_OK;
}

static unsigned grid_for(int64_t total, int block) {
  int64_t g = (total + block - 1) / block;
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (unsigned)g;
}

int launch_metrics_repack(const int16_t *dev_layout, int64_t P, int64_t D, int64_t n_cells, int64_t Y,
                          int16_t *ref_layout, hipStream_t stream) {
  const int64_t total = P * D * n_cells * 4 * Y;
  if (total == 0) return HDP_OK;
  hipLaunchKernelGGL(metrics_repack_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, dev_layout,
                     P * D, n_cells, Y, ref_layout);
  HDP_HIP_TRY(hipGetLastError());
  return HDP_OK;
}

int launch_metrics_planes_i64(const int16_t *dev_layout, int64_t P, int64_t D, int64_t n_cells, int64_t Y,
                              int64_t *planes, hipStream_t stream) {
  const int64_t total = 4 * P * D * n_cells * Y;
  if (total == 0) return HDP_OK;
  hipLaunchKernelGGL(metrics_planes_i64_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, dev_layout,
                     4 * P * D, n_cells, Y, reinterpret_cast<long long *>(planes));
  HDP_HIP_TRY(hipGetLastError());
  return HDP_OK;
}

int launch_indicate_hot_days(const float *x_dev, int64_t n_series, int64_t T, const double *thr_dev,
                             int64_t n_doy, const int64_t *doy_map_dev, uint8_t *hot_dev,
                             hipStream_t stream) {
  if (n_series * T == 0) return HDP_OK;
  hipLaunchKernelGGL(indicate_hot_days_kernel, dim3(grid_for(n_series * T, 256)), dim3(256), 0, stream,
                     x_dev, n_series, T, thr_dev, n_doy, doy_map_dev, hot_dev);
  HDP_HIP_TRY(hipGetLastError());
  return HDP_OK;
}

int launch_index_heatwaves(const uint8_t *hot_dev, int64_t n_series, int64_t T, int64_t min_dur,
                           int64_t max_break, int64_t max_subs, int64_t *ids_dev, hipStream_t stream) {
  if (n_series == 0) return HDP_OK;
  hipLaunchKernelGGL(index_heatwaves_kernel, dim3((unsigned)((n_series + 63) / 64)), dim3(64), 0, stream,
                     hot_dev, n_series, T, min_dur, max_break, max_subs, ids_dev);
  HDP_HIP_TRY(hipGetLastError());
  return HDP_OK;
}

int launch_season_metrics(const int64_t *ids_dev, int64_t n_series, int64_t T, const int64_t *ranges_dev,
                          int64_t Y, int64_t *out_dev, double *hwa_dev, hipStream_t stream) {
  if (n_series * Y == 0) return HDP_OK;
  hipLaunchKernelGGL(season_metrics_kernel, dim3((unsigned)((n_series * Y + 63) / 64)), dim3(64), 0, stream,
                     ids_dev, n_series, T, ranges_dev, Y, out_dev, hwa_dev);
  HDP_HIP_TRY(hipGetLastError());
  return HDP_OK;
}

int launch_metrics_planes_i64_gathered(const int16_t *gathered, int64_t MPD, int64_t Y, int64_t world, int64_t shard,
                                       int64_t n_mem, int64_t n_total, int64_t r0, int64_t nr, int64_t *planes,
                                       hipStream_t stream) {
  (void)world;
  const int64_t total = nr * n_mem * n_total * Y;
  if (total == 0) return HDP_OK;
  hipLaunchKernelGGL(metrics_planes_i64_gathered_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, gathered, MPD, Y,
                     shard, n_mem, n_total, r0, nr, reinterpret_cast<long long *>(planes));
  HDP_HIP_TRY(hipGetLastError());
  return HDP_OK;
}

int launch_generate(float *x_dev, int64_t n_cells, int64_t T, int64_t cell_offset, const float *lat_dev,
                    uint64_t seed, float noise_scale, float trend_per_day, hipStream_t stream) {
  if (n_cells * T == 0) return HDP_OK;
  hipLaunchKernelGGL(generate_kernel, dim3(grid_for(n_cells * T, 256)), dim3(256), 0, stream, x_dev, n_cells,
                     T, cell_offset, lat_dev, seed, noise_scale, trend_per_day);
  HDP_HIP_TRY(hipGetLastError());
  return HDP_OK;
}

}  // namespace hdp
