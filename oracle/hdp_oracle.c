/*
 * hdp_oracle.c -- TEST INFRASTRUCTURE ONLY: plain-C restatement of the reference's
 * CPU algorithm for the hot path, used (a) as a second, faster checker beside
 * oracle/hdp_oracle.py at sizes the Python oracle cannot reach and (b) as the
 * "port" CPU baseline that bench.py times on the GPU box's host cores.
 * Nothing under hdp_amd/ links or loads it.
 *
 * It follows the reference's per-cell structure on purpose (it is the baseline, not
 * an optimised CPU implementation):
 *   thresholds: per cell, per day-of-year row: gather the B window samples into a
 *     float64 buffer (hdp/threshold.py:74-77) and take each quantile with a
 *     quickselect of the two neighbouring order statistics on that same buffer
 *     (numba np.quantile -> _collect_percentiles_inner -> _select_two), then
 *     lower*(1-m) + upper*m.
 *   metrics: per (cell, percentile, definition): exceedance series
 *     (hdp/metric.py:280-301), edge list + run/gap state machine producing an id
 *     series (:11-60), then per season HWF (:85-102), HWN (:63-82), HWD (:105-137)
 *     and trunc(HWA) (:140-172, :336-340) from the id slice via sort + unique.
 *
 * Build: gcc -O2 -fopenmp -ffp-contract=off -shared -fPIC (see oracle/Makefile).
 * Pinned by tests/test_oracle_c.py against oracle/hdp_oracle.py, which is itself
 * pinned by the reference's known-answer tests and reference-generated vectors.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_set_threads(int n) {
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
}

int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ---- quickselect (median-of-three Lomuto-style partition, like numba's _partition) ---- */
static int64_t partition(double *a, int64_t low, int64_t high) {
  int64_t mid = (low + high) >> 1;
  double t;
  if (a[mid] < a[low]) { t = a[low]; a[low] = a[mid]; a[mid] = t; }
  if (a[high] < a[mid]) {
    t = a[high]; a[high] = a[mid]; a[mid] = t;
    if (a[mid] < a[low]) { t = a[low]; a[low] = a[mid]; a[mid] = t; }
  }
  double pivot = a[mid];
  t = a[high]; a[high] = a[mid]; a[mid] = t;
  int64_t i = low, j = high - 1;
  for (;;) {
    while (i < high && a[i] < pivot) i++;
    while (j >= low && pivot < a[j]) j--;
    if (i >= j) break;
    t = a[i]; a[i] = a[j]; a[j] = t;
    i++; j--;
  }
  t = a[i]; a[i] = a[high]; a[high] = t;
  return i;
}

static double select_k(double *a, int64_t k, int64_t low, int64_t high) {
  int64_t i = partition(a, low, high);
  while (i != k) {
    if (i < k) { low = i + 1; i = partition(a, low, high); }
    else { high = i - 1; i = partition(a, low, high); }
  }
  return a[k];
}

static void select_two(double *a, int64_t k, int64_t low, int64_t high, double *lo, double *hi) {
  for (;;) {
    int64_t i = partition(a, low, high);
    if (i < k) low = i + 1;
    else if (i > k + 1) high = i - 1;
    else if (i == k) { select_k(a, k + 1, i + 1, high); break; }
    else { select_k(a, k, low, i - 1); break; }
  }
  *lo = a[k];
  *hi = a[k + 1];
}

static void quantiles_numba(double *a, int64_t n, const double *q, int64_t P, double *out) {
  int any_nan = 0, all_finite = 1;
  int64_t n_pos = 0, n_neg = 0;
  for (int64_t i = 0; i < n; i++) {
    if (isnan(a[i])) any_nan = 1;
    if (!isfinite(a[i])) all_finite = 0;
    if (a[i] == INFINITY) n_pos++;
    if (a[i] == -INFINITY) n_neg++;
  }
  if (any_nan) { for (int64_t p = 0; p < P; p++) out[p] = NAN; return; }
  if (n == 1) { for (int64_t p = 0; p < P; p++) out[p] = isfinite(a[0]) ? a[0] : NAN; return; }
  for (int64_t p = 0; p < P; p++) {
    double pct = q[p] * 100.0, val;
    if (pct == 100.0) {
      val = a[0];
      for (int64_t i = 1; i < n; i++) if (a[i] > val) val = a[i];
      if (!all_finite && !isfinite(val)) val = NAN;
    } else if (pct == 0.0) {
      val = a[0];
      for (int64_t i = 1; i < n; i++) if (a[i] < val) val = a[i];
      if (!all_finite) {
        int64_t n_fin = n - (n_pos + n_neg);
        if (n_fin == 0) val = NAN;
        if (n_pos == 1 && n == 2) val = NAN;
        if (n_neg > 1) val = NAN;
        if (n_fin == 1 && n_pos > 1 && n_neg != 1) val = NAN;
      }
    } else {
      volatile double frac = pct / 100.0;
      volatile double prod = (double)(n - 1) * frac;
      volatile double rank = 1.0 + prod;
      double f = floor(rank);
      volatile double m = rank - f;
      volatile double om = 1.0 - m;
      int64_t k = (int64_t)f - 1;
      double lo, hi;
      if (k + 1 > n - 1) { lo = select_k(a, k, 0, n - 1); hi = lo; }
      else select_two(a, k, 0, n - 1, &lo, &hi);
      volatile double t1 = lo * om;
      volatile double t2 = hi * m;
      val = t1 + t2;
    }
    out[p] = val;
  }
}

/* x [n_cells][T] f32, win [n_doy][B] int64 (negative wraps), q [P] -> out [n_cells][n_doy][P] */
int oracle_thresholds(const float *x, int64_t n_cells, int64_t T, const int64_t *win, int64_t n_doy,
                      int64_t B, const double *q, int64_t P, double *out) {
#pragma omp parallel
  {
    double *buf = (double *)malloc(sizeof(double) * (size_t)B);
#pragma omp for schedule(dynamic, 1)
    for (int64_t c = 0; c < n_cells; c++) {
      const float *xc = x + c * T;
      for (int64_t d = 0; d < n_doy; d++) {
        for (int64_t i = 0; i < B; i++) {
          int64_t t = win[d * B + i];
          if (t < 0) t += T;
          buf[i] = (double)xc[t];
        }
        quantiles_numba(buf, B, q, P, out + (c * n_doy + d) * P);
      }
    }
    free(buf);
  }
  return 0;
}

/* ---- metrics ------------------------------------------------------------------------------ */
static int cmp_i64(const void *a, const void *b) {
  int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
  return (x > y) - (x < y);
}

static void index_heatwaves(const uint8_t *hot, int64_t T, int64_t min_dur, int64_t max_break,
                            int64_t max_subs, int64_t *ids, int64_t *edges) {
  /* edge list of the zero-padded series (metric.py:27-32) */
  int64_t ne = 0;
  int prev = 0;
  for (int64_t i = 0; i <= T; i++) {
    int cur = (i < T) ? (hot[i] != 0) : 0;
    if (cur != prev) edges[ne++] = i;
    prev = cur;
  }
  memset(ids, 0, sizeof(int64_t) * (size_t)T);
  int in_hw = 0;
  int64_t cur_id = 0, subs = 0;
  for (int64_t e = 0; e + 1 < ne; e++) {
    int64_t a = edges[e], b = edges[e + 1], span = b - a;
    int rising = (e % 2) == 0; /* edges alternate +1, -1 starting with +1 */
    int label = 0;
    if (rising && span >= min_dur && !in_hw) { cur_id++; in_hw = 1; label = 1; }
    else if (!rising && span > max_break) in_hw = 0;
    else if (rising && in_hw && subs < max_subs) { subs++; label = 1; }
    else if (rising && in_hw && subs >= max_subs) {
      if (span >= min_dur) { cur_id++; label = 1; }
      else in_hw = 0;
      subs = 0;
    }
    if (label) for (int64_t t = a; t < b; t++) ids[t] = cur_id;
  }
}

/* HWF, HWN, HWD, trunc(HWA) of one id slice; `tmp` holds the sorted copy */
static void season_metrics(const int64_t *ids, int64_t a, int64_t b, int64_t *tmp, int64_t *o4,
                           double *hwa_out) {
  int64_t n = b - a, hwf = 0;
  for (int64_t i = 0; i < n; i++) { tmp[i] = ids[a + i]; if (tmp[i] > 0) hwf++; }
  qsort(tmp, (size_t)n, sizeof(int64_t), cmp_i64);
  /* unique values with counts */
  int64_t nu = 0, nnz = 0;
  for (int64_t i = 0; i < n;) {
    int64_t j = i;
    while (j < n && tmp[j] == tmp[i]) j++;
    if (tmp[i] != 0) nnz++;
    nu++;
    i = j;
  }
  int64_t longest = 0, total = 0, kept = 0, seen = 0;
  for (int64_t i = 0; i < n;) {
    int64_t j = i;
    while (j < n && tmp[j] == tmp[i]) j++;
    int drop = (nu >= 2 && seen == 0); /* metric.py:124-128: smallest unique dropped */
    seen++;
    if (!drop) {
      kept++;
      if (tmp[i] != 0) {
        int64_t cnt = j - i;
        total += cnt;
        if (cnt > longest) longest = cnt;
      }
    }
    i = j;
  }
  double mean = kept ? (double)total / (double)kept : NAN;
  o4[0] = hwf; o4[1] = nnz; o4[2] = longest; o4[3] = (int64_t)mean;
  if (hwa_out) *hwa_out = mean;
}

/* x [n][T] f32, thr [n_thr][n_doy][P] f64 (series c uses row c % n_thr), doy_map [T],
 * defs [D][3], north/south [Y][2], is_south [n] -> out [P][D][n][4][Y] int64 */
int oracle_metrics(const float *x, int64_t n, int64_t T, const double *thr, int64_t n_thr, int64_t n_doy,
                   int64_t P, const int64_t *doy_map, const int64_t *defs, int64_t D,
                   const int64_t *north, const int64_t *south, const uint8_t *is_south, int64_t Y,
                   int64_t *out) {
#pragma omp parallel
  {
    uint8_t *hot = (uint8_t *)malloc((size_t)T);
    int64_t *ids = (int64_t *)malloc(sizeof(int64_t) * (size_t)T);
    int64_t *edges = (int64_t *)malloc(sizeof(int64_t) * (size_t)(T + 2));
    int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (size_t)T);
#pragma omp for schedule(dynamic, 1)
    for (int64_t c = 0; c < n; c++) {
      const float *xc = x + c * T;
      const double *tc = thr + (c % n_thr) * n_doy * P;
      const int64_t *seas = is_south[c] ? south : north;
      for (int64_t p = 0; p < P; p++) {
        for (int64_t d = 0; d < D; d++) {
          /* the reference recomputes the exceedance series for every definition (metric.py:329) */
          for (int64_t t = 0; t < T; t++) hot[t] = ((double)xc[t] > tc[doy_map[t] * P + p]) ? 1 : 0;
          index_heatwaves(hot, T, defs[3 * d], defs[3 * d + 1], defs[3 * d + 2], ids, edges);
          for (int64_t y = 0; y < Y; y++) {
            int64_t o4[4];
            season_metrics(ids, seas[2 * y], seas[2 * y + 1], tmp, o4, 0);
            int64_t *o = out + (((p * D + d) * n + c) * 4) * Y;
            o[0 * Y + y] = o4[0]; o[1 * Y + y] = o4[1]; o[2 * Y + y] = o4[2]; o[3 * Y + y] = o4[3];
          }
        }
      }
    }
    free(hot); free(ids); free(edges); free(tmp);
  }
  return 0;
}
