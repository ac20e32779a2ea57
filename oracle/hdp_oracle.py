"""CPU oracle for the HDP hot path -- TEST INFRASTRUCTURE ONLY.

This module is a plain numpy/Python restatement of the algorithms in the
reference (AgentOxygen/HDP) for the two hot loops:

  * hdp/threshold.py  -- day-of-year rolling-window percentile thresholds
  * hdp/metric.py     -- exceedance -> heatwave labelling -> HWF/HWN/HWD/HWA

It exists so the HIP kernels can be checked against the reference's
arithmetic.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; nothing under
``hdp_amd/`` does.  It is never the thing that is measured or shipped.

Pinning status (see DESIGN.md "Oracle"):
  * metric functions: pinned by the 37 known-answer tests of the reference
    (restated as data in tests/golden/reference_kat.json) and by vectors
    generated from the reference's own ``hdp/metric.py`` run in the build
    container (tests/golden/make_golden.py).
  * threshold values: the reference holds NO numeric test for them and numba
    is not installable here, so the quantile arithmetic follows numba's
    published algorithm (numba/np/arraymath.py ``_collect_percentiles_inner``,
    numba >= 0.60 per pyproject.toml:23; text read from the 0.54.1 copy in
    /opt/conda).  "parity unpinned" at the 1-ulp level; pinned to <=1e-6
    relative (north_star tolerance) against vectors produced by the
    reference's own gather loop + NumPy's np.quantile.

Every function cites the reference file:line it follows.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

# --------------------------------------------------------------------------
# calendar stand-in (the reference uses cftime objects; only these attributes
# are ever read: threshold.py:30, metric.py:190-201, metric.py:276)
# --------------------------------------------------------------------------

_NOLEAP_MONTH_LEN = (31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31)


@dataclass(frozen=True)
class NoLeapDate:
    year: int
    month: int
    day: int
    calendar: str = "noleap"

    @property
    def dayofyr(self) -> int:
        return sum(_NOLEAP_MONTH_LEN[: self.month - 1]) + self.day

    def __str__(self) -> str:  # cftime prints "YYYY-MM-DD 00:00:00"
        return f"{self.year:04d}-{self.month:02d}-{self.day:02d} 00:00:00"


def noleap_date_range(start: str, end: str) -> np.ndarray:
    """Daily noleap dates, inclusive of both ends ("YYYY-MM-DD" or "YYYY")."""

    def parse(s, last):
        parts = [int(p) for p in s.split("-")]
        if len(parts) == 1:
            parts += [1, 1]
        return parts

    y0, m0, d0 = parse(start, False)
    y1, m1, d1 = parse(end, True)
    out = []
    y, m, d = y0, m0, d0
    while (y, m, d) <= (y1, m1, d1):
        out.append(NoLeapDate(y, m, d))
        d += 1
        if d > _NOLEAP_MONTH_LEN[m - 1]:
            d = 1
            m += 1
            if m > 12:
                m = 1
                y += 1
    return np.array(out, dtype=object)


# --------------------------------------------------------------------------
# synthetic generators (hdp/utils.py:39-92), numpy-only
# --------------------------------------------------------------------------

def generate_control(start="1700-01-01", end="1749-12-31", grid_shape=(2, 3),
                     add_noise=False, seed=0):
    """utils.py:53-92 -> (data float64[lon,lat,time], lon, lat, dates)."""
    dates = noleap_date_range(start, end)
    t = np.arange(dates.size, dtype=float)
    north = 20 + 2 * np.sin(2 * np.pi * ((270 + t) / 365))
    south = 20 + 2 * np.sin(2 * np.pi * ((90 + t) / 365))
    n_lon, n_lat = grid_shape
    vals = np.zeros((n_lon, n_lat, t.size))
    vals[:, n_lat // 2:, :] = north
    vals[:, : n_lat // 2, :] = south
    if add_noise:
        np.random.seed(seed)
        vals += np.random.random(vals.shape) * (np.std(vals) / 2)
    lat = np.linspace(-90, 90, n_lat, dtype=float)
    lon = np.linspace(-180, 180, n_lon, dtype=float)
    grad = np.broadcast_to(np.abs(lat) / 90, grid_shape)
    vals = vals - 10 * grad[:, :, None]
    return vals, lon, lat, dates


def generate_warming(start="2000-01-01", end="2049-12-31", grid_shape=(2, 3),
                     warming_period=100, add_noise=False):
    """utils.py:39-42."""
    vals, lon, lat, dates = generate_control(start, end, grid_shape, add_noise)
    vals = vals + (np.arange(dates.size) / (365 * warming_period))[None, None, :]
    return vals, lon, lat, dates


# --------------------------------------------------------------------------
# thresholds: window table + numba quantile arithmetic
# --------------------------------------------------------------------------

def datetimes_to_windows(datetimes, window_radius: int) -> np.ndarray:
    """threshold.py:12-49 -> int64[n_doy, (2r+1)*S].

    Rows follow first-occurrence order of the day-of-year values, short rows
    are padded with -1, the lower edge wraps through negative indexing and the
    upper edge is *reflected* (``n - sample_index``), exactly as the reference.
    """
    buckets: dict[int, list[int]] = {}
    for i, d in enumerate(datetimes):
        buckets.setdefault(d.dayofyr, []).append(i)
    n_doy = len(buckets)
    s_max = max(len(v) for v in buckets.values())
    time_index = np.full((n_doy, s_max), -1, dtype=np.int64)
    for row, members in enumerate(buckets.values()):
        time_index[row, : len(members)] = members
    width = 2 * window_radius + 1
    table = np.empty((n_doy, width, s_max), dtype=np.int64)
    for row in range(n_doy):
        for w in range(width):
            src = row + window_radius - w
            if src >= n_doy:
                src = n_doy - src          # reflection, threshold.py:46-47
            table[row, w] = time_index[src]  # negative src wraps (numpy)
    return table.reshape(n_doy, width * s_max)


def numba_quantile(a, q) -> np.ndarray:
    """numba np.quantile (arraymath.py _collect_percentiles / _inner).

    Order statistics are taken from a full sort; numba's quickselect returns
    the same values.  Arithmetic order is numba's: ``q*100``, ``/100.0``,
    ``1 + (n-1)*frac``, ``lower*(1-m) + upper*m`` with no fused multiply-add.
    """
    q = np.asarray(q, dtype=np.float64).ravel()
    if np.any(np.isnan(q)) or np.any(q < 0.0) or np.any(q > 1.0):
        raise ValueError("Quantiles must be in the range [0, 1]")
    pct = q * 100.0
    a = np.asarray(a, dtype=np.float64).ravel()
    out = np.full(q.size, np.nan)
    if np.isnan(a).any():
        return out
    n = a.size
    if n == 1:
        if np.isfinite(a[0]):
            out[:] = a[0]
        return out
    s = np.sort(a)
    all_finite = bool(np.all(np.isfinite(a)))
    for i, p in enumerate(pct):
        if p == 100:
            val = s[-1]
            if not all_finite and not np.isfinite(val):
                val = np.nan
        elif p == 0:
            val = s[0]
            if not all_finite:
                n_pos = int(np.sum(a == np.inf))
                n_neg = int(np.sum(a == -np.inf))
                n_fin = n - (n_pos + n_neg)
                if n_fin == 0:
                    val = np.nan
                if n_pos == 1 and n == 2:
                    val = np.nan
                if n_neg > 1:
                    val = np.nan
                if n_fin == 1 and n_pos > 1 and n_neg != 1:
                    val = np.nan
        else:
            rank = np.float64(1) + np.float64(n - 1) * (np.float64(p) / np.float64(100.0))
            f = math.floor(rank)
            m = np.float64(rank) - np.float64(f)
            lower = s[int(f) - 1]
            upper = s[min(int(f), n - 1)]
            with np.errstate(invalid="ignore"):
                val = lower * (np.float64(1) - m) + upper * m
        out[i] = val
    return out


def compute_percentiles(temperatures, window_samples, percentiles) -> np.ndarray:
    """threshold.py:59-78 for one cell: float32[T] -> float64[n_doy, P]."""
    temperatures = np.asarray(temperatures)
    out = np.empty((window_samples.shape[0], len(percentiles)), dtype=np.float64)
    for row in range(window_samples.shape[0]):
        buf = temperatures[window_samples[row]].astype(np.float64)  # -1 -> last
        out[row] = numba_quantile(buf, percentiles)
    return out


def compute_thresholds_cells(x, window_samples, percentiles) -> np.ndarray:
    """All cells: x float32[n_cells, T] -> float64[n_cells, n_doy, P]."""
    x = np.asarray(x, dtype=np.float32)
    return np.stack([compute_percentiles(x[c], window_samples, percentiles)
                     for c in range(x.shape[0])])


# --------------------------------------------------------------------------
# metrics
# --------------------------------------------------------------------------

def build_doy_map(times) -> np.ndarray:
    """metric.py:265-277."""
    return np.array([t.dayofyr - 1 for t in times], dtype=np.int64)


def get_range_indices(times, start, end) -> np.ndarray:
    """metric.py:175-209 -> int64[num_years, 2], -1 = not found."""
    n_years = times[-1].year - times[0].year + 1
    ranges = np.full((n_years, 2), -1, dtype=np.int64)
    row = 0
    want_start = True
    for i, d in enumerate(times):
        if want_start:
            if (d.month, d.day) == tuple(start):
                ranges[row, 0] = i
                want_start = False
        elif (d.month, d.day) == tuple(end):
            ranges[row, 1] = i
            row += 1
            want_start = True
    if not want_start:
        ranges[-1, -1] = len(times)
    return ranges


def hemisphere_ranges(times):
    """metric.py:221-243: (north[Y,2], south[Y,2], years[Y]) after trimming."""
    north = get_range_indices(times, (5, 1), (10, 1))
    south = get_range_indices(times, (11, 1), (4, 1))
    lo, hi = 0, north.size
    started = False
    for y in range(north.shape[0]):
        has_gap = (-1 in north[y]) or (-1 in south[y])
        if has_gap and not started:
            lo = y
            continue
        if not started:
            started = True
        if started and has_gap:
            hi = y
            break
    years = np.arange(times[0].year, times[-1].year + 1)
    return north[lo:hi], south[lo:hi], years[lo:hi]


def indicate_hot_days(measure, threshold, doy_map) -> np.ndarray:
    """metric.py:280-301 (strict >, NaN compares False)."""
    measure = np.asarray(measure)
    thr = np.asarray(threshold, dtype=np.float64)[np.asarray(doy_map)]
    with np.errstate(invalid="ignore"):
        return measure.astype(np.float64) > thr


def index_heatwaves(hot, min_duration, max_break, max_subs) -> np.ndarray:
    """metric.py:11-60: edge list (pad + diff), then the run/gap state machine."""
    hot = np.asarray(hot).astype(bool)
    padded = np.zeros(hot.size + 2, dtype=np.int64)
    padded[1:-1] = hot
    step = np.diff(padded)
    edges = np.flatnonzero(step)
    ids = np.zeros(step.size, dtype=np.int64)
    active = False
    current = 0
    subs = 0
    for a, b in zip(edges[:-1], edges[1:]):
        span = b - a
        rising = step[a] == 1
        if rising and span >= min_duration and not active:
            current += 1
            active = True
            ids[a:b] = current
        elif (not rising) and span > max_break:
            active = False
        elif rising and active and subs < max_subs:
            subs += 1
            ids[a:b] = current
        elif rising and active and subs >= max_subs:
            if span >= min_duration:
                current += 1
                ids[a:b] = current
            else:
                active = False
            subs = 0
    return ids[:-1]


def heatwave_frequency(ids, seasons) -> np.ndarray:
    """metric.py:85-102."""
    ids = np.asarray(ids)
    return np.array([int(np.sum(ids[s:e] > 0)) for s, e in seasons], dtype=np.int64)


def heatwave_number(ids, seasons) -> np.ndarray:
    """metric.py:63-82."""
    ids = np.asarray(ids)
    out = []
    for s, e in seasons:
        u = np.unique(ids[s:e])
        out.append(int(np.sum(u != 0)))
    return np.array(out, dtype=np.int64)


def _season_lengths(piece):
    """Shared body of metric.py:121-134 / 156-169 incl. the 'drop the smallest
    unique value when there are >= 2' quirk."""
    u = np.unique(piece)
    if u.size != 1:
        u = u[1:]
    return np.array([int(np.sum(piece == v)) if v != 0 else 0 for v in u], dtype=np.int64)


def heatwave_duration(ids, seasons) -> np.ndarray:
    """metric.py:105-137."""
    ids = np.asarray(ids).astype(np.int64)
    return np.array([int(np.max(_season_lengths(ids[s:e]))) for s, e in seasons],
                    dtype=np.int64)


def heatwave_average(ids, seasons) -> np.ndarray:
    """metric.py:140-172 (float64; truncated later by the int64 store)."""
    ids = np.asarray(ids).astype(np.int64)
    return np.array([float(np.mean(_season_lengths(ids[s:e]))) for s, e in seasons],
                    dtype=np.float64)


def compute_heatwave_metrics(measure, threshold, doy_map, min_duration, max_break,
                             max_subs, seasons) -> np.ndarray:
    """metric.py:304-341 -> int64[4, Y] rows HWF, HWN, HWD, HWA."""
    hot = indicate_hot_days(measure, threshold, doy_map)
    ids = index_heatwaves(hot, min_duration, max_break, max_subs)
    out = np.zeros((4, len(seasons)), dtype=np.int64)
    out[0] = heatwave_frequency(ids, seasons)
    out[1] = heatwave_number(ids, seasons)
    out[2] = heatwave_duration(ids, seasons)
    out[3] = heatwave_average(ids, seasons).astype(np.int64)  # truncation
    return out


def compute_metrics_cells(x, thr, doy_map, defs, north, south, is_south) -> np.ndarray:
    """All cells: x[n_cells,T], thr[n_cells,n_doy,P] -> int64[P, D, n_cells, 4, Y]
    (metric.py:357-369 loop order)."""
    n_cells = x.shape[0]
    P = thr.shape[2]
    D = len(defs)
    Y = north.shape[0]
    out = np.zeros((P, D, n_cells, 4, Y), dtype=np.int64)
    for c in range(n_cells):
        seasons = south if is_south[c] else north
        for p in range(P):
            hot = indicate_hot_days(x[c], thr[c, :, p], doy_map)
            for d, (a, b, s) in enumerate(defs):
                ids = index_heatwaves(hot, a, b, s)
                out[p, d, c, 0] = heatwave_frequency(ids, seasons)
                out[p, d, c, 1] = heatwave_number(ids, seasons)
                out[p, d, c, 2] = heatwave_duration(ids, seasons)
                out[p, d, c, 3] = heatwave_average(ids, seasons).astype(np.int64)
    return out


# --------------------------------------------------------------------------
# heat index pre-step (hdp/measure.py:61-94) -- SURVEY.md 8(f) row 1
# --------------------------------------------------------------------------

def heat_index(temp, rel_humid) -> np.ndarray:
    """measure.py:61-94 with Numba's typing of ``float32(float32, float32)``: the literals are
    float64, so every operation promotes to float64 except ``rel_humid*temp`` (float32 * float32)
    in the last polynomial term; the result is rounded to float32 once on return.  "parity
    unpinned" at the ulp level (numba is not installable here); the stub-run of the reference
    (NumPy scalar semantics, float32 throughout) agrees to ~1e-5 relative."""
    t32 = np.asarray(temp, dtype=np.float32)
    r32 = np.broadcast_to(np.asarray(rel_humid, dtype=np.float32), t32.shape)
    t = t32.astype(np.float64)
    r = r32.astype(np.float64)
    simple = 0.5 * (t + 61.0 + ((t - 68.0) * 1.2) + (r * 0.094))
    hi = np.full(t.shape, -42.379)
    hi = hi + 2.04901523 * t
    hi = hi + 10.14333127 * r
    hi = hi + -0.22475541 * t * r
    hi = hi + -0.00683783 * (t * t)
    hi = hi + -0.05481717 * (r * r)
    hi = hi + 0.00122874 * (t * t) * r
    hi = hi + 0.00085282 * t * (r * r)
    rt = (r32 * t32).astype(np.float64)                       # float32 product
    hi = hi + -0.00000199 * (rt * rt)
    dry = (r32 < 13) & (80 <= t32) & (t32 <= 112)
    wet = (~dry) & (r32 > 85) & (80 <= t32) & (t32 <= 87)
    with np.errstate(invalid="ignore"):
        adj_dry = ((13 - r) / 4) * np.sqrt(np.abs(17 - np.abs(t - 95)) / 17)
    adj_wet = ((r - 85) / 10) * ((87 - t) / 5)
    hi = np.where(dry, hi - adj_dry, hi)
    hi = np.where(wet, hi + adj_wet, hi)
    return np.where(simple > 80, hi, simple).astype(np.float32)


def heat_index_celsius(temp_c, rel_humid) -> np.ndarray:
    """What format_standard_measures wraps around the ufunc (measure.py:185-189): float32
    C -> F (measure.py:54), heat index, float32 F -> C (measure.py:37)."""
    t = np.asarray(temp_c, dtype=np.float32)
    tf = (t * np.float32(1.8)) + np.float32(32)
    hi = heat_index(tf, rel_humid)
    return (hi - np.float32(32)) / np.float32(1.8)


def weighted_spatial_mean(values, lat, n_lon) -> np.ndarray:
    """compute_weighted_spatial_mean (hdp/graphics/figure.py:14-15) restated: xarray's
    ``da.weighted(cos(deg2rad(lat))).mean(dim=["lat", "lon"])`` is sum(w*x)/sum(w) over the non-NaN x
    (xarray/core/weighted.py: _weighted_sum / _sum_of_weights, both masked by da.notnull()).
    values [..., n_lat, n_lon] -> [...].  xarray is not installed here: parity with it is unpinned."""
    v = np.asarray(values, dtype=np.float64)
    w = np.broadcast_to(np.cos(np.deg2rad(np.asarray(lat, dtype=np.float64)))[:, None], v.shape[-2:])
    valid = ~np.isnan(v)
    sx = np.where(valid, v * w, 0.0).sum(axis=(-2, -1))
    sw = np.where(valid, w, 0.0).sum(axis=(-2, -1))
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.where(sw != 0.0, sx / sw, np.nan)
