"""Host-side index tables for the hot path (Python, like the reference's).

Product code: restates what the reference builds on the host before its Numba
kernels run, in the compact form the HIP kernels consume.

  window_columns          <- hdp/threshold.py:12-49  datetimes_to_windows
  build_doy_map           <- hdp/metric.py:265-277
  get_range_indices       <- hdp/metric.py:175-209
  hemisphere_season_tables<- hdp/metric.py:221-243   (the trimming part of
                             compute_hemisphere_ranges; the per-cell broadcast at
                             :245-252 becomes a one-byte hemisphere flag per cell)

Dates are duck-typed exactly as in the reference: anything exposing
``.dayofyr``, ``.year``, ``.month``, ``.day`` (cftime objects do).
"""
from __future__ import annotations

import numpy as np


def _attr(dates, name):
    return np.fromiter((getattr(d, name) for d in dates), dtype=np.int64, count=len(dates))


def window_columns(datetimes, window_radius: int):
    """Compact form of the reference window table.

    Returns ``(time_index int64[n_doy, S], cols int32[n_doy, 2r+1])`` such that row
    ``d`` of ``datetimes_to_windows`` equals ``time_index[cols[d]].ravel()``:
    every row of the reference's table is a concatenation of whole day-of-year
    columns (threshold.py:43-48).  Quirks kept: rows follow first-occurrence order
    of the day-of-year values (:28-33), short rows are -1 padded (:35), the lower
    window edge wraps through negative indexing and the upper edge is reflected,
    ``n_doy - sample_index`` (:46-47).
    """
    doy = _attr(datetimes, "dayofyr")
    if doy.size == 0:
        raise ValueError("max() arg is an empty sequence")  # what the reference raises (:35)
    uniq, first = np.unique(doy, return_index=True)
    order = uniq[np.argsort(first, kind="stable")]          # first-occurrence order
    row_of = {int(v): i for i, v in enumerate(order)}
    rows = np.fromiter((row_of[int(v)] for v in doy), dtype=np.int64, count=doy.size)
    n_doy = order.size
    counts = np.bincount(rows, minlength=n_doy)
    S = int(counts.max())
    time_index = np.full((n_doy, S), -1, dtype=np.int64)
    by_row = np.argsort(rows, kind="stable")                 # time order within each row
    slot = np.arange(doy.size) - np.repeat(np.cumsum(counts) - counts, counts)
    time_index[rows[by_row], slot] = by_row
    W = 2 * int(window_radius) + 1
    d = np.arange(n_doy)[:, None]
    src = d + int(window_radius) - np.arange(W)[None, :]
    src = np.where(src >= n_doy, n_doy - src, src)           # reflection, not wrap-around
    if np.any(src < -n_doy):
        raise IndexError(f"index {int(src.min())} is out of bounds for axis 0 with size {n_doy}")
    src = np.where(src < 0, src + n_doy, src)                # NumPy negative indexing
    return time_index, src.astype(np.int32)


def expand_window_table(time_index, cols):
    """The literal [n_doy, (2r+1)*S] gather table of the reference."""
    return time_index[cols].reshape(cols.shape[0], -1)


def datetimes_to_windows(datetimes, window_radius: int) -> np.ndarray:
    """Same signature and result as hdp.threshold.datetimes_to_windows."""
    return expand_window_table(*window_columns(datetimes, window_radius))


def build_doy_map(times) -> np.ndarray:
    """metric.py:265-277: ``dayofyr - 1`` per time step."""
    return _attr(times, "dayofyr") - 1


def get_range_indices(times, start, end) -> np.ndarray:
    """metric.py:175-209: [start, end) index ranges of each season, -1 = missing.

    Seasons are found by alternately scanning for the start and the end (month, day);
    a season still open at the end of the record is closed at ``len(times)`` in the
    LAST row of the table (metric.py:206-207)."""
    month, day = _attr(times, "month"), _attr(times, "day")
    n_years = int(times[-1].year - times[0].year + 1)
    ranges = np.full((n_years, 2), -1, dtype=np.int64)
    starts = np.flatnonzero((month == start[0]) & (day == start[1]))
    ends = np.flatnonzero((month == end[0]) & (day == end[1]))
    row, cursor, open_season = 0, -1, False
    si = ei = 0
    while True:
        if not open_season:
            while si < starts.size and starts[si] <= cursor:
                si += 1
            if si == starts.size:
                break
            cursor = starts[si]
            ranges[row, 0] = cursor
            open_season = True
        else:
            while ei < ends.size and ends[ei] <= cursor:
                ei += 1
            if ei == ends.size:
                break
            cursor = ends[ei]
            ranges[row, 1] = cursor
            row += 1
            open_season = False
    if open_season:
        ranges[-1, -1] = len(times)
    return ranges


def hemisphere_season_tables(times):
    """metric.py:221-243: northern (May 1 -> Oct 1) and southern (Nov 1 -> Apr 1)
    season tables with incomplete leading/trailing years trimmed, plus the years kept."""
    north = get_range_indices(times, (5, 1), (10, 1))
    south = get_range_indices(times, (11, 1), (4, 1))
    incomplete = (north == -1).any(axis=1) | (south == -1).any(axis=1)
    lo, hi = 0, north.size                     # sic: .size, as the reference initialises it
    started = False
    for y, bad in enumerate(incomplete):
        if bad and not started:
            lo = y
            continue
        started = True
        if bad:
            hi = y
            break
    years = np.arange(times[0].year, times[-1].year + 1)
    return north[lo:hi], south[lo:hi], years[lo:hi]
