"""Property tests (CPU, hypothesis) of the two reformulations the HIP kernels rely on, against the
oracle's literal restatement of the reference:

  * the run-streaming form of index_heatwaves (process hot RUNS; gap rule at the next run start)
    equals the reference's edge-list loop (metric.py:27-58) for any series and definition;
  * season metrics from runs + "same id as previous labelled run in this season" bookkeeping equal
    HWF/HWN/HWD/trunc(HWA) computed from the id series (metric.py:63-172) for disjoint increasing
    seasons -- including HWA == HWF // HWN;
  * f32 > f64 comparison equals f32 > round_down_f32(f64) (the threshold staging trick).
"""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import hdp_oracle as orc


def streaming_ids(hot, min_dur, max_break, max_subs):
    n = len(hot)
    ids = np.zeros(n, dtype=np.int64)
    in_hw, subs, cur, e_prev = False, 0, 0, -10**9
    t = 0
    runs = []
    while t < n:
        if not hot[t]:
            t += 1
            continue
        s = t
        while t < n and hot[t]:
            t += 1
        if s - e_prev > max_break:
            in_hw = False
        length = t - s
        label = False
        if not in_hw:
            if length >= min_dur:
                cur += 1; in_hw = True; label = True
        elif subs < max_subs:
            subs += 1; label = True
        else:
            if length >= min_dur:
                cur += 1; label = True
            else:
                in_hw = False
            subs = 0
        if label:
            ids[s:t] = cur
            runs.append((s, t, cur))
        e_prev = t
    return ids, runs


series = st.lists(st.booleans(), min_size=1, max_size=120)
defn = st.tuples(st.integers(0, 6), st.integers(0, 4), st.integers(0, 4))


@settings(max_examples=400, deadline=None)
@given(series, defn)
def test_streaming_runs_equal_edge_list_form(hot, d):
    hot = np.array(hot, dtype=bool)
    ids, _ = streaming_ids(hot, *d)
    assert np.array_equal(ids, orc.index_heatwaves(hot, *d))


@settings(max_examples=300, deadline=None)
@given(series, defn, st.lists(st.integers(0, 120), min_size=2, max_size=8))
def test_season_metrics_from_runs(hot, d, cuts):
    hot = np.array(hot, dtype=bool)
    n = hot.size
    cuts = sorted(set(min(c, n) for c in cuts))
    seasons = [(a, b) for a, b in zip(cuts[::2], cuts[1::2]) if b > a]
    if not seasons:
        return
    ids, runs = streaming_ids(hot, *d)
    want = np.stack([orc.heatwave_frequency(ids, seasons), orc.heatwave_number(ids, seasons),
                     orc.heatwave_duration(ids, seasons),
                     orc.heatwave_average(ids, seasons).astype(np.int64)])
    got = np.zeros((4, len(seasons)), dtype=np.int64)
    for y, (a, b) in enumerate(seasons):
        hwf = hwn = hwd = cur = 0
        last = 0
        for s, e, k in runs:
            days = min(e, b) - max(s, a)
            if days <= 0:
                continue
            hwf += days
            if k != last:
                hwn += 1; cur = days; last = k
            else:
                cur += days
            hwd = max(hwd, cur)
        got[:, y] = (hwf, hwn, hwd, hwf // hwn if hwn else 0)
    assert np.array_equal(got, want)


@settings(max_examples=500, deadline=None)
@given(st.floats(allow_nan=False, width=64), st.floats(allow_nan=False, width=32))
def test_round_down_threshold_preserves_strict_compare(thr, x):
    x32 = np.float32(x)
    with np.errstate(over="ignore"):
        r = np.float32(thr)
        if np.float64(r) > thr:
            r = np.nextafter(r, np.float32(-np.inf))
    assert bool(np.float64(x32) > thr) == bool(x32 > r)
