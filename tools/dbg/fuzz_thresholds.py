#!/usr/bin/env python
"""Extended fuzz of the thresholds plans against the C oracle (development aid; the committed tests run 20 seeds):
random record lengths, window radii and quantile SETS (tails, straddling the median, spread, duplicates, 0 and 1)."""
import sys, os, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hdp_amd import calendar as cal, core
from oracle import c_oracle, hdp_oracle as orc

def same(a, b):
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))

bad = 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
for seed in range(n):
    rng = np.random.default_rng(50000 + seed)
    years = int(rng.choice([3, 9, 17, 40, 64, 65, 80, 100, 100, 100, 128]))
    dates = orc.noleap_date_range("0001-01-01", f"{years:04d}-12-31")
    T = dates.size
    radius = int(rng.choice([0, 1, 3, 7, 7, 7, 8]))
    kind = int(rng.integers(0, 6))
    P = int(rng.integers(1, 13))
    if kind == 0: q = np.sort(rng.uniform(0.85, 1.0, P))
    elif kind == 1: q = np.sort(rng.uniform(0.0, 0.15, P))
    elif kind == 2: q = np.sort(rng.uniform(0.4, 0.6, P))
    elif kind == 3: q = np.sort(rng.uniform(0.0, 1.0, P))
    elif kind == 4: q = np.sort(np.concatenate([rng.uniform(0.0, 0.1, P // 2 + 1), rng.uniform(0.9, 1.0, P // 2 + 1)]))
    else: q = np.sort(rng.choice([0.0, 0.5, 1.0, 0.25, 0.75, 0.5000001, 0.4999999], P))
    ncell = int(rng.integers(1, 6))
    x = (15 + 6 * np.sin(2 * np.pi * np.arange(T) / 365.0)[None, :] + rng.normal(0, 2, size=(ncell, T))).astype(np.float32)
    if rng.random() < 0.3: x = np.round(x)
    if rng.random() < 0.2: x[0, rng.integers(0, T, 3)] = np.inf
    if rng.random() < 0.2: x[0, rng.integers(0, T, 3)] = -np.inf
    if rng.random() < 0.1: x[-1, rng.integers(0, T)] = np.nan
    ti, cols = cal.window_columns(dates, radius)
    win = cal.expand_window_table(ti, cols)
    with np.errstate(invalid="ignore"):
        want = c_oracle.thresholds(x, win, q)
    got = core.compute_percentiles(x, ti, cols, q)
    if not same(got, want):
        bad += 1
        print("MISMATCH seed", seed, years, radius, kind, q)
print("fuzz done:", n, "cases,", bad, "mismatches")
