#!/usr/bin/env python
"""PCIe-inclusive throughput of the host-pointer API (numpy in / numpy out) at config C2,
time-contiguous and time-major inputs.  Reported in DESIGN.md; never bench.py's `value`."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hdp_amd import calendar as cal, core, utils

dates = utils.noleap_date_range("2001-01-01", "2010-12-31")
T, n = dates.size, 180 * 360
rng = np.random.default_rng(0)
x = rng.normal(15, 4, size=(n, T)).astype(np.float32)
xt = np.ascontiguousarray(x.T)
ti, cols = cal.window_columns(dates, 7)
q = np.arange(0.9, 1, 0.01)
defs = [[3, 0, 0], [3, 1, 1], [4, 0, 0], [4, 1, 1], [5, 0, 0], [5, 1, 1]]
dm = cal.build_doy_map(dates); north, south, _ = cal.hemisphere_season_tables(dates)
hemi = (np.arange(n) % 2).astype(np.uint8)
core.compute_percentiles(x[:64], ti, cols, q)  # warm-up (library init)
thr = met = None
for name, arr in (("time-contiguous [cells][T]", x), ("time-major [T][cells]", xt.T)):
    del thr, met   # releasing 2 GB of results is not part of the next call
    t0 = time.perf_counter(); thr = core.compute_percentiles(arr, ti, cols, q); t1 = time.perf_counter()
    met = core.compute_heatwave_metrics(arr, thr, dm, defs, north, south, hemi); t2 = time.perf_counter()
    print(f"{name}: thresholds {n*T/(t1-t0):.3e} cell-days/s ({t1-t0:.2f} s), metrics {n*T/(t2-t1):.3e} cell-days/s ({t2-t1:.2f} s)")
