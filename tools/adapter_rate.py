#!/usr/bin/env python
"""Wall time of the xarray-level drop-in calls at config C2 (Datasets in, Datasets out), split into the
device-backed core calls and the Python plumbing around them.  Reported in DESIGN.md."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hdp_amd.metric, hdp_amd.threshold
from hdp_amd import core, utils
from tests.helpers import measure_dataset

dates = utils.noleap_date_range("2001-01-01", "2010-12-31")
n_lat, n_lon, T = 180, 360, dates.size
lat = np.linspace(-90, 90, n_lat); lon = np.linspace(0, 360, n_lon, endpoint=False)
rng = np.random.default_rng(0)
layouts = {"(lon, lat, time)": ("lon", "lat", "time"), "(time, lat, lon)": ("time", "lat", "lon")}
q = np.arange(0.9, 1, 0.01)
defs = [[3, 0, 0], [3, 1, 1], [4, 0, 0], [4, 1, 1], [5, 0, 0], [5, 1, 1]]
core.compute_percentiles(np.zeros((4, T), np.float32) + rng.random((4, T), dtype=np.float32), *__import__("hdp_amd.calendar", fromlist=["x"]).window_columns(dates, 7), q)

calls = {}
for fn in ("compute_percentiles", "compute_heatwave_metrics"):
    orig = getattr(core, fn)
    def wrap(*a, _o=orig, _n=fn, **k):
        t0 = time.perf_counter(); r = _o(*a, **k); calls[_n] = time.perf_counter() - t0; return r
    setattr(core, fn, wrap)

for name, dims in layouts.items():
    shape = tuple({"lon": n_lon, "lat": n_lat, "time": T}[d] for d in dims)
    base = rng.normal(15, 4, size=shape).astype(np.float32)
    meas = base + np.float32(0.5)
    bds, mds = measure_dataset(base, lon, lat, dates, dims=dims), measure_dataset(meas, lon, lat, dates, dims=dims)
    t0 = time.perf_counter(); thr = hdp_amd.threshold.compute_thresholds(bds, q); t1 = time.perf_counter()
    met = hdp_amd.metric.compute_group_metrics(mds, thr, defs); t2 = time.perf_counter()
    print(f"{name}: compute_thresholds {t1-t0:.2f} s (core {calls['compute_percentiles']:.2f}), "
          f"compute_group_metrics {t2-t1:.2f} s (core {calls['compute_heatwave_metrics']:.2f})")
    del thr, met, bds, mds, base, meas
