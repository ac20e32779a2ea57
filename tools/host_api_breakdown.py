#!/usr/bin/env python
"""Where the host-pointer API spends its time at config C2 (numpy in / numpy out): plan creation, upload,
kernels, download.  Uses the device-resident entry points for the pieces; reported in DESIGN.md."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hdp_amd import _lib, calendar as cal, core, utils

lib = _lib.ensure_device()
dates = utils.noleap_date_range("2001-01-01", "2010-12-31")
T, n = dates.size, 180 * 360
rng = np.random.default_rng(0)
x = rng.normal(15, 4, size=(n, T)).astype(np.float32)
ti, cols = cal.window_columns(dates, 7)
q = np.arange(0.9, 1, 0.01)
core.compute_percentiles(x[:64], ti, cols, q)


def timed(f, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); lib.hdp_sync(None); best = min(best, time.perf_counter() - t0)
    return best, r


t_all, thr = timed(lambda: core.compute_percentiles(x, ti, cols, q))
t_plan, plan = timed(lambda: core.ThresholdPlan(ti, cols, q, T))
dx = core.DeviceArray((n, T), np.float32)
dout = core.DeviceArray((n, q.size, 365), np.float64)
t_h2d, _ = timed(lambda: _lib.check(lib.hdp_memcpy_h2d(dx.ptr, x.ctypes.data, x.nbytes)))
t_k, _ = timed(lambda: plan.run(dx.ptr, n, dout.ptr))
host_out = np.empty((n, q.size, 365), np.float64)
t_d2h, _ = timed(lambda: _lib.check(lib.hdp_memcpy_d2h(host_out.ctypes.data, dout.ptr, host_out.nbytes)))
print(f"C2 thresholds host API: total {t_all*1e3:.1f} ms = plan {t_plan*1e3:.1f} + H2D {t_h2d*1e3:.1f} ({x.nbytes/t_h2d/1e9:.1f} GB/s) "
      f"+ kernel {t_k*1e3:.1f} + D2H {t_d2h*1e3:.1f} ({host_out.nbytes/t_d2h/1e9:.1f} GB/s) + rest")
