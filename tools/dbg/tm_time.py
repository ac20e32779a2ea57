"""Transposition cost of the time-major thresholds call: tm call (HDP_TM_SERIAL=1: copy, then kernel) minus the
series-major call on the same cells.  python tools/dbg/tm_time.py [cells] [years]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hdp_amd import _lib, calendar as cal, core, utils
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
years = int(sys.argv[2]) if len(sys.argv) > 2 else 100
lib = _lib.ensure_device(0)
dev = torch.device("cuda", 0)
ts = torch.cuda.Stream(dev); torch.cuda.set_stream(ts); stream = ts.cuda_stream
T = years * 365
dates = utils.noleap_date_range("2000-01-01", f"{2000 + years - 1}-12-31")
ti, cols = cal.window_columns(dates, 7)
q = np.arange(0.9, 1.0, 0.01)
plan = core.ThresholdPlan(ti, cols, q, T)
x = torch.empty(n * T, dtype=torch.float32, device=dev)
lat = torch.linspace(-60, 60, n, device=dev)
_lib.check(lib.hdp_generate_series_dev(x.data_ptr(), n, T, 0, lat.data_ptr(), 0, 0.7, 0.0, stream))
src = x.view(n, T).t().contiguous()
out = torch.empty(n * 365 * q.size, dtype=torch.float64, device=dev)
out2 = torch.empty_like(out)
def timed(f):
    best = 1e9
    for it in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
a = timed(lambda: plan.run(x.data_ptr(), n, out.data_ptr(), stream))
b = timed(lambda: _lib.check(lib.hdp_thresholds_f32_tm_dev(plan.handle, src.data_ptr(), n, n, out2.data_ptr(), stream)))
gb = 2 * n * T * 4 / 1e9
print(f"T={T} ({T*4} B per series, mod 128 = {T*4 % 128}) cells={n}: series-major {a:.3f} ms, time-major {b:.3f} ms, "
      f"difference {b-a:.3f} ms = {gb/(b-a):.0f} GB/s of copy traffic; identical {bool(torch.equal(out, out2))}")
del plan
