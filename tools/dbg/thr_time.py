"""Time hdp_thresholds_f32_dev on n cells of the C3 calendar for the kernel variant the environment selects."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hdp_amd import _lib, calendar as cal, core, utils
if os.environ.get('HDP_DBG_LIB'):
    _lib.LIB_PATH = os.environ['HDP_DBG_LIB']

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
years = int(sys.argv[2]) if len(sys.argv) > 2 else 100
members = int(sys.argv[3]) if len(sys.argv) > 3 else 1     # C5: 10 members appended along time (S = members * years)
lib = _lib.ensure_device(0)
dev = torch.device("cuda", 0)
ts = torch.cuda.Stream(dev)
torch.cuda.set_stream(ts)
stream = ts.cuda_stream
T = years * 365 * members
dates = utils.noleap_date_range("2000-01-01", f"{2000 + years - 1}-12-31")
ti, cols = cal.window_columns(np.concatenate([dates] * members), 7)
q = np.arange(0.9, 1.0, 0.01) if members == 1 else np.linspace(0.80, 0.99, 20)
if os.environ.get('HDP_QSET') == 'median': q = np.linspace(0.455, 0.545, 10)
if os.environ.get('HDP_QSET') == 'spread': q = np.linspace(0.05, 0.95, 10)
plan = core.ThresholdPlan(ti, cols, q, T)
print(plan.describe() if hasattr(plan, "describe") else "")
x = torch.empty(n * T, dtype=torch.float32, device=dev)
lat = torch.linspace(-60, 60, n, device=dev)
_lib.check(lib.hdp_generate_series_dev(x.data_ptr(), n, T, 0, lat.data_ptr(), 0, 0.7, 0.0, stream))
out = torch.empty(n * 365 * q.size, dtype=torch.float64, device=dev)
for it in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    plan.run(x.data_ptr(), n, out.data_ptr(), stream)
    e1.record()
    torch.cuda.synchronize()
    print(f"iter {it}: {e0.elapsed_time(e1):.3f} ms for {n} cells  (checksum {float(out.sum()):.6f})")
del plan
