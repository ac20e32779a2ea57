"""Thresholds of one batch on stream A beside the state machines (cells16 only, HDP_METRICS_DEBUG=8 in an ablation build)
of another batch on stream B.  HDP_DBG_LIB=<ablation build>."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hdp_amd import _lib, calendar as cal, core, utils
if os.environ.get('HDP_DBG_LIB'):
    _lib.LIB_PATH = os.environ['HDP_DBG_LIB']
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
lib = _lib.ensure_device(0)
dev = torch.device("cuda", 0)
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev, priority=int(os.environ.get("CORUN_PRIO", "0")))
T = 36500
dates = utils.noleap_date_range("2000-01-01", "2099-12-31")
ti, cols = cal.window_columns(dates, 7)
q = np.arange(0.9, 1.0, 0.01)
defs = [[3, 0, 0], [3, 1, 1], [4, 0, 0], [4, 1, 1], [5, 0, 0], [5, 1, 1]]
doy_map = cal.build_doy_map(dates)
north, south, _ = cal.hemisphere_season_tables(dates)
os.environ["HDP_METRICS_OVERLAP"] = "0"
tplan = core.ThresholdPlan(ti, cols, q, T)
mplan = core.MetricsPlan(doy_map, 365, defs, north, south, q.size)
print(tplan.describe() if hasattr(tplan, "describe") else "")
lat = torch.linspace(-60, 60, n, device=dev)
xb = torch.empty(n * T, dtype=torch.float32, device=dev)
xm = torch.empty(n * T, dtype=torch.float32, device=dev)
with torch.cuda.stream(sa):
    _lib.check(lib.hdp_generate_series_dev(xb.data_ptr(), n, T, 0, lat.data_ptr(), 0, 0.7, 0.0, sa.cuda_stream))
    _lib.check(lib.hdp_generate_series_dev(xm.data_ptr(), n, T, 0, lat.data_ptr(), 1, 0.7, 1.0 / 36500.0, sa.cuda_stream))
thr = torch.empty(n * 365 * q.size, dtype=torch.float64, device=dev)
thr2 = torch.empty(n * 365 * q.size, dtype=torch.float64, device=dev)
south_dev = (lat < 0).to(torch.uint8)
out = torch.zeros(4 * q.size * len(defs) * north.shape[0] * n, dtype=torch.int16, device=dev)
out2 = torch.zeros_like(out)
tplan.run(xb.data_ptr(), n, thr.data_ptr(), sa.cuda_stream)
torch.cuda.synchronize()
mplan.run(xm.data_ptr(), thr.data_ptr(), n, south_dev.data_ptr(), n, out.data_ptr(), sb.cuda_stream)   # full: fills the words
torch.cuda.synchronize()
def ev():
    return torch.cuda.Event(enable_timing=True)
for it in range(3):
    os.environ["HDP_METRICS_DEBUG"] = os.environ.get("CORUN_BITS", "8")
    e0, e1 = ev(), ev()
    e0.record(sb)
    mplan.run(xm.data_ptr(), thr.data_ptr(), n, south_dev.data_ptr(), n, out2.data_ptr(), sb.cuda_stream); e1.record(sb)
    torch.cuda.synchronize()
    t_cells = e0.elapsed_time(e1)
    e0, e1 = ev(), ev()
    e0.record(sa)
    tplan.run(xb.data_ptr(), n, thr2.data_ptr(), sa.cuda_stream); e1.record(sa)
    torch.cuda.synchronize()
    t_thr = e0.elapsed_time(e1)
    e0, ea, eb = ev(), ev(), ev()
    e0.record(sa); sb.wait_event(e0)
    tplan.run(xb.data_ptr(), n, thr2.data_ptr(), sa.cuda_stream); ea.record(sa)
    mplan.run(xm.data_ptr(), thr.data_ptr(), n, south_dev.data_ptr(), n, out2.data_ptr(), sb.cuda_stream); eb.record(sb)
    torch.cuda.synchronize()
    same = (os.environ.get('CORUN_BITS', '8') != '8' or bool((out == out2).all())) and bool((thr == thr2).all())
    print(f"iter {it}: alone thresholds {t_thr:.2f} ms, state machines {t_cells:.2f} ms (sum {t_thr + t_cells:.2f}); together: "
          f"thresholds done at {e0.elapsed_time(ea):.2f}, state machines at {e0.elapsed_time(eb):.2f}; results identical {same}")
    os.environ.pop("HDP_METRICS_DEBUG")
