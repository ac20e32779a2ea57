"""Feasibility: thresholds of one set of cells on stream A beside metrics of another set on stream B."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hdp_amd import _lib, calendar as cal, core, utils
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
if os.environ.get('HDP_DBG_LIB'):
    _lib.LIB_PATH = os.environ['HDP_DBG_LIB']
lib = _lib.ensure_device(0)
dev = torch.device("cuda", 0)
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
T = 36500
dates = utils.noleap_date_range("2000-01-01", "2099-12-31")
ti, cols = cal.window_columns(dates, 7)
q = np.arange(0.9, 1.0, 0.01)
defs = [[3, 0, 0], [3, 1, 1], [4, 0, 0], [4, 1, 1], [5, 0, 0], [5, 1, 1]]
doy_map = cal.build_doy_map(dates)
north, south, _ = cal.hemisphere_season_tables(dates)
tplan = core.ThresholdPlan(ti, cols, q, T)
mplan = core.MetricsPlan(doy_map, 365, defs, north, south, q.size)
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
tplan.run(xb.data_ptr(), n, thr.data_ptr(), sa.cuda_stream)
mplan.reserve(n)
torch.cuda.synchronize()
def timed(fn):
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for it in range(3):
    t_thr = timed(lambda: tplan.run(xb.data_ptr(), n, thr2.data_ptr(), sa.cuda_stream))
    t_met = timed(lambda: mplan.run(xm.data_ptr(), thr.data_ptr(), n, south_dev.data_ptr(), n, out.data_ptr(), sb.cuda_stream))
    def both():
        tplan.run(xb.data_ptr(), n, thr2.data_ptr(), sa.cuda_stream)
        mplan.run(xm.data_ptr(), thr.data_ptr(), n, south_dev.data_ptr(), n, out.data_ptr(), sb.cuda_stream)
    t_both = timed(both)
    def both_rev():
        mplan.run(xm.data_ptr(), thr.data_ptr(), n, south_dev.data_ptr(), n, out.data_ptr(), sb.cuda_stream)
        tplan.run(xb.data_ptr(), n, thr2.data_ptr(), sa.cuda_stream)
    t_rev = timed(both_rev)
    e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    torch.cuda.synchronize()
    e0.record(sa); sb.wait_event(e0)
    tplan.run(xb.data_ptr(), n, thr2.data_ptr(), sa.cuda_stream); ea.record(sa)
    mplan.run(xm.data_ptr(), thr.data_ptr(), n, south_dev.data_ptr(), n, out.data_ptr(), sb.cuda_stream); eb.record(sb)
    torch.cuda.synchronize()
    print(f"   thr-first: thresholds done at {e0.elapsed_time(ea):.2f} ms, metrics done at {e0.elapsed_time(eb):.2f} ms")
    e0.record(sb); sa.wait_event(e0)
    mplan.run(xm.data_ptr(), thr.data_ptr(), n, south_dev.data_ptr(), n, out.data_ptr(), sb.cuda_stream); eb.record(sb)
    tplan.run(xb.data_ptr(), n, thr2.data_ptr(), sa.cuda_stream); ea.record(sa)
    torch.cuda.synchronize()
    print(f"   met-first: thresholds done at {e0.elapsed_time(ea):.2f} ms, metrics done at {e0.elapsed_time(eb):.2f} ms")
    print(f"iter {it}: thresholds {t_thr:.2f} ms, metrics {t_met:.2f} ms, sum {t_thr + t_met:.2f}; concurrent (thr first) {t_both:.2f}, (metrics first) {t_rev:.2f}")
