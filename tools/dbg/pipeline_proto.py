"""Prototype of the batch pipeline: thresholds of batch b+1 (stream A) beside the metrics of batch b (stream B)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hdp_amd import _lib, calendar as cal, core, utils
if os.environ.get('HDP_DBG_LIB'):
    _lib.LIB_PATH = os.environ['HDP_DBG_LIB']
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
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
mplans = [core.MetricsPlan(doy_map, 365, defs, north, south, q.size) for _ in range(2)]
print(tplan.describe())
lat = torch.linspace(-60, 60, n, device=dev)
xb = torch.empty(n * T, dtype=torch.float32, device=dev)
xm = torch.empty(n * T, dtype=torch.float32, device=dev)
_lib.check(lib.hdp_generate_series_dev(xb.data_ptr(), n, T, 0, lat.data_ptr(), 0, 0.7, 0.0, sa.cuda_stream))
_lib.check(lib.hdp_generate_series_dev(xm.data_ptr(), n, T, 0, lat.data_ptr(), 1, 0.7, 1.0 / 36500.0, sa.cuda_stream))
PN = 365 * q.size
Y = north.shape[0]
thr = torch.empty(n * PN, dtype=torch.float64, device=dev)
south_dev = (lat < 0).to(torch.uint8)
per = n // nb
outs = [torch.zeros(4 * q.size * len(defs) * Y * per, dtype=torch.int16, device=dev) for _ in range(nb)]
ref = None
for p in mplans:
    p.reserve(per)
torch.cuda.synchronize()
def serial():
    for b in range(nb):
        tplan.run(xb.data_ptr() + b * per * T * 4, per, thr.data_ptr() + b * per * PN * 8, sa.cuda_stream)
    for b in range(nb):
        mplans[0].run(xm.data_ptr() + b * per * T * 4, thr.data_ptr() + b * per * PN * 8, per, south_dev.data_ptr() + b * per, per,
                      outs[b].data_ptr(), sa.cuda_stream)
def piped():
    evs = [torch.cuda.Event() for _ in range(nb)]
    e0 = torch.cuda.Event(); e0.record(sa); sb.wait_event(e0)
    for b in range(nb):
        tplan.run(xb.data_ptr() + b * per * T * 4, per, thr.data_ptr() + b * per * PN * 8, sa.cuda_stream)
        evs[b].record(sa)
        sb.wait_event(evs[b])
        mplans[b & 1].run(xm.data_ptr() + b * per * T * 4, thr.data_ptr() + b * per * PN * 8, per, south_dev.data_ptr() + b * per, per,
                          outs[b].data_ptr(), sb.cuda_stream)
    e1 = torch.cuda.Event(); e1.record(sb); sa.wait_event(e1)
def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for it in range(3):
    ts = timed(serial)
    chk = [o.clone() for o in outs] if it == 0 else chk
    for o in outs: o.zero_()
    tp = timed(piped)
    same = all(bool((a == b).all()) for a, b in zip(chk, outs))
    print(f"iter {it}: {n} cells in {nb} batches: serial {ts:.1f} ms, pipelined {tp:.1f} ms ({ts / tp:.3f}x), identical {same}")
