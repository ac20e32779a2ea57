"""Time hdp_metrics_f32_dev on n series of the C3 calendar (thresholds from the thresholds kernel) for the kernel
variant the environment selects; prints a checksum of the metrics."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hdp_amd import _lib, calendar as cal, core, utils
if os.environ.get('HDP_DBG_LIB'):
    _lib.LIB_PATH = os.environ['HDP_DBG_LIB']
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
years = 100
lib = _lib.ensure_device(0)
dev = torch.device("cuda", 0)
ts = torch.cuda.Stream(dev); torch.cuda.set_stream(ts); stream = ts.cuda_stream
T = years * 365
dates = utils.noleap_date_range("2000-01-01", f"{2000 + years - 1}-12-31")
ti, cols = cal.window_columns(dates, 7)
q = np.arange(0.9, 1.0, 0.01)
defs = [[3, 0, 0], [3, 1, 1], [4, 0, 0], [4, 1, 1], [5, 0, 0], [5, 1, 1]]
doy_map = cal.build_doy_map(dates)
north, south, _ = cal.hemisphere_season_tables(dates)
tplan = core.ThresholdPlan(ti, cols, q, T)
mplan = core.MetricsPlan(doy_map, 365, defs, north, south, q.size)
print(mplan.describe())
lat = torch.linspace(-60, 60, n, device=dev)
xb = torch.empty(n * T, dtype=torch.float32, device=dev)
xm = torch.empty(n * T, dtype=torch.float32, device=dev)
_lib.check(lib.hdp_generate_series_dev(xb.data_ptr(), n, T, 0, lat.data_ptr(), 0, 0.7, 0.0, stream))
_lib.check(lib.hdp_generate_series_dev(xm.data_ptr(), n, T, 0, lat.data_ptr(), 1, 0.7, 1.0 / 36500.0, stream))
thr = torch.empty(n * 365 * q.size, dtype=torch.float64, device=dev)
south_dev = (lat < 0).to(torch.uint8)
out = torch.zeros(4 * q.size * len(defs) * north.shape[0] * n, dtype=torch.int16, device=dev)
tplan.run(xb.data_ptr(), n, thr.data_ptr(), stream)
mplan.reserve(n)
for it in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    mplan.run(xm.data_ptr(), thr.data_ptr(), n, south_dev.data_ptr(), n, out.data_ptr(), stream)
    e1.record()
    torch.cuda.synchronize()
    print(f"iter {it}: {e0.elapsed_time(e1):.3f} ms for {n} series  (checksum {int(out.to(torch.int64).sum())})")
