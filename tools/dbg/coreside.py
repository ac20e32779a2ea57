"""Which workgroups does the hardware place beside the persistent thresholds kernel?  (probe kernel on a second stream)"""
import os, sys, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hdp_amd import _lib, calendar as cal, core, utils
if os.environ.get('HDP_DBG_LIB'):
    _lib.LIB_PATH = os.environ['HDP_DBG_LIB']
lib = _lib.ensure_device(0)
probe = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libprobe.so"))
probe.probe_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p]
n = 65536
dev = torch.device("cuda", 0)
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
T = 36500
dates = utils.noleap_date_range("2000-01-01", "2099-12-31")
ti, cols = cal.window_columns(dates, 7)
q = np.arange(0.9, 1.0, 0.01)
tplan = core.ThresholdPlan(ti, cols, q, T)
lat = torch.linspace(-60, 60, n, device=dev)
xb = torch.empty(n * T, dtype=torch.float32, device=dev)
_lib.check(lib.hdp_generate_series_dev(xb.data_ptr(), n, T, 0, lat.data_ptr(), 0, 0.7, 0.0, sa.cuda_stream))
thr = torch.empty(n * 365 * q.size, dtype=torch.float64, device=dev)
sink = torch.zeros(4, dtype=torch.int32, device=dev)
tplan.run(xb.data_ptr(), n, thr.data_ptr(), sa.cuda_stream)
torch.cuda.synchronize()
print(tplan.describe())
for var, grid, threads, lds in [(0, 256, 256, 0), (0, 256, 256, 17000), (0, 256, 256, 20480), (72, 256, 256, 0), (72, 256, 256, 20480), (80, 256, 256, 0), (88, 256, 256, 0), (1, 256, 256, 0), (1, 256, 256, 20480), (72, 2560, 256, 20480), (0, 2560, 256, 0)]:
    e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    torch.cuda.synchronize()
    e0.record(sa); sb.wait_event(e0)
    tplan.run(xb.data_ptr(), n, thr.data_ptr(), sa.cuda_stream); ea.record(sa)
    rc = probe.probe_launch(var, grid, threads, lds, 100000, sink.data_ptr(), sb.cuda_stream); eb.record(sb)   # 1 ms of spinning
    torch.cuda.synchronize()
    print(f"probe variant {var}, grid {grid}, {threads} threads, {lds} B LDS (rc {rc}): thresholds done at {e0.elapsed_time(ea):.2f} ms, probe done at {e0.elapsed_time(eb):.2f} ms")
