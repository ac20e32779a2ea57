"""Where do the C5-shaped thresholds differ from the oracle?  HDP_LIB_PATH selects the build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from hdp_amd import core, calendar as cal
from oracle import c_oracle, hdp_oracle as orc
rng = np.random.default_rng(55)
dates = orc.noleap_date_range("2001-01-01", "2100-12-31")
T, members, n_cells = dates.size, 10, 3
x = rng.normal(12, 3, size=(members, n_cells, T)).astype(np.float32)
x += (np.arange(T, dtype=np.float32) / np.float32(36500.0))[None, None, :]
cat = np.concatenate([x[m] for m in range(members)], axis=1)
cdates = np.concatenate([dates] * members)
ti, cols = cal.window_columns(cdates, 7)
q = np.linspace(0.80, 0.99, 20)
thr = core.compute_percentiles(cat, ti, cols, q)
rows = list(range(0, 365))
win = cal.expand_window_table(ti, cols)[rows]
want = c_oracle.thresholds(cat, win, q)
got = thr[:, rows]
bad = got.view(np.uint64) != want.view(np.uint64)
print(os.environ.get("HDP_LIB_PATH", "default"), "mismatches", int(bad.sum()), "of", bad.size)
idx = np.argwhere(bad)
for i in idx[:12]:
    print(" cell,row,p", i, "got", got[tuple(i)], "want", want[tuple(i)])
if bad.any():
    print(' failing (cell,row):', sorted({(int(a), int(b)) for a, b, c in idx}))
    print(" by percentile:", bad.sum(axis=(0, 1)).tolist() if bad.ndim == 3 else bad.sum(0))
