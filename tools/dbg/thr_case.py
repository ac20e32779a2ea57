import sys, zlib, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hdp_amd import calendar as cal, core
from oracle import c_oracle, hdp_oracle as orc

def run(name, d0, d1, radius, q, n_cells, special):
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    dates = orc.noleap_date_range(d0, d1)
    T = dates.size
    t = np.arange(T)
    x = (15 + 8 * np.sin(2 * np.pi * t / 365.0)[None, :] + rng.normal(0, 2.5, size=(n_cells, T))).astype(np.float32)
    x[-1] = np.round(x[-1])
    if special:
        x[0, rng.integers(0, T, 3)] = np.inf
        x[0, rng.integers(0, T, 2)] = -np.inf
        x[1, rng.integers(0, T)] = np.nan
        x[min(2, n_cells - 1), T - 1] = -np.inf
    ti, cols = cal.window_columns(dates, radius)
    win = cal.expand_window_table(ti, cols)
    with np.errstate(invalid="ignore"):
        want = c_oracle.thresholds(x, win, q)
    got = core.compute_percentiles(x, ti, cols, q)
    bad = ~((got == want) | (np.isnan(got) & np.isnan(want)))
    print(name, "mismatches:", int(bad.sum()), "of", bad.size)
    idx = np.argwhere(bad)
    for c in range(n_cells):
        for p in range(len(q)):
            n = int(bad[c, :, p].sum())
            if n:
                d = np.argwhere(bad[c, :, p])[:, 0]
                print("  cell", c, "q", q[p], "rows bad", n, "first", d[:8], "got", got[c, d[0], p], "want", want[c, d[0], p])

run("S100-vec-ng4-both", "0001-01-01", "0100-12-31", 7, [0.0, 0.03, 0.5, 0.97, 1.0], 3, True)
run("nospecial", "0001-01-01", "0100-12-31", 7, [0.0, 0.03, 0.5, 0.97, 1.0], 3, False)
run("S70", "0001-01-01", "0070-03-17", 7, [0.1, 0.9, 0.99], 4, True)
run("S5", "0001-01-01", "0005-12-31", 7, [0.0, 0.9, 1.0], 11, True)
run("S100-vec-ng4-top", "0001-01-01", "0100-12-31", 7, list(np.arange(0.9, 1.0, 0.01)), 5, False)
