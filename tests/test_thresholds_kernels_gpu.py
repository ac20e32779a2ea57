"""Every thresholds kernel variant against the C oracle and against each other (bit-exact).

The library picks a kernel from the plan: the lane-per-column kernel (up to 100 samples per column and 16 window
columns: one lane sorts one column in registers by a merge-exchange network; whole-cell or blocked form; HDP_THR_LANE=0
turns it off) or the one-workgroup-per-cell kernel (every other plan: more than 100 samples per column, wider
windows), which finishes a block either with the merge or, when the requested ranks lie deep (HDP_THR_SELECT), with a
rank selection per (row, requested rank).  (The round-1 pipelined kernel that used to sit between the two is gone.)
The environment switches are read when a plan is created (core.compute_percentiles makes one per call), so one
process can run them all on the same input.
"""
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from hdp_amd import calendar as cal  # noqa: E402
from hdp_amd import core  # noqa: E402
from oracle import c_oracle  # noqa: E402
from oracle import hdp_oracle as orc  # noqa: E402


def same_f64(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


CASES = [
    # (id, first date, last date, window radius, quantiles, cells, sprinkle special values)
    ("S100-vec-ng4-top", "0001-01-01", "0100-12-31", 7, list(np.arange(0.9, 1.0, 0.01)), 5, False),
    ("S100-vec-ng4-both", "0001-01-01", "0100-12-31", 7, [0.0, 0.03, 0.5, 0.97, 1.0], 3, True),
    ("S70-ragged-novec", "0001-01-01", "0070-03-17", 7, [0.1, 0.9, 0.99], 4, True),
    ("S40-lpc8-ng2", "0001-01-01", "0040-12-31", 3, [0.02, 0.5, 0.97], 6, False),
    ("S20-lpc4-ng1", "0001-01-01", "0020-12-31", 1, [0.25, 0.95], 9, True),
    ("S12-lpc2-generic", "0001-01-01", "0012-12-31", 10, [0.05, 0.9, 0.99], 7, False),
    ("S5-lpc1-ng4", "0001-01-01", "0005-12-31", 7, [0.0, 0.9, 1.0], 11, True),
    ("S3-lpc1-ragged", "0001-01-01", "0003-02-10", 7, [0.5, 0.9], 8, False),
    # columns that fill their register slots exactly (S = 8 * LPC): no padding slot lands on the sentinel
    ("S8-lpc1-full", "0001-01-01", "0008-12-31", 7, [0.1, 0.9], 5, True),
    ("S16-lpc2-full", "0001-01-01", "0016-12-31", 2, [0.0, 0.5, 1.0], 5, False),
    ("S64-lpc8-full", "0001-01-01", "0064-12-31", 7, [0.9, 0.99], 3, False),
    ("S128-lpc16-full", "0001-01-01", "0128-12-31", 7, [0.05, 0.95], 2, True),
    # more than 128 samples per column: LDS columns, lane-major wave sort (4, 8, 16 keys per lane), merge or selection
    ("S150-epl4-ragged", "0001-01-01", "0150-06-30", 7, [0.0, 0.02, 0.5, 0.93, 1.0], 3, True),
    ("S260-epl8", "0001-01-01", "0260-12-31", 3, [0.2, 0.8, 0.99], 3, True),
    ("S600-epl16-w5", "0001-01-01", "0600-12-31", 2, [0.001, 0.35, 0.65, 0.999], 2, False),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_all_thresholds_kernels_agree_with_the_oracle(case, monkeypatch):
    _, d0, d1, radius, q, n_cells, special = case
    rng = np.random.default_rng(zlib.crc32(case[0].encode()))
    dates = orc.noleap_date_range(d0, d1)
    T = dates.size
    t = np.arange(T)
    x = (15 + 8 * np.sin(2 * np.pi * t / 365.0)[None, :] + rng.normal(0, 2.5, size=(n_cells, T))).astype(np.float32)
    x[-1] = np.round(x[-1])                       # many exact ties
    if special:
        x[0, rng.integers(0, T, 3)] = np.inf
        x[0, rng.integers(0, T, 2)] = -np.inf
        x[1, rng.integers(0, T)] = np.nan
        x[min(2, n_cells - 1), T - 1] = -np.inf   # the sample a -1 padded slot reads
    ti, cols = cal.window_columns(dates, radius)
    win = cal.expand_window_table(ti, cols)
    with np.errstate(invalid="ignore"):
        want = c_oracle.thresholds(x, win, q)

    def run(**env):
        for k in ("HDP_THR_PIPE", "HDP_THR_VEC", "HDP_THR_SELECT", "HDP_THR_LANE", "HDP_THR_WHOLE", "HDP_THR_DUAL"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        return core.compute_percentiles(x, ti, cols, q)

    got = run()                                        # lane-per-column kernel where the plan allows it
    assert same_f64(got, want)
    assert same_f64(run(HDP_THR_WHOLE="0"), want)     # its blocked form where the default is the whole-cell one
    # blocked form with the two walks of a row on different merging waves (plans with a top AND a bottom walk), and without
    assert same_f64(run(HDP_THR_WHOLE="0", HDP_THR_DUAL="1"), want)
    assert same_f64(run(HDP_THR_WHOLE="0", HDP_THR_DUAL="0"), want)
    assert same_f64(run(HDP_THR_LANE="0"), want)      # one workgroup per cell, merge or selection as the plan chooses
    assert same_f64(run(HDP_THR_PIPE="0", HDP_THR_SELECT="0"), want)   # one workgroup per cell, merge
    assert same_f64(run(HDP_THR_PIPE="0", HDP_THR_SELECT="1"), want)   # same, rank selection per (row, rank)


def test_tiered_image_deep_columns_reach_the_global_tail():
    """Whole-cell lane kernel: a column keeps its top 60 samples in LDS and the rest in a global tail.  With a steep
    seasonal slope and little noise the warmest column of a window supplies (almost) all of its 100 samples to the
    top 10 %, so the merges walk deep into the tails -- and must still match the oracle bit for bit."""
    rng = np.random.default_rng(11)
    dates = orc.noleap_date_range("0001-01-01", "0100-12-31")
    T = dates.size
    t = np.arange(T)
    n_cells = 4
    x = (20 * np.sin(2 * np.pi * t / 365.0)[None, :] + rng.normal(0, 0.01, size=(n_cells, T))).astype(np.float32)
    x[1] = np.round(x[1], 1)          # ties
    x[2, ::7] = -np.inf               # special values in the tails
    x[3] += (t / 3650.0).astype(np.float32)   # trend: recent years on top of every column
    ti, cols = cal.window_columns(dates, 7)
    q = list(np.arange(0.9, 1.0, 0.01))
    win = cal.expand_window_table(ti, cols)
    with np.errstate(invalid="ignore"):
        want = c_oracle.thresholds(x, win, q)
    assert same_f64(core.compute_percentiles(x, ti, cols, q), want)


def test_persistent_kernel_many_cells_vs_single_cell_launches():
    """Persistent workgroups walk cells with a stride; every cell must come out as if it were alone
    (no state leaks between the items of a workgroup, odd cell counts, more workgroups than cells)."""
    rng = np.random.default_rng(77)
    dates = orc.noleap_date_range("0001-01-01", "0030-12-31")
    x = rng.normal(0, 3, size=(1543, dates.size)).astype(np.float32)
    ti, cols = cal.window_columns(dates, 7)
    q = [0.9, 0.95, 0.99]
    got = core.compute_percentiles(x, ti, cols, q)
    for c in (0, 1, 511, 512, 1023, 1542):
        assert same_f64(got[c:c + 1], core.compute_percentiles(x[c:c + 1], ti, cols, q))
    win = cal.expand_window_table(ti, cols)
    sel = rng.choice(x.shape[0], 24, replace=False)
    assert same_f64(got[sel], c_oracle.thresholds(x[sel], win, q))


@pytest.mark.parametrize("seed", range(20))
def test_random_calendars_windows_and_quantiles(seed, monkeypatch):
    """Seeded random configurations -- years of record (1..260 samples per day of year), a ragged last year,
    window radius 0..12 (up to 25 columns: the generic merge; selection only up to 16), 1..6 quantiles anywhere
    in [0, 1], ties and special values -- every kernel variant against the C oracle."""
    rng = np.random.default_rng(1000 + seed)
    years = int(rng.choice([1, 2, 3, 7, 11, 19, 33, 64, 65, 100, 129, 180, 260]))
    end_month, end_day = (12, 31) if rng.random() < 0.5 else (int(rng.integers(1, 12)), int(rng.integers(1, 28)))
    dates = orc.noleap_date_range("0001-01-01", f"{years:04d}-{end_month:02d}-{end_day:02d}")
    T = dates.size
    if T < 365:
        dates = orc.noleap_date_range("0001-01-01", "0001-12-31")
        T = dates.size
    radius = int(rng.integers(0, 13))
    n_q = int(rng.integers(1, 7))
    q = np.sort(rng.random(n_q))
    if rng.random() < 0.3:
        q[0] = 0.0
    if rng.random() < 0.3:
        q[-1] = 1.0
    n_cells = int(rng.integers(1, 10))
    x = rng.normal(0, 5, size=(n_cells, T)).astype(np.float32)
    if rng.random() < 0.5:
        x = np.round(x)                               # heavy ties
    if rng.random() < 0.4:
        x[0, rng.integers(0, T, 4)] = np.inf
        x[0, rng.integers(0, T, 4)] = -np.inf
    if rng.random() < 0.3:
        x[-1, rng.integers(0, T)] = np.nan
    ti, cols = cal.window_columns(dates, radius)
    win = cal.expand_window_table(ti, cols)
    with np.errstate(invalid="ignore"):
        want = c_oracle.thresholds(x, win, q)
    for env in ({}, {"HDP_THR_WHOLE": "0"}, {"HDP_THR_LANE": "0"},
                {"HDP_THR_PIPE": "0", "HDP_THR_SELECT": "0"}, {"HDP_THR_PIPE": "0", "HDP_THR_SELECT": "1"},
                {"HDP_THR_WHOLE": "0", "HDP_THR_DUAL": "1"}, {"HDP_THR_WHOLE": "0", "HDP_THR_DUAL": "0"}):
        for k in ("HDP_THR_PIPE", "HDP_THR_VEC", "HDP_THR_SELECT", "HDP_THR_LANE", "HDP_THR_WHOLE", "HDP_THR_DUAL"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        assert same_f64(core.compute_percentiles(x, ti, cols, q), want), (env, years, radius, q)
