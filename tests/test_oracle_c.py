"""The C restatement (oracle/hdp_oracle.c, the CPU baseline bench.py times) against the
Python oracle and the reference-generated fixtures.  CPU only."""
import os

import numpy as np
import pytest

from oracle import c_oracle
from oracle import hdp_oracle as orc


def same(a, b):
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


@pytest.mark.parametrize("tag", ["full3", "ragged"])
def test_c_oracle_small_workflow(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "small_workflow.npz"))
    thr = c_oracle.thresholds(g[f"{tag}_baseline"], g[f"{tag}_window"], g["percentiles"])
    assert same(thr, orc.compute_thresholds_cells(g[f"{tag}_baseline"], g[f"{tag}_window"], g["percentiles"]))
    np.testing.assert_allclose(thr, g[f"{tag}_thresholds"], rtol=1e-6, atol=0)
    met = c_oracle.metrics(g[f"{tag}_measure"], g[f"{tag}_thresholds"], g[f"{tag}_doy_map"], g["definitions"],
                           g[f"{tag}_north"], g[f"{tag}_south"], g[f"{tag}_is_south"])
    assert np.array_equal(met, g[f"{tag}_metrics"])


def test_c_oracle_special_values():
    rng = np.random.default_rng(5)
    dates = orc.noleap_date_range("2001-01-01", "2003-08-20")
    x = rng.normal(size=(5, dates.size)).astype(np.float32)
    x[0, 400] = np.nan
    x[1, 10] = np.inf
    x[2, 20] = x[2, 21] = -np.inf
    x[3, :] = 1.5
    x[4, :] = np.round(x[4, :])
    win = orc.datetimes_to_windows(dates, 7)
    q = [0.0, 0.3, 0.9, 0.95, 1.0]
    with np.errstate(invalid="ignore"):
        want = orc.compute_thresholds_cells(x, win, q)
    assert same(c_oracle.thresholds(x, win, q), want)


def test_c_oracle_metrics_random_definitions():
    rng = np.random.default_rng(12)
    dates = orc.noleap_date_range("2001-01-01", "2004-12-31")
    x = rng.normal(size=(4, dates.size)).astype(np.float32)
    thr = np.sort(rng.normal(0.5, 0.5, size=(4, 365, 3)), axis=2)
    defs = [[int(rng.integers(0, 6)), int(rng.integers(0, 4)), int(rng.integers(0, 4))] for _ in range(7)]
    north, south, _ = orc.hemisphere_ranges(dates)
    dm = orc.build_doy_map(dates)
    hemi = np.array([0, 1, 1, 0], dtype=np.uint8)
    assert np.array_equal(c_oracle.metrics(x, thr, dm, defs, north, south, hemi),
                          orc.compute_metrics_cells(x, thr, dm, defs, north, south, hemi))
