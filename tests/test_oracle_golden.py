"""Pin the CPU oracle (oracle/hdp_oracle.py) against the reference's own
known-answer tests and against vectors produced by the reference's code
(tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import hdp_oracle as orc


@pytest.fixture(scope="module")
def kat(golden_dir):
    return json.load(open(os.path.join(golden_dir, "reference_kat.json")))


def test_index_heatwaves_reference_kat(kat):
    assert len(kat["index_heatwaves"]) == 15
    for case in kat["index_heatwaves"]:
        got = orc.index_heatwaves(np.array(case["hot"], dtype=bool), *case["definition"])
        assert np.array_equal(got, case["expected"]), case["case"]


def test_season_metrics_reference_kat(kat):
    assert len(kat["season_metrics"]) == 32
    for case in kat["season_metrics"]:
        fn = getattr(orc, case["function"])
        got = fn(np.array(case["ids"]), np.array(case["ranges"]))
        if case["function"] == "heatwave_average":
            assert np.allclose(got, case["expected"], rtol=1e-15, atol=0), case
        else:
            assert np.array_equal(got, case["expected"]), case


def test_window_docstring_example(kat):
    ex = kat["window_docstring_example"]
    dates = orc.noleap_date_range("2001-01-01", "2001-01-06")
    assert np.array_equal(orc.datetimes_to_windows(dates, ex["radius"]), ex["expected"])


def test_random_series_vs_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "metric_random.npz"))
    off, roff = g["offsets"], g["range_offsets"]
    for i in range(off.size - 1):
        hot = g["hot"][off[i]:off[i + 1]].astype(bool)
        ids = orc.index_heatwaves(hot, *g["definitions"][i])
        assert np.array_equal(ids, g["ids"][off[i]:off[i + 1]]), i
        rng = g["ranges"][roff[i]:roff[i + 1]]
        sl = slice(roff[i], roff[i + 1])
        assert np.array_equal(orc.heatwave_frequency(ids, rng), g["hwf"][sl])
        assert np.array_equal(orc.heatwave_number(ids, rng), g["hwn"][sl])
        assert np.array_equal(orc.heatwave_duration(ids, rng), g["hwd"][sl])
        assert np.array_equal(orc.heatwave_average(ids, rng), g["hwa"][sl])


WINDOW_CASES = {
    "r7_3yr": ("2001-01-01", "2003-12-31", 7),
    "r7_partial_final_year": ("2001-01-01", "2003-06-30", 7),
    "r7_midyear_start": ("2001-03-15", "2004-03-14", 7),
    "r1_6days": ("2001-01-01", "2001-01-06", 1),
    "r2_2yr": ("1999-01-01", "2000-12-31", 2),
    "r0_2yr": ("1999-01-01", "2000-12-31", 0),
}


@pytest.mark.parametrize("name", sorted(WINDOW_CASES))
def test_window_tables_vs_reference(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "window_tables.npz"))
    s, e, r = WINDOW_CASES[name]
    assert np.array_equal(orc.datetimes_to_windows(orc.noleap_date_range(s, e), r), g[name])


def test_season_tables_vs_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "season_tables.npz"))
    for tag in ("50yr", "midyear", "short"):
        s, e = g[f"{tag}_range"]
        dates = orc.noleap_date_range(str(s), str(e))
        assert np.array_equal(orc.get_range_indices(dates, (5, 1), (10, 1)), g[f"{tag}_north"])
        assert np.array_equal(orc.get_range_indices(dates, (11, 1), (4, 1)), g[f"{tag}_south"])
        assert np.array_equal(orc.build_doy_map(dates)[:400], g[f"{tag}_doy_map_head"])


def test_c1_workflow_vs_reference(golden_dir):
    """Config 1 (generator defaults): thresholds within 1e-6 relative of the
    reference's gather + NumPy quantile (Numba arithmetic differs by ~1 ulp),
    metrics bit-exact when fed the reference's thresholds."""
    g = np.load(os.path.join(golden_dir, "c1_workflow.npz"))
    for tag, noise in (("plain", False), ("noise", True)):
        base, lon, lat, bdates = orc.generate_control(add_noise=noise)
        meas, _, _, mdates = orc.generate_warming(add_noise=noise)
        win = orc.datetimes_to_windows(bdates, 7)
        if tag == "plain":
            assert np.array_equal(win[g["window_table_row_ids"]], g["window_table_rows"])
            assert np.array_equal(win.sum(axis=1), g["window_table_sum_per_row"])
            assert np.array_equal(orc.build_doy_map(mdates), g["doy_map"])
        x = base.astype(np.float32).reshape(-1, base.shape[-1])
        thr = orc.compute_thresholds_cells(x, win, g["percentiles"]).reshape(g[f"{tag}_thresholds"].shape)
        np.testing.assert_allclose(thr, g[f"{tag}_thresholds"], rtol=1e-6, atol=0)
        assert np.max(np.abs(thr - g[f"{tag}_thresholds"])) < 1e-12
        north, south, years = orc.hemisphere_ranges(mdates)
        assert np.array_equal(north, g["north"]) and np.array_equal(south, g["south"])
        m = meas.astype(np.float32).reshape(-1, meas.shape[-1])
        is_south = np.repeat((lat < 0)[None, :], base.shape[0], axis=0).reshape(-1)
        got = orc.compute_metrics_cells(m, g[f"{tag}_thresholds"].reshape(m.shape[0], 365, -1),
                                        g["doy_map"], g["definitions"], north, south, is_south)
        assert np.array_equal(got.reshape(g[f"{tag}_metrics"].shape), g[f"{tag}_metrics"])


@pytest.mark.parametrize("tag", ["full3", "ragged"])
def test_small_workflow_vs_reference(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "small_workflow.npz"))
    s, e = g[f"{tag}_range"]
    dates = orc.noleap_date_range(str(s), str(e))
    win = orc.datetimes_to_windows(dates, 7)
    assert np.array_equal(win, g[f"{tag}_window"])
    thr = orc.compute_thresholds_cells(g[f"{tag}_baseline"], win, g["percentiles"])
    np.testing.assert_allclose(thr, g[f"{tag}_thresholds"], rtol=1e-6, atol=0)
    got = orc.compute_metrics_cells(g[f"{tag}_measure"], g[f"{tag}_thresholds"], g[f"{tag}_doy_map"],
                                    g["definitions"], g[f"{tag}_north"], g[f"{tag}_south"],
                                    g[f"{tag}_is_south"])
    assert np.array_equal(got, g[f"{tag}_metrics"])


def test_quantile_matches_numpy_linear_closely():
    rng = np.random.default_rng(3)
    for n in (2, 3, 10, 150, 751, 1500):
        a = rng.normal(size=n).astype(np.float32)
        q = np.concatenate([np.arange(0.9, 1, 0.01), [0.0, 1.0, 0.5, 0.123]])
        got = orc.numba_quantile(a, q)
        want = np.quantile(a.astype(np.float64), q)
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-13)


def test_quantile_edge_cases():
    q = np.array([0.0, 0.5, 1.0])
    assert np.all(np.isnan(orc.numba_quantile([1.0, np.nan, 2.0], q)))
    assert np.array_equal(orc.numba_quantile([3.0], q), [3.0, 3.0, 3.0])
    assert np.all(np.isnan(orc.numba_quantile([np.inf], q)))
    assert np.array_equal(orc.numba_quantile([2.0, 2.0, 2.0, 2.0], q), [2.0, 2.0, 2.0])
    # +inf as the maximum is reported as NaN at q=1 (numba's heuristic)
    r = orc.numba_quantile([1.0, 2.0, np.inf], q)
    assert r[0] == 1.0 and np.isnan(r[2])
    with pytest.raises(ValueError):
        orc.numba_quantile([1.0, 2.0], [1.5])


def test_heat_index_vs_reference_stub_run(golden_dir):
    """SURVEY 8f row 1.  The fixture is the reference ufunc body run as a plain Python function on
    NumPy float32 scalars; the oracle follows Numba's float64 typing of the same expression, so
    the two agree to ~1e-5 relative, not bit for bit ("parity unpinned" at the ulp level)."""
    g = np.load(os.path.join(golden_dir, "heat_index.npz"))
    got = orc.heat_index(g["temp_f"], g["rel_humid"]).astype(np.float64)
    ref = g["reference_stub_run"]
    assert got.shape == ref.shape and got.size > 6000
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-4)
    # every branch of the regression is exercised
    t, r = g["temp_f"], g["rel_humid"]
    assert ((ref <= 80).sum() > 100 and ((r < 13) & (t >= 80) & (t <= 112)).sum() > 50
            and ((r > 85) & (t >= 80) & (t <= 87)).sum() > 20)


def test_weighted_spatial_mean_restatement():
    """oracle.weighted_spatial_mean (figure.py:14-15 via xarray's weighted mean): a constant field averages to itself
    whatever the weights, NaNs drop out of both sums, equal weights reduce to the plain mean, and the poles'
    ~0 weights make them irrelevant."""
    lat = np.linspace(-90, 90, 7)
    v = np.full((2, 7, 4), 3.25)
    assert np.allclose(orc.weighted_spatial_mean(v, lat, 4), 3.25, rtol=1e-15)
    rng = np.random.default_rng(0)
    x = rng.normal(size=(7, 4))
    w = np.cos(np.deg2rad(lat))[:, None] * np.ones((1, 4))
    assert np.isclose(orc.weighted_spatial_mean(x, lat, 4), (x * w).sum() / w.sum(), rtol=1e-14)
    y = x.copy(); y[2, 1] = np.nan
    keep = ~np.isnan(y)
    assert np.isclose(orc.weighted_spatial_mean(y, lat, 4), (x * w)[keep].sum() / w[keep].sum(), rtol=1e-14)
    assert np.isclose(orc.weighted_spatial_mean(x, np.zeros(7), 4), x.mean(), rtol=1e-14)
    z = x.copy(); z[0] += 1e6; z[-1] -= 1e6           # rows at the poles carry cos(90 deg) ~ 6e-17
    assert np.isclose(orc.weighted_spatial_mean(z, lat, 4), orc.weighted_spatial_mean(x, lat, 4), atol=1e-8)
    assert np.isnan(orc.weighted_spatial_mean(np.full((7, 4), np.nan), lat, 4))
