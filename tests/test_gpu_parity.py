"""Parity of the HIP path (through the C ABI) against the CPU oracle and the
reference-generated golden fixtures.  Integers bit-exact; thresholds bit-exact against
the oracle (same float64 operation order) and <= 1e-6 relative (north_star tolerance)
against the fixtures that went through NumPy's quantile."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from hdp_amd import calendar as cal  # noqa: E402
from hdp_amd import core  # noqa: E402
from oracle import hdp_oracle as orc  # noqa: E402

REL_TOL = 1e-6  # north_star: "within 1e-6 relative for float thresholds"


def same_f64(a, b):
    """bit-level equality up to NaN payload and the sign of zero"""
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


# ---- known-answer tests of the reference, through the C ABI --------------------------------

@pytest.fixture(scope="module")
def kat(golden_dir):
    return json.load(open(os.path.join(golden_dir, "reference_kat.json")))


def test_index_heatwaves_reference_kat(kat):
    for case in kat["index_heatwaves"]:
        got = core.index_heatwaves(np.array(case["hot"], dtype=bool), *case["definition"])
        assert np.array_equal(got, case["expected"]), case["case"]


def test_season_metrics_reference_kat(kat):
    for case in kat["season_metrics"]:
        fn = getattr(core, case["function"])
        got = fn(np.array(case["ids"]), np.array(case["ranges"]))
        if case["function"] == "heatwave_average":
            assert np.allclose(got, case["expected"], rtol=1e-15, atol=0), case
        else:
            assert np.array_equal(got, case["expected"]), case


def test_random_series_vs_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "metric_random.npz"))
    off, roff = g["offsets"], g["range_offsets"]
    for i in range(off.size - 1):
        hot = g["hot"][off[i]:off[i + 1]].astype(bool)
        ids = core.index_heatwaves(hot, *g["definitions"][i])
        assert np.array_equal(ids, g["ids"][off[i]:off[i + 1]]), i
        rng = g["ranges"][roff[i]:roff[i + 1]]
        sl = slice(roff[i], roff[i + 1])
        assert np.array_equal(core.heatwave_frequency(ids, rng), g["hwf"][sl])
        assert np.array_equal(core.heatwave_number(ids, rng), g["hwn"][sl])
        assert np.array_equal(core.heatwave_duration(ids, rng), g["hwd"][sl])
        assert np.array_equal(core.heatwave_average(ids, rng), g["hwa"][sl])


def test_empty_season_raises_like_reference():
    with pytest.raises(ValueError):
        core.heatwave_duration(np.zeros(10, dtype=np.int64), np.array([[3, 3]]))


# ---- thresholds ---------------------------------------------------------------------------------------

@pytest.mark.parametrize("tag", ["full3", "ragged"])
def test_thresholds_small_workflow(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "small_workflow.npz"))
    s, e = g[f"{tag}_range"]
    dates = orc.noleap_date_range(str(s), str(e))
    ti, cols = cal.window_columns(dates, 7)
    x = g[f"{tag}_baseline"]
    got = core.compute_percentiles(x, ti, cols, g["percentiles"])
    want = orc.compute_thresholds_cells(x, g[f"{tag}_window"], g["percentiles"])
    assert same_f64(got, want)                                   # vs oracle: bit-exact
    np.testing.assert_allclose(got, g[f"{tag}_thresholds"], rtol=REL_TOL, atol=0)  # vs reference run
    tab = core.compute_percentiles_table(x, g[f"{tag}_window"], g["percentiles"])
    assert same_f64(tab, want)


def test_thresholds_c1_generator_defaults(golden_dir):
    g = np.load(os.path.join(golden_dir, "c1_workflow.npz"))
    for tag, noise in (("plain", False), ("noise", True)):
        base, lon, lat, dates = orc.generate_control(add_noise=noise)
        x = base.astype(np.float32).reshape(-1, base.shape[-1])
        ti, cols = cal.window_columns(dates, 7)
        got = core.compute_percentiles(x, ti, cols, g["percentiles"])
        want = orc.compute_thresholds_cells(x, cal.expand_window_table(ti, cols), g["percentiles"])
        assert same_f64(got, want)
        np.testing.assert_allclose(got.reshape(g[f"{tag}_thresholds"].shape), g[f"{tag}_thresholds"],
                                   rtol=REL_TOL, atol=0)


@pytest.mark.parametrize("q", [
    [0.0, 1.0], [0.5], [0.0, 0.1, 0.25, 0.5, 0.75, 0.9, 1.0], list(np.arange(0.9, 1, 0.01)),
    [0.999, 0.001], [1 / 3], list(np.linspace(0.80, 0.99, 20)),
])
def test_thresholds_arbitrary_quantiles(q):
    rng = np.random.default_rng(11)
    dates = orc.noleap_date_range("2001-01-01", "2006-12-31")
    x = rng.normal(10, 5, size=(7, dates.size)).astype(np.float32)
    ti, cols = cal.window_columns(dates, 7)
    want = orc.compute_thresholds_cells(x, cal.expand_window_table(ti, cols), q)
    assert same_f64(core.compute_percentiles(x, ti, cols, q), want)
    assert same_f64(core.compute_percentiles_table(x, cal.expand_window_table(ti, cols), q), want)


@pytest.mark.parametrize("radius", [0, 1, 3, 10])
def test_thresholds_other_window_sizes(radius):
    rng = np.random.default_rng(radius)
    dates = orc.noleap_date_range("2001-01-01", "2004-12-31")
    x = rng.normal(size=(3, dates.size)).astype(np.float32)
    ti, cols = cal.window_columns(dates, radius)
    q = [0.05, 0.5, 0.9, 0.99]
    want = orc.compute_thresholds_cells(x, cal.expand_window_table(ti, cols), q)
    assert same_f64(core.compute_percentiles(x, ti, cols, q), want)


def test_thresholds_special_values():
    """NaN anywhere in a window -> all quantiles NaN; infinities follow numba's heuristics;
    duplicated samples; -1 padding samples the last time step."""
    rng = np.random.default_rng(5)
    dates = orc.noleap_date_range("2001-01-01", "2003-08-20")   # ragged: -1 padding
    T = dates.size
    x = rng.normal(size=(6, T)).astype(np.float32)
    x[0, 400] = np.nan
    x[1, 10] = np.inf
    x[2, 20] = -np.inf
    x[2, 21] = -np.inf
    x[3, :] = 1.5                      # all duplicates
    x[4, :] = np.round(x[4, :])        # many ties
    x[5, T - 1] = np.inf               # the padded sample
    ti, cols = cal.window_columns(dates, 7)
    q = [0.0, 0.3, 0.9, 0.95, 1.0]
    with np.errstate(invalid="ignore"):
        want = orc.compute_thresholds_cells(x, cal.expand_window_table(ti, cols), q)
    got = core.compute_percentiles(x, ti, cols, q)
    assert same_f64(got, want)
    assert np.isnan(got[0]).any() and not np.isnan(got[0]).all()
    assert same_f64(core.compute_percentiles_table(x, cal.expand_window_table(ti, cols), q), want)


def test_thresholds_bad_quantile_raises_like_numba():
    dates = orc.noleap_date_range("2001-01-01", "2002-12-31")
    x = np.zeros((1, dates.size), dtype=np.float32)
    ti, cols = cal.window_columns(dates, 7)
    with pytest.raises(ValueError, match="Quantiles must be in the range"):
        core.compute_percentiles(x, ti, cols, [0.5, 1.5])


def test_thresholds_strided_layouts():
    """time-major [T, cells] input (CMIP layout) gives the same result as time-contiguous."""
    rng = np.random.default_rng(8)
    dates = orc.noleap_date_range("2001-01-01", "2004-12-31")
    xt = rng.normal(size=(dates.size, 9)).astype(np.float32)     # [T, cells]
    ti, cols = cal.window_columns(dates, 7)
    q = [0.9, 0.95]
    a = core.compute_percentiles(xt.T, ti, cols, q)
    b = core.compute_percentiles(np.ascontiguousarray(xt.T), ti, cols, q)
    assert same_f64(a, b)


def test_thresholds_many_samples_per_doy():
    """S = 120 (two registers per lane) and S = 300 (ensemble-like: members concatenated)."""
    rng = np.random.default_rng(21)
    for years in (120, 300):
        dates = orc.noleap_date_range("0001-01-01", f"{years:04d}-12-31")
        x = rng.normal(size=(2, dates.size)).astype(np.float32)
        ti, cols = cal.window_columns(dates, 7)
        q = [0.9, 0.99]
        got = core.compute_percentiles(x, ti, cols, q)
        # oracle on a subset of rows (full table is slow in pure Python)
        rows = [0, 1, 100, 357, 358, 364]
        win = cal.expand_window_table(ti, cols)[rows]
        want = orc.compute_thresholds_cells(x, win, q)
        assert same_f64(got[:, rows], want)


# ---- metrics ------------------------------------------------------------------------------------------

@pytest.mark.parametrize("tag", ["full3", "ragged"])
def test_metrics_small_workflow(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "small_workflow.npz"))
    got = core.compute_heatwave_metrics(g[f"{tag}_measure"], g[f"{tag}_thresholds"], g[f"{tag}_doy_map"],
                                        g["definitions"], g[f"{tag}_north"], g[f"{tag}_south"],
                                        g[f"{tag}_is_south"])
    assert got.dtype == np.int16
    assert np.array_equal(got.astype(np.int64), g[f"{tag}_metrics"])


def test_metrics_c1_generator_defaults(golden_dir):
    g = np.load(os.path.join(golden_dir, "c1_workflow.npz"))
    for tag, noise in (("plain", False), ("noise", True)):
        meas, lon, lat, dates = orc.generate_warming(add_noise=noise)
        m = meas.astype(np.float32).reshape(-1, meas.shape[-1])
        is_south = np.repeat((lat < 0)[None, :], meas.shape[0], axis=0).reshape(-1)
        thr = g[f"{tag}_thresholds"].reshape(m.shape[0], 365, -1)
        got = core.compute_heatwave_metrics(m, thr, g["doy_map"], g["definitions"], g["north"], g["south"],
                                            is_south)
        assert np.array_equal(got.astype(np.int64).reshape(g[f"{tag}_metrics"].shape), g[f"{tag}_metrics"])


def test_metrics_many_definitions_and_percentiles():
    """P*D = 240 > 64: several lane groups per series; shared thresholds across members."""
    rng = np.random.default_rng(2)
    dates = orc.noleap_date_range("2001-01-01", "2006-12-31")
    T = dates.size
    n_cells, members = 3, 2
    x = rng.normal(0, 1, size=(members * n_cells, T)).astype(np.float32)
    P = 20
    thr = np.sort(rng.normal(0.8, 0.5, size=(n_cells, 365, P)), axis=2)
    defs = [[a, b, b] for a in (3, 4, 5, 6) for b in (0, 1, 2)]
    doy_map = cal.build_doy_map(dates)
    north, south, _ = cal.hemisphere_season_tables(dates)
    is_south = np.array([0, 1, 0] * members, dtype=np.uint8)
    got = core.compute_heatwave_metrics(x, thr, doy_map, defs, north, south, is_south)
    want = orc.compute_metrics_cells(x, np.concatenate([thr] * members), doy_map, defs, north, south, is_south)
    assert np.array_equal(got.astype(np.int64), want)


def test_metrics_edge_series():
    """never hot / always hot (one run spanning every season) / NaN measure / NaN threshold /
    exact ties (strict >)."""
    dates = orc.noleap_date_range("2001-01-01", "2005-12-31")
    T = dates.size
    rng = np.random.default_rng(9)
    x = rng.normal(size=(6, T)).astype(np.float32)
    thr = np.zeros((6, 365, 2))
    thr[..., 1] = 1.0
    x[0] = -5          # never hot
    x[1] = 5           # always hot
    x[2, ::7] = np.nan  # NaN days are not hot
    thr[3, 100:200, :] = np.nan
    x[4] = 1.0         # equals threshold p=1 -> not hot (strict), hot for p=0
    defs = [[3, 0, 0], [1, 1, 1], [0, 0, 1], [2, 3, 2], [5, 1, 4]]
    doy_map = cal.build_doy_map(dates)
    north, south, _ = cal.hemisphere_season_tables(dates)
    is_south = np.array([0, 1, 0, 1, 0, 1], dtype=np.uint8)
    got = core.compute_heatwave_metrics(x, thr, doy_map, defs, north, south, is_south)
    want = orc.compute_metrics_cells(x, thr, doy_map, defs, north, south, is_south)
    assert np.array_equal(got.astype(np.int64), want)
    assert got[0, :, 0].max() == 0 and got[0, 0, 1, 0].min() >= 61


def test_indicate_hot_days_matches_oracle():
    rng = np.random.default_rng(4)
    dates = orc.noleap_date_range("2001-01-01", "2002-12-31")
    x = rng.normal(size=dates.size).astype(np.float32)
    thr = rng.normal(size=365)
    thr[5] = float(x[5])  # exact tie
    dm = cal.build_doy_map(dates)
    assert np.array_equal(core.indicate_hot_days(x, thr, dm), orc.indicate_hot_days(x, thr, dm))


def test_end_to_end_c2_shape_sampled():
    """3650 d x 24 cells, 10 percentiles x 6 definitions: HIP thresholds -> HIP metrics
    against oracle thresholds -> oracle metrics on the same seeded inputs."""
    rng = np.random.default_rng(1)
    dates = orc.noleap_date_range("2001-01-01", "2010-12-31")
    T = dates.size
    n = 24
    t = np.arange(T)
    base = (15 + 8 * np.sin(2 * np.pi * (t[None, :] - 110) / 365) + rng.normal(0, 2.5, size=(n, T))).astype(np.float32)
    meas = (base + 0.5 + t[None, :] / 36500.0 + rng.normal(0, 1.0, size=(n, T))).astype(np.float32)
    q = np.arange(0.9, 1, 0.01)
    defs = [[3, 0, 0], [3, 1, 1], [4, 0, 0], [4, 1, 1], [5, 0, 0], [5, 1, 1]]
    ti, cols = cal.window_columns(dates, 7)
    thr = core.compute_percentiles(base, ti, cols, q)
    sample = [0, 5, 23]
    want_thr = orc.compute_thresholds_cells(base[sample], cal.expand_window_table(ti, cols), q)
    assert same_f64(thr[sample], want_thr)
    doy_map = cal.build_doy_map(dates)
    north, south, _ = cal.hemisphere_season_tables(dates)
    is_south = (np.arange(n) % 2).astype(np.uint8)
    got = core.compute_heatwave_metrics(meas, thr, doy_map, defs, north, south, is_south)
    want = orc.compute_metrics_cells(meas[sample], thr[sample], doy_map, defs, north, south, is_south[sample])
    assert np.array_equal(got[:, :, sample].astype(np.int64), want)


def _random_metrics_case(seed, years, n, P, defs, trend=0.0):
    rng = np.random.default_rng(seed)
    dates = orc.noleap_date_range("2001-01-01", f"{2000 + years}-12-31")
    T = dates.size
    x = (rng.normal(0, 1, size=(n, T)) + trend * np.arange(T)[None, :] / T).astype(np.float32)
    thr = np.sort(rng.normal(0.6, 0.4, size=(n, 365, P)), axis=2)
    doy_map = cal.build_doy_map(dates)
    north, south, _ = cal.hemisphere_season_tables(dates)
    is_south = (np.arange(n) % 2).astype(np.uint8)
    return x, thr, doy_map, defs, north, south, is_south


@pytest.mark.parametrize("years", [16, 17, 32, 3])
def test_metrics_season_count_vs_packed_stores(years):
    """Y = 16, 32 (exact 16-season store groups), 17 and 3 (partial groups)."""
    case = _random_metrics_case(years, years, 5, 3, [[3, 0, 0], [2, 1, 1], [6, 2, 0], [1, 0, 3]], trend=2.0)
    got = core.compute_heatwave_metrics(*case)
    assert np.array_equal(got.astype(np.int64), orc.compute_metrics_cells(*case))


def test_metrics_general_kernel_forced(monkeypatch):
    """The per-lane lazy-season kernel (used when seasons are too close for the uniform one)."""
    case = _random_metrics_case(77, 6, 6, 4, [[3, 0, 0], [3, 1, 1], [5, 1, 4], [0, 0, 1]], trend=1.5)
    want = orc.compute_metrics_cells(*case)
    monkeypatch.setenv("HDP_METRICS_GENERAL", "1")
    got_general = core.compute_heatwave_metrics(*case)
    monkeypatch.delenv("HDP_METRICS_GENERAL")
    got_uniform = core.compute_heatwave_metrics(*case)
    assert np.array_equal(got_general.astype(np.int64), want)
    assert np.array_equal(got_uniform.astype(np.int64), want)


def test_metrics_adjacent_and_short_gap_seasons():
    """User-supplied season tables with touching / nearly touching ranges (general kernel) and
    with gaps just above the uniform-kernel threshold."""
    x, thr, doy_map, defs, _, _, is_south = _random_metrics_case(5, 4, 6, 3, [[3, 1, 1], [2, 0, 0], [7, 2, 1]], 2.0)
    T = x.shape[1]
    for north, south in (
        (np.array([[0, 100], [100, 200], [230, 400], [401, T]]), np.array([[10, 50], [50, 51], [60, 700], [700, T]])),
        (np.array([[5, 100], [171, 300], [371, 600], [700, T]]), np.array([[0, 30], [101, 131], [202, 232], [303, 333]])),
    ):
        got = core.compute_heatwave_metrics(x, thr, doy_map, defs, north, south, is_south)
        want = orc.compute_metrics_cells(x, thr, doy_map, defs, north, south, is_south)
        assert np.array_equal(got.astype(np.int64), want)


def test_metrics_long_min_duration_and_record_end():
    """min_duration far above typical runs, T a multiple of 64 with a run open at the end."""
    dates = orc.noleap_date_range("2001-01-01", "2007-12-31")[:2560]
    T = dates.size
    rng = np.random.default_rng(31)
    x = rng.normal(size=(4, T)).astype(np.float32)
    x[:, -40:] = 9.0
    x[1, 100:400] = 9.0
    thr = np.zeros((4, 365, 2)); thr[..., 1] = 0.8
    defs = [[40, 2, 1], [3, 0, 0], [30, 0, 0]]
    doy_map = cal.build_doy_map(dates)
    north = np.array([[120, 273], [485, 638], [850, 1003], [1215, 1368], [1580, 1733], [1945, 2098], [2310, 2463]])
    south = np.array([[304, 455], [669, 820], [1034, 1185], [1399, 1550], [1764, 1915], [2129, 2280], [2494, T]])
    is_south = np.array([0, 1, 0, 1], dtype=np.uint8)
    got = core.compute_heatwave_metrics(x, thr, doy_map, defs, north, south, is_south)
    want = orc.compute_metrics_cells(x, thr, doy_map, defs, north, south, is_south)
    assert np.array_equal(got.astype(np.int64), want)


@pytest.mark.parametrize("n_defs", [1, 5, 6, 7, 12])
def test_metrics_series_per_lane_kernel_definition_passes(n_defs, monkeypatch):
    """metrics_kernel_cells<DG>: every template width (1..6 definitions per lane) and more than six
    definitions (two passes over the bit words); hemispheres alternate inside every wave (two passes per
    wave), 131 series (two full waves + a ragged one); against the oracle and the pair-per-lane kernel."""
    from oracle import c_oracle
    rng = np.random.default_rng(1000 + n_defs)
    defs = [[int(rng.integers(0, 7)), int(rng.integers(0, 3)), int(rng.integers(0, 3))] for _ in range(n_defs)]
    case = _random_metrics_case(500 + n_defs, 7, 131, 3, defs, trend=1.2)
    want = c_oracle.metrics(*case)
    got = core.compute_heatwave_metrics(*case)
    assert np.array_equal(got.astype(np.int64), want)
    monkeypatch.setenv("HDP_METRICS_BATCH", "50")   # three batches: both plan streams + the double buffer
    assert np.array_equal(core.compute_heatwave_metrics(*case), got)
    monkeypatch.setenv("HDP_METRICS_OVERLAP", "0")
    assert np.array_equal(core.compute_heatwave_metrics(*case), got)


def test_metrics_packed_and_unpacked_state_machines(monkeypatch):
    """The series-per-lane kernel keeps two definitions per register as 16-bit halves when every min_duration and
    max_break is <= 16383 (and T <= 65535); otherwise 32-bit state.  Same results, including at the limit, with an
    odd number of definitions (a padding half), max_subs beyond 16 bits and a long always-hot stretch."""
    from oracle import c_oracle
    rng = np.random.default_rng(4242)
    base_defs = [[3, 0, 0], [3, 1, 1], [1, 2, 100000], [0, 0, 1], [7, 3, 2]]
    case = list(_random_metrics_case(321, 9, 97, 4, base_defs, trend=1.5))
    case[0][5, 400:1900] = 50.0                      # one series: a 1500-day run
    want = c_oracle.metrics(*case)
    got = core.compute_heatwave_metrics(*case)
    assert np.array_equal(got.astype(np.int64), want)
    monkeypatch.setenv("HDP_METRICS_PACKED", "0")
    assert np.array_equal(core.compute_heatwave_metrics(*case), got)
    monkeypatch.delenv("HDP_METRICS_PACKED")
    for edge in ([[16383, 0, 0], [3, 16383, 1]], [[16384, 0, 0], [3, 1, 1]], [[3, 16384, 2]]):
        case[3] = edge
        assert np.array_equal(core.compute_heatwave_metrics(*case).astype(np.int64), c_oracle.metrics(*case))


def test_metrics_split_path_in_small_batches(monkeypatch):
    """Exceedance-scratch path forced into several batches (ragged last one), and the fused kernel."""
    case = _random_metrics_case(123, 5, 11, 4, [[3, 0, 0], [3, 1, 1], [4, 2, 2]], trend=1.0)
    want = orc.compute_metrics_cells(*case)
    monkeypatch.setenv("HDP_METRICS_BATCH", "4")
    assert np.array_equal(core.compute_heatwave_metrics(*case).astype(np.int64), want)
    monkeypatch.delenv("HDP_METRICS_BATCH")
    monkeypatch.setenv("HDP_METRICS_GENERAL", "1")   # seasons closed per lane: the one cross-check kernel in every build
    assert np.array_equal(core.compute_heatwave_metrics(*case).astype(np.int64), want)


def test_c2_full_size_cross_kernel_properties(monkeypatch):
    """BASELINE config 2 at FULL size (3650 d x 180 x 360 = 64800 cells, 10 percentiles x 6
    definitions), too big for the Python oracle: size-independent properties instead --
      * the four independently written metrics kernels (series per lane, (percentile, definition) per lane,
        fused, general) agree bit for bit;
      * the plan-based threshold kernel agrees bit for bit with the literal-table kernel on a sample;
      * thresholds are non-decreasing in the percentile; HWF >= HWD >= HWA >= 0; HWN <= HWF;
        HWA == HWF // HWN;  all-zero where HWN == 0;
      * a sample of cells matches the C oracle exactly."""
    from oracle import c_oracle
    rng = np.random.default_rng(2024)
    dates = orc.noleap_date_range("2001-01-01", "2010-12-31")
    T, n_lat, n_lon = dates.size, 180, 360
    n = n_lat * n_lon
    lat = np.repeat(np.linspace(-90, 90, n_lat), n_lon)
    t = np.arange(T, dtype=np.float32)
    season = (20 + 8 * np.sin(2 * np.pi * (t - 110) / 365)).astype(np.float32)
    base = season[None, :] - (10 * np.abs(lat) / 90).astype(np.float32)[:, None] \
        + rng.normal(0, 2.0, size=(n, T)).astype(np.float32)
    meas = base + np.float32(0.6) + (t / np.float32(36500.0))[None, :] \
        + rng.normal(0, 1.0, size=(n, T)).astype(np.float32)
    q = np.arange(0.9, 1, 0.01)
    defs = [[3, 0, 0], [3, 1, 1], [4, 0, 0], [4, 1, 1], [5, 0, 0], [5, 1, 1]]
    ti, cols = cal.window_columns(dates, 7)
    thr = core.compute_percentiles(base, ti, cols, q)
    assert thr.shape == (n, 365, 10) and not np.isnan(thr).any()
    assert np.all(np.diff(thr, axis=2) >= 0)
    sample = rng.choice(n, size=48, replace=False)
    win = cal.expand_window_table(ti, cols)
    assert same_f64(thr[sample], core.compute_percentiles_table(base[sample], win, q))
    assert same_f64(thr[sample], c_oracle.thresholds(base[sample], win, q))

    doy_map = cal.build_doy_map(dates)
    north, south, _ = cal.hemisphere_season_tables(dates)
    is_south = (lat < 0).astype(np.uint8)
    split = core.compute_heatwave_metrics(meas, thr, doy_map, defs, north, south, is_south)
    monkeypatch.setenv("HDP_METRICS_GENERAL", "1")
    general = core.compute_heatwave_metrics(meas, thr, doy_map, defs, north, south, is_south)
    monkeypatch.delenv("HDP_METRICS_GENERAL")
    # (the round-1 fused and (percentile, definition)-per-lane kernels are built only with -DHDP_CROSSCHECK_KERNELS; in the
    # shipped library their switches select nothing, so they are not compared here)
    assert np.array_equal(split, general)
    hwf, hwn, hwd, hwa = (split[:, :, :, i, :].astype(np.int64) for i in range(4))
    assert hwf.min() >= 0 and np.all(hwf >= hwd) and np.all(hwd >= hwa) and np.all(hwn <= hwf)
    assert np.array_equal(hwa, np.where(hwn > 0, hwf // np.maximum(hwn, 1), 0))
    assert np.all(hwf[hwn == 0] == 0) and hwf.max() <= 153 and hwf.sum() > 0
    want = c_oracle.metrics(meas[sample], thr[sample], doy_map, defs, north, south, is_south[sample])
    assert np.array_equal(split[:, :, sample].astype(np.int64), want)


def test_c3_shape_slice_cross_kernel_properties(monkeypatch):
    """BASELINE config 3's series shape (36500 d = 100 noleap years, S = 100 samples per day of year, 151-step
    merges, 100 seasons) on a 2048-cell slice spanning both hemispheres -- the kernel variants the headline
    bench runs (lane-per-column thresholds kernel, packed series-per-lane state machine):
      * lane-per-column / one-workgroup-per-cell merge / rank-selection thresholds kernels agree bit for bit;
      * packed 16-bit, 32-bit and (percentile, definition)-per-lane metrics kernels agree bit for bit;
      * monotone thresholds, HWF >= HWD >= HWA >= 0, HWN <= HWF, HWA == HWF // HWN;
      * a sample of cells matches the C oracle exactly."""
    from oracle import c_oracle
    rng = np.random.default_rng(33)
    dates = orc.noleap_date_range("1901-01-01", "2000-12-31")
    T, n = dates.size, 2048
    lat = np.linspace(-80, 80, n)
    t = np.arange(T, dtype=np.float32)
    season = (20 + 2 * np.sin(2 * np.pi * (t + 90) / 365)).astype(np.float32)
    base = season[None, :] - (10 * np.abs(lat) / 90).astype(np.float32)[:, None] \
        + rng.random(size=(n, T), dtype=np.float32) * np.float32(0.35)
    meas = base + (t / np.float32(36500.0))[None, :]
    meas[:, ::3] += np.float32(0.05)
    q = np.arange(0.9, 1, 0.01)
    defs = [[3, 0, 0], [3, 1, 1], [4, 0, 0], [4, 1, 1], [5, 0, 0], [5, 1, 1]]
    ti, cols = cal.window_columns(dates, 7)
    assert ti.shape == (365, 100)
    thr = core.compute_percentiles(base, ti, cols, q)
    assert thr.shape == (n, 365, 10) and not np.isnan(thr).any() and np.all(np.diff(thr, axis=2) >= 0)
    monkeypatch.setenv("HDP_THR_LANE", "0")        # the one-workgroup-per-cell kernel instead of the lane-per-column one
    assert same_f64(thr, core.compute_percentiles(base, ti, cols, q))
    monkeypatch.delenv("HDP_THR_LANE")
    monkeypatch.setenv("HDP_THR_PIPE", "0")
    assert same_f64(thr, core.compute_percentiles(base, ti, cols, q))
    monkeypatch.setenv("HDP_THR_SELECT", "1")
    assert same_f64(thr[:256], core.compute_percentiles(base[:256], ti, cols, q))
    monkeypatch.delenv("HDP_THR_SELECT")
    monkeypatch.delenv("HDP_THR_PIPE")
    sample = rng.choice(n, size=16, replace=False)
    win = cal.expand_window_table(ti, cols)
    assert same_f64(thr[sample], c_oracle.thresholds(base[sample], win, q))

    doy_map = cal.build_doy_map(dates)
    north, south, _ = cal.hemisphere_season_tables(dates)
    is_south = (lat < 0).astype(np.uint8)
    packed = core.compute_heatwave_metrics(meas, thr, doy_map, defs, north, south, is_south)
    monkeypatch.setenv("HDP_METRICS_PACKED", "0")
    unpacked = core.compute_heatwave_metrics(meas, thr, doy_map, defs, north, south, is_south)
    monkeypatch.delenv("HDP_METRICS_PACKED")
    assert np.array_equal(packed, unpacked)
    hwf, hwn, hwd, hwa = (packed[:, :, :, i, :].astype(np.int64) for i in range(4))
    assert packed.shape[-1] == 100 and hwf.min() >= 0 and np.all(hwf >= hwd) and np.all(hwd >= hwa) and np.all(hwn <= hwf)
    assert np.array_equal(hwa, np.where(hwn > 0, hwf // np.maximum(hwn, 1), 0))
    assert np.all(hwf[hwn == 0] == 0) and hwf.max() <= 153 and hwf.sum() > 0
    want = c_oracle.metrics(meas[sample], thr[sample], doy_map, defs, north, south, is_south[sample])
    assert np.array_equal(packed[:, :, sample].astype(np.int64), want)


def test_time_major_inputs_through_device_transpose():
    """CMIP order [time, cells] (stride_cell == 1): strided 2-D upload + device transpose must give
    exactly what the time-contiguous copy gives, for both passes; odd sizes exercise tile edges."""
    rng = np.random.default_rng(17)
    dates = orc.noleap_date_range("2001-01-01", "2004-12-31")
    T, n = dates.size, 131
    xt = rng.normal(1.0, 2.0, size=(T, n)).astype(np.float32)
    ti, cols = cal.window_columns(dates, 7)
    q = [0.9, 0.97]
    thr_a = core.compute_percentiles(xt.T, ti, cols, q)
    thr_b = core.compute_percentiles(np.ascontiguousarray(xt.T), ti, cols, q)
    assert same_f64(thr_a, thr_b)
    dm = cal.build_doy_map(dates)
    north, south, _ = cal.hemisphere_season_tables(dates)
    hemi = (np.arange(n) % 3 == 0).astype(np.uint8)
    defs = [[3, 0, 0], [2, 1, 1]]
    met_a = core.compute_heatwave_metrics(xt.T, thr_a, dm, defs, north, south, hemi)
    met_b = core.compute_heatwave_metrics(np.ascontiguousarray(xt.T), thr_a, dm, defs, north, south, hemi)
    assert np.array_equal(met_a, met_b)
    # a column slice of a wider time-major array (stride_time > n)
    wide = rng.normal(size=(T, n + 40)).astype(np.float32)
    sub = wide[:, 7:7 + n]
    assert same_f64(core.compute_percentiles(sub.T, ti, cols, q),
                    core.compute_percentiles(np.ascontiguousarray(sub.T), ti, cols, q))


def test_c5_shaped_ensemble_thresholds_and_metrics():
    """Config 5 shape at small grid: 10 members x 100 years concatenated along time for thresholds
    (S = 1000 samples per day of year, window of 15000 samples, generic wide-column sort path),
    20 percentiles x 12 definitions = 240 lane pairs, members sharing their cell's thresholds."""
    from oracle import c_oracle
    rng = np.random.default_rng(55)
    dates = orc.noleap_date_range("2001-01-01", "2100-12-31")
    T, members, n_cells = dates.size, 10, 3
    x = rng.normal(12, 3, size=(members, n_cells, T)).astype(np.float32)
    x += (np.arange(T, dtype=np.float32) / np.float32(36500.0))[None, None, :]
    cat = np.concatenate([x[m] for m in range(members)], axis=1)          # [cells, members*T]
    cdates = np.concatenate([dates] * members)
    ti, cols = cal.window_columns(cdates, 7)
    assert ti.shape == (365, 1000)
    q = np.linspace(0.80, 0.99, 20)
    thr = core.compute_percentiles(cat, ti, cols, q)
    # the windows of days 359 .. 364 list one column twice (the reference wraps past the year's end modulo 366): the
    # selection's popped finish has to keep the two copies apart
    rows = [0, 3, 182, 357, 358, 359, 361, 363, 364]
    win = cal.expand_window_table(ti, cols)[rows]
    assert same_f64(thr[:, rows], c_oracle.thresholds(cat, win, q))
    defs = [[a, b, b] for a in (3, 4, 5, 6) for b in (0, 1, 2)]
    dm = cal.build_doy_map(dates)
    north, south, _ = cal.hemisphere_season_tables(dates)
    series = x.reshape(members * n_cells, T)                                # member-major
    hemi = np.tile(np.array([0, 1, 0], dtype=np.uint8), members)
    got = core.compute_heatwave_metrics(series, thr, dm, defs, north, south, hemi)
    want = c_oracle.metrics(series, np.concatenate([thr] * members), dm, defs, north, south, hemi)
    assert np.array_equal(got.astype(np.int64), want)


# ---- heat index pre-step ----------------------------------------------------------

def test_heat_index_matches_oracle_and_reference_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "heat_index.npz"))
    got = core.heat_index(g["temp_f"], g["rel_humid"])
    assert got.dtype == np.float32
    assert np.array_equal(got, orc.heat_index(g["temp_f"], g["rel_humid"]))          # bit-exact vs oracle
    np.testing.assert_allclose(got.astype(np.float64), g["reference_stub_run"], rtol=1e-5, atol=1e-4)
    rng = np.random.default_rng(0)
    t = rng.uniform(-40, 130, size=100003).astype(np.float32)                         # odd size: tail path
    r = rng.uniform(0, 100, size=100003).astype(np.float32)
    assert np.array_equal(core.heat_index(t, r), orc.heat_index(t, r))


def test_tiny_and_degenerate_shapes():
    """one cell / one percentile / one definition; a single year; a record too short for any
    complete season (Y = 0: empty metrics, like the reference's empty 'year' axis)."""
    rng = np.random.default_rng(99)
    # one year, one cell, P = D = 1
    dates = orc.noleap_date_range("2001-01-01", "2001-12-31")
    x = rng.normal(size=(1, dates.size)).astype(np.float32)
    ti, cols = cal.window_columns(dates, 7)
    q = [0.9]
    thr = core.compute_percentiles(x, ti, cols, q)
    assert same_f64(thr, orc.compute_thresholds_cells(x, cal.expand_window_table(ti, cols), q))
    dm = cal.build_doy_map(dates)
    north, south, years = cal.hemisphere_season_tables(dates)
    got = core.compute_heatwave_metrics(x, thr, dm, [[3, 0, 0]], north, south, np.zeros(1, np.uint8))
    want = orc.compute_metrics_cells(x, thr, dm, [[3, 0, 0]], north, south, np.zeros(1, np.uint8))
    assert got.shape == want.shape and np.array_equal(got.astype(np.int64), want)
    # 100 days: no complete season.  The reference's trimming keeps the (-1, -1) row
    # (metric.py:230-232) and then fails inside np.max on the empty slice; same error here.
    short = orc.noleap_date_range("2001-01-01", "2001-04-10")
    n2, s2, y2 = cal.hemisphere_season_tables(short)
    n3, s3, _ = orc.hemisphere_ranges(short)
    assert np.array_equal(n2, n3) and np.array_equal(s2, s3) and (n2 == -1).all()
    xs = rng.normal(size=(2, short.size)).astype(np.float32)
    with pytest.raises(ValueError, match="zero-size array"):
        core.compute_heatwave_metrics(xs, np.zeros((2, 365, 1)), cal.build_doy_map(short), [[3, 0, 0]], n2, s2,
                                      np.zeros(2, np.uint8))
    # an empty season table is accepted and yields an empty year axis
    empty = np.zeros((0, 2), dtype=np.int64)
    out = core.compute_heatwave_metrics(xs, np.zeros((2, 365, 1)), cal.build_doy_map(short), [[3, 0, 0]], empty,
                                        empty, np.zeros(2, np.uint8))
    assert out.shape == (1, 1, 2, 4, 0)
    # window radius larger than the record's day-of-year count is an IndexError in the reference too
    with pytest.raises(IndexError):
        cal.window_columns(orc.noleap_date_range("2001-01-01", "2001-01-05"), 7)


def test_thresholds_extreme_magnitudes_and_subnormals():
    """float32 subnormals, +-FLT_MAX-scale values, signed zeros and mixed signs survive the
    key transform, the med3-based sort and the integer merge unchanged."""
    rng = np.random.default_rng(404)
    dates = orc.noleap_date_range("2001-01-01", "2008-12-31")
    T = dates.size
    x = np.empty((5, T), dtype=np.float32)
    x[0] = (rng.integers(-2000, 2000, size=T) * np.float32(1e-42)).astype(np.float32)   # subnormals of both signs
    x[1] = rng.choice(np.array([3.0e38, -3.0e38, 1.0, -1.0, 0.0, -0.0], dtype=np.float32), size=T)
    x[2] = rng.normal(0, 1e-30, size=T).astype(np.float32)
    x[3] = np.where(rng.random(T) < 0.5, np.float32(0.0), np.float32(-0.0))
    x[4] = rng.normal(-250.0, 40.0, size=T).astype(np.float32)                          # all negative
    assert np.any((np.abs(x[0]) > 0) & (np.abs(x[0]) < 1.2e-38))
    ti, cols = cal.window_columns(dates, 7)
    q = [0.0, 0.05, 0.5, 0.9, 0.99, 1.0]
    with np.errstate(all="ignore"):
        want = orc.compute_thresholds_cells(x, cal.expand_window_table(ti, cols), q)
    got = core.compute_percentiles(x, ti, cols, q)
    assert same_f64(got, want)
    assert same_f64(core.compute_percentiles_table(x, cal.expand_window_table(ti, cols), q), want)


def test_weighted_spatial_mean_matches_numpy_restatement():
    """compute_weighted_spatial_mean (figure.py:14-15) on the device: float64 rows with NaNs through the host entry
    point, int16 metrics in the device layout through the _dev entry point; 1e-12 relative against the NumPy
    restatement (different but fixed summation order; xarray itself is not importable: parity unpinned)."""
    import ctypes as C
    from hdp_amd import _lib, figure, minixr
    rng = np.random.default_rng(8)
    n_lat, n_lon = 37, 53
    lat = np.linspace(-90, 90, n_lat)
    lon = np.linspace(0, 360, n_lon, endpoint=False)
    v = rng.integers(0, 150, size=(3, 2, n_lat, n_lon, 5)).astype(np.int64)      # (percentile, definition, lat, lon, time)
    da = minixr.DataArray(v, dims=["percentile", "definition", "lat", "lon", "time"],
                          coords={"percentile": [0.9, 0.95, 0.99], "definition": ["a", "b"], "lat": lat, "lon": lon,
                                  "time": np.arange(5)}, name="HWF")
    got = figure.compute_weighted_spatial_mean(da)
    assert tuple(got.dims) == ("percentile", "definition", "time") and got.shape == (3, 2, 5)
    want = orc.weighted_spatial_mean(np.moveaxis(v, (2, 3), (-2, -1)), lat, n_lon)
    assert np.allclose(got.values, want, rtol=1e-12, atol=0)
    f = rng.normal(0, 3, size=(n_lat, n_lon, 4))
    f[rng.random(f.shape) < 0.2] = np.nan
    f[:, :, 3] = np.nan                                                       # no valid value: NaN
    daf = minixr.DataArray(f, dims=["lat", "lon", "time"], coords={"lat": lat, "lon": lon, "time": np.arange(4)})
    gotf = figure.compute_weighted_spatial_mean(daf).values
    wantf = orc.weighted_spatial_mean(np.moveaxis(f, 2, 0), lat, n_lon)
    assert np.isnan(gotf[3]) and np.allclose(gotf[:3], wantf[:3], rtol=1e-12, atol=0)
    # device layout: int16 rows of n series (vector path: n % 8 == 0; scalar path otherwise)
    lib = _lib.ensure_device()
    for n in (4096, 1001):
        rows = rng.integers(-5, 200, size=(7, n)).astype(np.int16)
        w = np.cos(np.deg2rad(rng.uniform(-90, 90, n)))
        dv, dw = core.DeviceArray.from_host(rows), core.DeviceArray.from_host(w)
        dout = core.DeviceArray((7,), np.float64)
        _lib.check(lib.hdp_weighted_mean_i16_dev(dv.ptr, 7, n, dw.ptr, dout.ptr, None))
        _lib.check(lib.hdp_sync(None))
        out = dout.to_host()
        assert np.allclose(out, (rows.astype(np.float64) * w).sum(axis=1) / w.sum(), rtol=1e-12, atol=0)


def test_metric_planes_entry_point_matches_block_order():
    """hdp_metrics_f32_planes_i64: int64 [4][P][D][series][Y] planes == the int16 block-order result regrouped;
    several chunks (ragged last one) through the 2-D downloads, shared member thresholds."""
    defs = [[3, 0, 0], [2, 1, 1], [4, 2, 0]]
    case = _random_metrics_case(77, 6, 45, 3, defs, trend=1.1)
    blocks = core.compute_heatwave_metrics(*case)
    planes = core.compute_heatwave_metric_planes(*case)
    assert planes.dtype == np.int64 and planes.shape == (4,) + blocks.shape[:3] + blocks.shape[4:]
    assert np.array_equal(planes, np.moveaxis(blocks.astype(np.int64), 3, 0))
    assert np.array_equal(np.moveaxis(planes, 0, 3), orc.compute_metrics_cells(*case))
