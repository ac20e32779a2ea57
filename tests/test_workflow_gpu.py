"""End-to-end drop-in test, modelled on the reference's hdp/tests/test_workflow.py:15-66
(2x3 grid, control -> thresholds for arange(0.9, 1, 0.01), warming -> metrics for six
definitions) with the numeric checks the reference lacks added against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import hdp_amd.metric  # noqa: E402
import hdp_amd.threshold  # noqa: E402
from hdp_amd import calendar as cal, core, utils  # noqa: E402
from oracle import c_oracle  # noqa: E402
from oracle import hdp_oracle as orc  # noqa: E402
from tests.helpers import measure_dataset  # noqa: E402


def test_full_data_workflow():
    grid_shape = (2, 3)
    base, lon, lat, bdates = utils.generate_control_array(grid_shape=grid_shape)
    baseline_measures = measure_dataset(base, lon, lat, bdates, "temp")
    percentiles = np.arange(0.9, 1, 0.01)
    thresholds = hdp_amd.threshold.compute_thresholds(baseline_measures, percentiles=percentiles)

    warm, _, _, mdates = utils.generate_warming_array(grid_shape=grid_shape)
    hw_definitions = [[3, 0, 0], [3, 1, 1], [4, 2, 0], [4, 1, 3], [5, 0, 1], [5, 1, 4]]
    test_measures = measure_dataset(warm, lon, lat, mdates, "temp")
    metrics = hdp_amd.metric.compute_group_metrics(test_measures, thresholds, hw_definitions)

    metrics = metrics.compute()
    thresholds = thresholds.compute()

    assert (thresholds.percentile.values == percentiles).all()
    assert len(thresholds.data_vars) == 1
    thr = thresholds["temp_threshold"]
    assert tuple(thr.dims) == ("lon", "lat", "doy", "percentile") and thr.dtype == np.float64
    assert thr.attrs["hdp_type"] == "threshold" and thr.attrs["baseline_variable"] == "temp"
    assert thr.attrs["baseline_calendar"] == "noleap" and thr.attrs["param_rolling_window_size"] == "7"

    assert list(metrics.definition.values) == ["3-0-0", "3-1-1", "4-2-0", "4-1-3", "5-0-1", "5-1-4"]
    assert (metrics.percentile.values == percentiles).all()
    means = metrics.mean()
    assert means["temp.temp_threshold.HWF"] >= means["temp.temp_threshold.HWD"]
    assert means["temp.temp_threshold.HWD"] >= means["temp.temp_threshold.HWA"]
    for var in metrics:
        assert metrics[var].shape == (metrics.percentile.size, metrics.definition.size, metrics.lon.size,
                                      metrics.lat.size, metrics.time.size)
        assert metrics[var].dtype == int
        if "HWF" in var or "HWD" in var:
            assert metrics[var].attrs["units"] == "heatwave days"
        elif "HWN" in var or "HWA" in var:
            assert metrics[var].attrs["units"] == "heatwave events"
        else:
            assert False, var
    assert metrics.attrs["variable_naming_delimeter"] == "."
    assert str(metrics.time.values[0]).startswith("2000-01-01")

    # numeric parity the reference's own test does not assert
    x = base.astype(np.float32).reshape(-1, base.shape[-1])
    want_thr = orc.compute_thresholds_cells(x, orc.datetimes_to_windows(bdates, 7), percentiles)
    assert np.array_equal(thr.values.reshape(want_thr.shape), want_thr)
    m = warm.astype(np.float32).reshape(-1, warm.shape[-1])
    north, south, _ = orc.hemisphere_ranges(mdates)
    is_south = np.repeat((lat < 0)[None, :], grid_shape[0], axis=0).reshape(-1)
    want = orc.compute_metrics_cells(m, want_thr, orc.build_doy_map(mdates), hw_definitions, north, south, is_south)
    for i, name in enumerate(("HWF", "HWN", "HWD", "HWA")):
        got = metrics[f"temp.temp_threshold.{name}"].values
        assert np.array_equal(got.reshape(got.shape[:2] + (-1, got.shape[-1])), want[:, :, :, i, :]), name


def test_time_major_and_member_inputs():
    """CMIP layout (time, lat, lon) and an ensemble 'member' dim: members are concatenated
    along time for thresholds (threshold.py:114-119) and share thresholds for metrics."""
    rng = np.random.default_rng(3)
    dates = utils.noleap_date_range("2001-01-01", "2004-12-31")
    T, n_lat, n_lon, n_mem = dates.size, 3, 2, 2
    lat = np.array([-30.0, 0.0, 45.0]); lon = np.array([10.0, 20.0])
    data = rng.normal(10, 3, size=(n_mem, T, n_lat, n_lon)).astype(np.float32)
    ds = measure_dataset(data, lon, lat, dates, "tas", dims=("member", "time", "lat", "lon"),
                         extra_coords={"member": np.arange(n_mem)})
    q = [0.9, 0.95]
    thr_ds = hdp_amd.threshold.compute_thresholds(ds, q)
    thr = thr_ds["tas_threshold"]
    assert tuple(thr.dims) == ("lat", "lon", "doy", "percentile")
    # oracle: concatenate members along time per cell
    cat = np.concatenate([data[m] for m in range(n_mem)], axis=0)          # [n_mem*T, lat, lon]
    x = np.moveaxis(cat, 0, -1).reshape(-1, n_mem * T)
    win = orc.datetimes_to_windows(np.concatenate([dates] * n_mem), 7)
    want_thr = orc.compute_thresholds_cells(x, win, q)
    assert np.array_equal(thr.values.reshape(want_thr.shape), want_thr)

    defs = [[3, 0, 0], [2, 1, 1]]
    met = hdp_amd.metric.compute_group_metrics(ds, thr_ds, defs)
    hwf = met["tas.tas_threshold.HWF"]
    assert tuple(hwf.dims) == ("percentile", "definition", "member", "lat", "lon", "time")
    north, south, years = orc.hemisphere_ranges(dates)
    dm = orc.build_doy_map(dates)
    for m in range(n_mem):
        xm = np.moveaxis(data[m], 0, -1).reshape(-1, T)
        hemi = np.repeat((lat < 0)[:, None], n_lon, axis=1).reshape(-1)
        want = orc.compute_metrics_cells(xm, want_thr, dm, defs, north, south, hemi)
        for i, name in enumerate(("HWF", "HWN", "HWD", "HWA")):
            got = met[f"tas.tas_threshold.{name}"].values[:, :, m]
            assert np.array_equal(got.reshape(got.shape[:2] + (-1, got.shape[-1])), want[:, :, :, i, :])


def test_full_data_workflow_with_relative_humidity():
    """The reference's own workflow test shape (hdp/tests/test_workflow.py:15-61): temperature +
    relative humidity -> format_standard_measures -> two measures (temp, temp_hi) -> thresholds for
    both -> metrics for both pairs; plus numeric parity of the heat-index branch against the oracle."""
    import hdp_amd.measure
    from hdp_amd._xr import backend
    xr = backend()
    grid_shape = (2, 3)

    def da(values, lon, lat, dates, name, units):
        return xr.DataArray(values, dims=["lon", "lat", "time"], coords={"lon": lon, "lat": lat, "time": dates},
                            name=name, attrs={"units": units})

    base, lon, lat, bdates = utils.generate_control_array(grid_shape=grid_shape)
    rh_vals = np.abs(base / base.max() - 0.3)                                  # utils.py:45-50
    baseline_temp = da(base, lon, lat, bdates, "temp", "degC")
    baseline_rh = da(rh_vals, lon, lat, bdates, "rh", "g/g")
    baseline_measures = hdp_amd.measure.format_standard_measures([baseline_temp], rh=baseline_rh)
    percentiles = np.arange(0.9, 1, 0.01)
    thresholds = hdp_amd.threshold.compute_thresholds(baseline_measures, percentiles=percentiles)
    warm, _, _, mdates = utils.generate_warming_array(grid_shape=grid_shape)
    test_temp = da(warm, lon, lat, mdates, "temp", "degC")
    test_rh = da(rh_vals, lon, lat, mdates, "rh", "g/g")
    hw_definitions = [[3, 0, 0], [3, 1, 1], [4, 2, 0], [4, 1, 3], [5, 0, 1], [5, 1, 4]]
    test_measures = hdp_amd.measure.format_standard_measures([test_temp], rh=test_rh)
    metrics = hdp_amd.metric.compute_group_metrics(test_measures, thresholds, hw_definitions).compute()

    assert (thresholds.percentile.values == percentiles).all()
    assert len(thresholds.data_vars) == 2                                      # test_workflow.py:40
    assert sorted(metrics.data_vars) == sorted(f"{m}.{m}_threshold.{k}" for m in ("temp", "temp_hi")
                                               for k in ("HWF", "HWN", "HWD", "HWA"))
    means = metrics.mean()
    assert means["temp.temp_threshold.HWF"] >= means["temp.temp_threshold.HWD"] >= means["temp.temp_threshold.HWA"]
    for var in metrics:
        assert metrics[var].shape == (10, 6, 2, 3, 50) and metrics[var].dtype == int

    # heat-index measure: bit-exact against the oracle's restatement of the same float32/float64 steps
    want_hi = orc.heat_index_celsius(warm.astype(np.float32), (rh_vals.astype(np.float32) * np.float32(100)))
    assert np.array_equal(test_measures["temp_hi"].values, want_hi)
    x = baseline_measures["temp_hi"].values.reshape(-1, base.shape[-1])
    want_thr = orc.compute_thresholds_cells(x, orc.datetimes_to_windows(bdates, 7), percentiles)
    assert np.array_equal(thresholds["temp_hi_threshold"].values.reshape(want_thr.shape), want_thr)


def test_io_wrappers_band_streaming_on_the_device(tmp_path, monkeypatch):
    """compute_threshold_io / compute_metrics_io through the real kernels: streaming the grid in bands of
    latitude rows gives the same Datasets as one pass (files are an in-memory dict: no xarray I/O here)."""
    from hdp_amd import _xr, minixr
    if _xr.backend() is not minixr:
        pytest.skip("in-memory store is built on the stand-in container")
    files = {}
    monkeypatch.setattr(minixr, "open_dataset", lambda path: files[str(path)], raising=False)
    monkeypatch.setattr(minixr.Dataset, "to_netcdf", lambda self, path: files.__setitem__(str(path), self), raising=False)
    base, lon, lat, bdates = utils.generate_control_array(start_date="1700-01-01", end_date="1709-12-31",
                                                          grid_shape=(3, 7), add_noise=True, seed=5)
    warm, _, _, mdates = utils.generate_warming_array(start_date="2000-01-01", end_date="2009-12-31",
                                                      grid_shape=(3, 7), add_noise=True)
    files[str(tmp_path / "base.nc")] = measure_dataset(base, lon, lat, bdates)
    files[str(tmp_path / "meas.nc")] = measure_dataset(warm, lon, lat, mdates)
    q = np.arange(0.9, 1, 0.02)
    defs = [[3, 0, 0], [3, 1, 1], [5, 2, 2]]
    hdp_amd.threshold.compute_threshold_io(tmp_path / "base.nc", "temp", tmp_path / "thr.nc", q)
    hdp_amd.threshold.compute_threshold_io(tmp_path / "base.nc", "temp", tmp_path / "thr_b.nc", q, lat_band=2)
    whole, banded = files[str(tmp_path / "thr.nc")], files[str(tmp_path / "thr_b.nc")]
    assert np.array_equal(whole["temp_threshold"].values, banded["temp_threshold"].values)
    assert np.array_equal(whole["lat"].values, banded["lat"].values)
    hdp_amd.metric.compute_metrics_io(tmp_path / "hw.nc", tmp_path / "meas.nc", "temp", tmp_path / "thr.nc", defs)
    hdp_amd.metric.compute_metrics_io(tmp_path / "hw_b.nc", tmp_path / "meas.nc", "temp", tmp_path / "thr_b.nc", defs,
                                      override_threshold_var="temp_threshold", lat_band=3)
    for name in ("HWF", "HWN", "HWD", "HWA"):
        a, b = files[str(tmp_path / "hw.nc")][name], files[str(tmp_path / "hw_b.nc")][name]
        assert tuple(a.dims) == tuple(b.dims) and np.array_equal(a.values, b.values) and a.values.sum() >= 0
    assert files[str(tmp_path / "hw.nc")]["HWF"].values.sum() > 0
    # ... and both are what the ORACLE computes from the same arrays (not only each other): thresholds bit for bit,
    # metrics as the int64 planes (percentile, definition, lon, lat, year)
    xb = base.astype(np.float32).reshape(-1, base.shape[-1])
    want_thr = orc.compute_thresholds_cells(xb, orc.datetimes_to_windows(bdates, 7), q)          # [cells, doy, P]
    assert np.array_equal(banded["temp_threshold"].values.reshape(want_thr.shape), want_thr)
    north, south, _ = orc.hemisphere_ranges(mdates)
    is_south = np.repeat((lat < 0)[None, :], base.shape[0], axis=0).reshape(-1)
    xm = warm.astype(np.float32).reshape(-1, warm.shape[-1])
    want = orc.compute_metrics_cells(xm, want_thr, orc.build_doy_map(mdates), defs, north, south, is_south)  # [P, D, n, 4, Y]
    for i, name in enumerate(("HWF", "HWN", "HWD", "HWA")):
        got = files[str(tmp_path / "hw_b.nc")][name].values                                       # [P, D, lon, lat, Y]
        assert got.dtype == np.int64 and np.array_equal(got.reshape(want.shape[0], want.shape[1], -1, want.shape[4]),
                                                        want[:, :, :, i, :])


def test_chunked_inputs_walk_block_by_block_on_the_device():
    """SURVEY 8f row 3 on the GPU: inputs that expose dask-style ``.chunks`` are pushed through the kernels one block at
    a time (the next block fetched on a helper thread meanwhile) and give what the oracle computes for the whole grid."""
    base, lon, lat, bdates = utils.generate_control_array(start_date="1700-01-01", end_date="1707-12-31",
                                                          grid_shape=(4, 9), add_noise=True, seed=11)
    warm, _, _, mdates = utils.generate_warming_array(start_date="2000-01-01", end_date="2007-12-31",
                                                      grid_shape=(4, 9), add_noise=True)
    q = [0.9, 0.93, 0.99]
    defs = [[3, 0, 0], [2, 1, 1], [4, 2, 2]]
    bda = measure_dataset(base, lon, lat, bdates)["temp"]                  # dims (lon, lat, time)
    bda.chunks = ((4,), (2, 3, 4), (bdates.size,))                          # three ragged blocks along lat
    thr = hdp_amd.threshold.compute_threshold(bda, q)
    xb = base.astype(np.float32).reshape(-1, base.shape[-1])
    want_thr = orc.compute_thresholds_cells(xb, orc.datetimes_to_windows(bdates, 7), q)
    assert np.array_equal(thr["temp_threshold"].values.reshape(want_thr.shape), want_thr)
    mda = measure_dataset(warm, lon, lat, mdates)["temp"]
    mda.chunks = ((1, 3), (9,), (mdates.size,))                             # two blocks along lon
    got = hdp_amd.metric.compute_individual_metrics(mda, thr["temp_threshold"], defs, check_variables=False)
    north, south, _ = orc.hemisphere_ranges(mdates)
    is_south = np.repeat((lat < 0)[None, :], base.shape[0], axis=0).reshape(-1)
    xm = warm.astype(np.float32).reshape(-1, warm.shape[-1])
    want = orc.compute_metrics_cells(xm, want_thr, orc.build_doy_map(mdates), defs, north, south, is_south)
    for i, name in enumerate(("HWF", "HWN", "HWD", "HWA")):
        assert np.array_equal(got[name].values.reshape(want.shape[0], want.shape[1], -1, want.shape[4]), want[:, :, :, i, :])


@pytest.mark.timeout(600)
def test_host_pointer_calls_with_several_chunks_match_one_chunk(monkeypatch):
    """The host-pointer entry points stream the series through the device in chunks, chunk k + 1 uploading (helper
    thread, second stream, second device buffer) under the kernels of chunk k: a call cut into many chunks returns
    exactly what a one-chunk call returns, for time-contiguous, time-major and arbitrarily strided inputs."""
    rng = np.random.default_rng(21)
    dates = utils.noleap_date_range("2001-01-01", "2008-12-31")
    T, n = dates.size, 300000 // 8                       # 37 500 series x 2 920 days = 438 MB: one chunk of the 1 GiB budget
    x = rng.normal(15, 4, size=(n, T)).astype(np.float32)
    ti, cols = cal.window_columns(dates, 7)
    q = [0.9, 0.95]
    dm = cal.build_doy_map(dates)
    north, south, _ = cal.hemisphere_season_tables(dates)
    hemi = (np.arange(n) % 2).astype(np.uint8)
    defs = [[3, 0, 0], [3, 1, 1]]
    thr1 = core.compute_percentiles(x, ti, cols, q)
    met1 = core.compute_heatwave_metrics(x, thr1, dm, defs, north, south, hemi)
    pick = np.arange(0, n, n // 48)
    want_thr = c_oracle.thresholds(x[pick], cal.expand_window_table(ti, cols), q)
    assert np.array_equal(thr1[pick], want_thr)
    assert np.array_equal(met1[:, :, pick].astype(np.int64),
                          c_oracle.metrics(x[pick], want_thr, dm, defs, north, south, hemi[pick]))
    big = np.concatenate([x] * 4)                        # 1.75 GB: several chunks, double-buffered
    thr4 = core.compute_percentiles(big, ti, cols, q)
    assert all(np.array_equal(thr4[i * n:(i + 1) * n], thr1) for i in range(4))
    met4 = core.compute_heatwave_metrics(big, thr4, dm, defs, north, south, np.tile(hemi, 4))
    assert all(np.array_equal(met4[:, :, i * n:(i + 1) * n], met1) for i in range(4))
    wide = np.empty((4 * n, T + 3), dtype=np.float32)    # a padded row pitch: packed on the host, chunk by chunk
    wide[:, :T] = big
    assert np.array_equal(core.compute_percentiles(wide[:, :T], ti, cols, q), thr4)
