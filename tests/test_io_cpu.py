"""compute_threshold_io / compute_metrics_io (hdp/threshold.py:232-289, hdp/metric.py:526-590): path checks
with the reference's exceptions, variable lookup, and latitude-band streaming giving the same Dataset as one
pass.  No xarray / netCDF4 / zarr in the image, so files are an in-memory dict behind the stand-in container
and the GPU calls are replaced by the oracle (the kernels are covered by the -m gpu tests)."""
import re

import numpy as np
import pytest

import hdp_amd._xr
import hdp_amd.metric
import hdp_amd.threshold
from hdp_amd import calendar as cal
from hdp_amd import core, minixr, utils
from oracle import hdp_oracle as orc
from tests.helpers import measure_dataset

pytestmark = pytest.mark.skipif(hdp_amd._xr.backend() is not minixr, reason="in-memory store is built on the stand-in")


@pytest.fixture()
def store(monkeypatch):
    """path -> Dataset 'filesystem' + oracle in place of the device."""
    files, opened = {}, []

    def fake_open(path):
        opened.append(str(path))
        return files[str(path)]

    monkeypatch.setattr(minixr, "open_dataset", fake_open, raising=False)
    monkeypatch.setattr(minixr, "open_zarr", fake_open, raising=False)
    monkeypatch.setattr(minixr.Dataset, "to_netcdf", lambda self, path: files.__setitem__(str(path), self), raising=False)
    monkeypatch.setattr(minixr.Dataset, "to_zarr", lambda self, path: files.__setitem__(str(path), self), raising=False)

    def fake_percentiles(x, time_index, cols, q):
        return orc.compute_thresholds_cells(np.ascontiguousarray(x), cal.expand_window_table(time_index, cols), q)

    def fake_metrics(x, thr, doy_map, defs, north, south, is_south):
        x = np.ascontiguousarray(x)
        full = thr[np.arange(x.shape[0]) % thr.shape[0]]
        return orc.compute_metrics_cells(x, full, doy_map, defs, north, south, is_south).astype(np.int16)

    monkeypatch.setattr(core, "compute_percentiles", fake_percentiles)
    monkeypatch.setattr(core, "compute_heatwave_metrics", fake_metrics)
    monkeypatch.setattr(core, "compute_heatwave_metric_planes",
                        lambda *a: np.ascontiguousarray(np.moveaxis(fake_metrics(*a).astype(np.int64), 3, 0)))
    return files, opened


def _grids():
    base, lon, lat, bdates = utils.generate_control_array(start_date="1700-01-01", end_date="1703-12-31",
                                                          grid_shape=(2, 5), add_noise=True, seed=3)
    warm, _, _, mdates = utils.generate_warming_array(start_date="2000-01-01", end_date="2003-12-31",
                                                      grid_shape=(2, 5), add_noise=True)
    return base, warm, lon, lat, bdates, mdates


def same_dataset(a, b):
    assert sorted(a.data_vars) == sorted(b.data_vars)
    for name in a.data_vars:
        x, y = a[name], b[name]
        assert tuple(x.dims) == tuple(y.dims) and x.dtype == y.dtype
        assert np.array_equal(x.values, y.values, equal_nan=x.dtype.kind == "f")
        # history entries carry a wall-clock stamp to the second (hdp/utils.py:10-20): compare them without it
        strip = lambda a: {k: (re.sub(r"\(\d{4}-[^)]*\)", "(stamp)", v) if k == "history" else v) for k, v in a.items()}
        assert strip(x.attrs) == strip(y.attrs)
    for k in a.coords:
        assert np.array_equal(np.asarray(a.coords[k].values), np.asarray(b.coords[k].values))


def test_output_path_checks(tmp_path, store):
    files, _ = store
    base, _, lon, lat, bdates, _ = _grids()
    files[str(tmp_path / "base.nc")] = measure_dataset(base, lon, lat, bdates)
    q = [0.9, 0.95]
    existing = tmp_path / "out.nc"
    existing.write_bytes(b"")
    with pytest.raises(FileExistsError, match="file exists"):
        hdp_amd.threshold.compute_threshold_io(tmp_path / "base.nc", "temp", existing, q)
    with pytest.raises(FileExistsError, match="does not exist"):
        hdp_amd.threshold.compute_threshold_io(tmp_path / "base.nc", "temp", tmp_path / "missing" / "out.nc", q)
    with pytest.raises(ValueError, match="not supported"):
        hdp_amd.threshold.compute_threshold_io(tmp_path / "base.nc", "temp", tmp_path / "out.h5", q)
    with pytest.raises(FileExistsError):
        hdp_amd.metric.compute_metrics_io(existing, tmp_path / "m.nc", "temp", tmp_path / "t.nc", [[3, 0, 0]])
    with pytest.raises(ValueError, match="not supported"):
        hdp_amd.metric.compute_metrics_io(tmp_path / "out.csv", tmp_path / "m.nc", "temp", tmp_path / "t.nc", [[3, 0, 0]])
    # overwrite: an existing output is replaced and a missing parent directory is created
    hdp_amd.threshold.compute_threshold_io(tmp_path / "base.nc", "temp", existing, q, overwrite=True)
    hdp_amd.threshold.compute_threshold_io(tmp_path / "base.nc", "temp", tmp_path / "new" / "out.zarr", q, overwrite=True)
    assert (tmp_path / "new").is_dir() and str(tmp_path / "new" / "out.zarr") in files


def test_threshold_io_matches_in_memory_call_and_bands(tmp_path, store):
    files, opened = store
    base, _, lon, lat, bdates, _ = _grids()
    src = tmp_path / "base.nc"
    files[str(src)] = measure_dataset(base, lon, lat, bdates)
    q = np.arange(0.9, 1, 0.03)
    want = hdp_amd.threshold.compute_threshold(measure_dataset(base, lon, lat, bdates)["temp"], q)
    hdp_amd.threshold.compute_threshold_io(src, "temp", tmp_path / "thr.nc", q)
    got = files[str(tmp_path / "thr.nc")]
    assert opened == [str(src)]
    assert got["temp_threshold"].attrs.pop("baseline_source") == str(src)
    same_dataset(got, want)
    for band in (1, 2, 5, 64):
        out = tmp_path / f"thr_band{band}.zarr"
        hdp_amd.threshold.compute_threshold_io(src, "temp", out, q, lat_band=band)
        banded = files[str(out)]
        assert banded["temp_threshold"].attrs.pop("baseline_source") == str(src)
        same_dataset(banded, want)


def test_metrics_io_matches_in_memory_call_and_bands(tmp_path, store):
    files, _ = store
    base, warm, lon, lat, bdates, mdates = _grids()
    q = [0.9, 0.97]
    defs = [[3, 0, 0], [3, 1, 1], [4, 2, 1]]
    thr = hdp_amd.threshold.compute_threshold(measure_dataset(base, lon, lat, bdates)["temp"], q)
    meas = measure_dataset(warm, lon, lat, mdates)
    files[str(tmp_path / "thr.zarr")] = thr
    files[str(tmp_path / "meas.nc")] = meas
    want = hdp_amd.metric.compute_individual_metrics(meas["temp"], thr["temp_threshold"], defs, check_variables=False)
    # default threshold variable: the documented name is absent, the name compute_threshold writes is used
    hdp_amd.metric.compute_metrics_io(tmp_path / "hw.nc", tmp_path / "meas.nc", "temp", tmp_path / "thr.zarr", defs)
    same_dataset(files[str(tmp_path / "hw.nc")], want)
    for band in (1, 3):
        out = tmp_path / f"hw_band{band}.nc"
        hdp_amd.metric.compute_metrics_io(out, tmp_path / "meas.nc", "temp", tmp_path / "thr.zarr", defs,
                                          override_threshold_var="temp_threshold", lat_band=band)
        same_dataset(files[str(out)], want)
    # an explicit variable is checked like compute_individual_metrics does (metric.py:394-398)
    bad = hdp_amd.threshold.compute_threshold(measure_dataset(base, lon, lat, bdates, name="other")["other"], q)
    files[str(tmp_path / "bad.nc")] = bad
    with pytest.raises(AssertionError):
        hdp_amd.metric.compute_metrics_io(tmp_path / "x.nc", tmp_path / "meas.nc", "temp", tmp_path / "bad.nc", defs,
                                          override_threshold_var="other_threshold")


def test_io_without_xarray_fails_loudly(tmp_path):
    with pytest.raises(ImportError, match="needs xarray"):
        hdp_amd.threshold.compute_threshold_io(tmp_path / "base.nc", "temp", tmp_path / "out.nc", [0.9])


def test_chunked_inputs_are_walked_block_by_block(store, monkeypatch):
    """an input that exposes dask-style ``.chunks`` is processed one block at a time along its
    first chunked non-time dimension (thresholds sliced alike for the metrics) and gives the same Datasets."""
    base, warm, lon, lat, bdates, mdates = _grids()
    q = [0.9, 0.95]
    defs = [[3, 0, 0], [3, 1, 1]]
    whole_thr = hdp_amd.threshold.compute_threshold(measure_dataset(base, lon, lat, bdates)["temp"], q)
    bda = measure_dataset(base, lon, lat, bdates)["temp"]                  # dims (lon, lat, time)
    bda.chunks = ((2,), (2, 2, 1), (bdates.size,))                          # three blocks along lat
    sizes = []
    real = core.compute_percentiles
    monkeypatch.setattr(core, "compute_percentiles", lambda x, *a: (sizes.append(x.shape[0]), real(x, *a))[1])
    thr = hdp_amd.threshold.compute_threshold(bda, q)
    assert sizes == [4, 4, 2]                                               # lon x lat cells per block
    same_dataset(thr, whole_thr)
    mda = measure_dataset(warm, lon, lat, mdates)["temp"]
    whole = hdp_amd.metric.compute_individual_metrics(mda, whole_thr["temp_threshold"], defs, check_variables=False)
    mda.chunks = ((1, 1), (5,), (mdates.size,))                             # two blocks along lon
    calls = []
    real_m = core.compute_heatwave_metric_planes
    monkeypatch.setattr(core, "compute_heatwave_metric_planes", lambda x, *a: (calls.append(x.shape[0]), real_m(x, *a))[1])
    got = hdp_amd.metric.compute_individual_metrics(mda, whole_thr["temp_threshold"], defs, check_variables=False)
    assert calls == [5, 5]
    same_dataset(got, whole)


def test_iter_bands_prefetches_the_next_band_and_keeps_order():
    """hio.iter_bands: band k + 1 is fetched (isel + load) on a helper thread while the consumer holds band k; bands
    come in order, each loaded exactly once, and an exception raised while fetching reaches the consumer."""
    import threading
    import time as _time
    from hdp_amd import io as hio

    log = []

    class Lazy:
        def __init__(self, lo=0, hi=10):
            self.lo, self.hi = lo, hi

        def isel(self, lat):
            return Lazy(self.lo + lat.start, self.lo + lat.stop)

        def load(self):
            log.append(("load", self.lo, self.hi, threading.current_thread() is threading.main_thread()))
            if self.lo == 6:
                raise OSError("read failed")
            _time.sleep(0.05)
            return self

    got = []
    with pytest.raises(OSError, match="read failed"):
        for (band,) in hio.iter_bands((Lazy(),), [(0, 3), (3, 6), (6, 9), (9, 10)]):
            got.append((band.lo, band.hi))
            _time.sleep(0.02)
    assert got == [(0, 3), (3, 6)]                                  # the failing band is never delivered
    assert [e[1:3] for e in log] == [(0, 3), (3, 6), (6, 9)]        # each band loaded once, in order
    assert log[0][3] and not log[1][3] and not log[2][3]            # the first on the caller's thread, the rest prefetched
    # two variables travel together (measure and thresholds of compute_metrics_io)
    pairs = list(hio.iter_bands((Lazy(), Lazy(100, 110)), [(0, 5), (5, 10)]))
    assert [(a.lo, b.lo) for a, b in pairs] == [(0, 100), (5, 105)]
