"""numpy-level API over the C ABI (include/hdp_hip.h).

Array-level counterparts of the reference's two Numba kernels and of the njit helpers
its unit tests call directly.  Everything here runs on the GPU through
libhdp_hip.so; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _as_series_2d(x):
    """View x as [n_cells, T] float32 without copying when possible; returns
    (array, stride_cell, stride_time) in elements."""
    x = np.asarray(x)
    if x.dtype != np.float32:
        x = x.astype(np.float32)
    if x.ndim != 2:
        raise ValueError("expected a [n_cells, T] array")
    if x.size and (x.strides[0] % 4 or x.strides[1] % 4 or x.strides[0] < 0 or x.strides[1] < 0):
        x = np.ascontiguousarray(x)
    return x, x.strides[0] // 4, x.strides[1] // 4


# ---- thresholds ---------------------------------------------------------------------

def compute_percentiles(x, time_index, cols, percentiles):
    """Per-cell window quantiles: the gufunc compute_percentiles (threshold.py:52-78)
    for all cells at once, window table given in compact (time_index, cols) form
    (see hdp_amd.calendar.window_columns).

    x [n_cells, T] float32 (any strides) -> float64 [n_cells, n_doy, P]
    """
    lib = _lib.ensure_device()
    x, sc, st = _as_series_2d(x)
    ti = np.ascontiguousarray(time_index, dtype=np.int64)
    cl = np.ascontiguousarray(cols, dtype=np.int32)
    q = np.ascontiguousarray(np.asarray(percentiles, dtype=np.float64).ravel())
    n_cells, T = x.shape
    n_doy, S = ti.shape
    W = cl.shape[1]
    out = np.empty((n_cells, n_doy, q.size), dtype=np.float64)
    _lib.check(lib.hdp_thresholds_f32(_ptr(x), n_cells, T, sc, st, _ptr(ti), n_doy, S, _ptr(cl), W,
                                      _ptr(q), q.size, _ptr(out)))
    return out


def compute_percentiles_table(x, window_samples, percentiles):
    """Literal gufunc operands (threshold.py:53-57): window_samples int64 [n_doy, B]."""
    lib = _lib.ensure_device()
    x, sc, st = _as_series_2d(x)
    win = np.ascontiguousarray(window_samples, dtype=np.int64)
    q = np.ascontiguousarray(np.asarray(percentiles, dtype=np.float64).ravel())
    n_cells, T = x.shape
    out = np.empty((n_cells, win.shape[0], q.size), dtype=np.float64)
    _lib.check(lib.hdp_percentiles_table_f32(_ptr(x), n_cells, T, sc, st, _ptr(win), win.shape[0],
                                             win.shape[1], _ptr(q), q.size, _ptr(out)))
    return out


# ---- metrics ------------------------------------------------------------------------

def compute_heatwave_metrics(x, thresholds, doy_map, hw_definitions, north, south, is_south, planes=False):
    """compute_heatwave_metrics (metric.py:304-341) for every (percentile, definition,
    series) in one pass over the measure.

    x [n_series, T] f32; thresholds [n_thr_cells, n_doy, P] f64 with series c using row
    c % n_thr_cells; -> int16 [P, D, n_series, 4, Y], metric order HWF, HWN, HWD, HWA.
    """
    lib = _lib.ensure_device()
    x, sc, st = _as_series_2d(x)
    thr = np.ascontiguousarray(thresholds, dtype=np.float64)
    if thr.ndim != 3:
        raise ValueError("thresholds must be [n_thr_cells, n_doy, P]")
    dm = np.ascontiguousarray(doy_map, dtype=np.int64)
    defs = np.ascontiguousarray(np.asarray(hw_definitions, dtype=np.int64).reshape(-1, 3))
    north = np.ascontiguousarray(north, dtype=np.int64).reshape(-1, 2)
    south = np.ascontiguousarray(south, dtype=np.int64).reshape(-1, 2)
    hemi = np.ascontiguousarray(is_south, dtype=np.uint8)
    n, T = x.shape
    n_thr, n_doy, P = thr.shape
    D, Y = defs.shape[0], north.shape[0]
    if dm.size != T or hemi.size != n or south.shape[0] != Y:
        raise ValueError("inconsistent table sizes")
    if planes:
        out = np.zeros((4, P, D, n, Y), dtype=np.int64)
        fn = lib.hdp_metrics_f32_planes_i64
    else:
        out = np.zeros((P, D, n, 4, Y), dtype=np.int16)
        fn = lib.hdp_metrics_f32
    _lib.check(fn(_ptr(x), n, T, sc, st, _ptr(thr), n_thr, n_doy, P, _ptr(dm), _ptr(defs), D, _ptr(north),
                  _ptr(south), _ptr(hemi), Y, _ptr(out)))
    return out


def compute_heatwave_metrics_layout(x, thresholds, doy_map, hw_definitions, north, south, is_south):
    """The same pass, returned in the DEVICE layout: int16 [4, P, D, Y, n_series], series-minor -- the form a collective
    moves when the grid is sharded over ranks (hdp_amd.dist): 2 bytes per value, and shards of the last axis
    concatenate row by row."""
    lib = _lib.ensure_device()
    x, sc, st = _as_series_2d(x)
    thr = np.ascontiguousarray(thresholds, dtype=np.float64)
    dm = np.ascontiguousarray(doy_map, dtype=np.int64)
    defs = np.ascontiguousarray(np.asarray(hw_definitions, dtype=np.int64).reshape(-1, 3))
    north = np.ascontiguousarray(north, dtype=np.int64).reshape(-1, 2)
    south = np.ascontiguousarray(south, dtype=np.int64).reshape(-1, 2)
    hemi = np.ascontiguousarray(is_south, dtype=np.uint8)
    n, T = x.shape
    n_thr, n_doy, P = thr.shape
    D, Y = defs.shape[0], north.shape[0]
    if dm.size != T or hemi.size != n or south.shape[0] != Y:
        raise ValueError("inconsistent table sizes")
    out = np.zeros((4, P, D, Y, n), dtype=np.int16)
    if n:
        _lib.check(lib.hdp_metrics_f32_layout_i16(_ptr(x), n, T, sc, st, _ptr(thr), n_thr, n_doy, P, _ptr(dm), _ptr(defs),
                                                  D, _ptr(north), _ptr(south), _ptr(hemi), Y, _ptr(out)))
    return out


def compute_heatwave_metric_planes_sharded(x, thresholds, doy_map, hw_definitions, north, south, is_south, n_mem,
                                           n_total):
    """Collective over the library's RCCL communicator (hdp_amd.dist.comm_init_rank / init_from_env): this rank passes
    ITS cells -- x [n_mem * n_loc, T] member-major, thresholds [n_loc, n_doy, P] -- of a grid of n_total cells; the int16
    result is all-gathered on the device and widened once there.  -> (int64 [4, P, D, n_mem * n_total, Y], bytes this
    rank handed to the collective); every rank receives the whole grid's planes."""
    lib = _lib.ensure_device()
    # The call is a collective: a rank that fails in its own preparation (shapes that do not fit, an allocation) must
    # still meet the others in the library's status exchange, or they wait in it for ever.  Such a rank makes the C call
    # with arguments the library refuses (n_mem = 0): every rank then gets an error back, and this one re-raises its own.
    try:
        x, sc, st = _as_series_2d(x)
        thr = np.ascontiguousarray(thresholds, dtype=np.float64)
        dm = np.ascontiguousarray(doy_map, dtype=np.int64)
        defs = np.ascontiguousarray(np.asarray(hw_definitions, dtype=np.int64).reshape(-1, 3))
        north = np.ascontiguousarray(north, dtype=np.int64).reshape(-1, 2)
        south = np.ascontiguousarray(south, dtype=np.int64).reshape(-1, 2)
        hemi = np.ascontiguousarray(is_south, dtype=np.uint8)
        n, T = x.shape
        n_loc, n_doy, P = thr.shape
        D, Y = defs.shape[0], north.shape[0]
        if n != n_mem * n_loc or dm.size != T or hemi.size != n or south.shape[0] != Y:
            raise ValueError("inconsistent table sizes")
        out = np.zeros((4, P, D, int(n_mem) * int(n_total), Y), dtype=np.int64)
    except Exception:
        lib.hdp_metrics_f32_planes_i64_sharded(None, 0, 0, 1, 1, 1, None, 1, 1, None, None, 1, None, None, None, 0, 0, None,
                                               None)
        raise
    wire = C.c_int64(0)
    _lib.check(lib.hdp_metrics_f32_planes_i64_sharded(_ptr(x), int(n_mem), n_loc, T, sc, st, _ptr(thr), n_doy, P, _ptr(dm),
                                                      _ptr(defs), D, _ptr(north), _ptr(south), _ptr(hemi), Y,
                                                      int(n_total), _ptr(out), C.byref(wire)))
    return out, int(wire.value)


def regroup_gathered_planes(gathered, world, n_mem, n_total):
    """The widening / regrouping half of the sharded call on a host-built gathered buffer (no communicator): gathered
    int16 [world, 4, P, D, Y, n_mem * shard], shard = ceil(n_total / world), zero columns past a rank's cells ->
    int64 [4, P, D, n_mem * n_total, Y].  Unit-level: pins the multi-rank layout on one GPU."""
    lib = _lib.ensure_device()
    g = np.ascontiguousarray(gathered, dtype=np.int16)
    w, four, P, D, Y, pad = g.shape
    shard = -(-int(n_total) // int(world)) if n_total else 0
    if w != world or four != 4 or pad != n_mem * shard:
        raise ValueError("gathered buffer does not have the layout [world, 4, P, D, Y, n_mem * ceil(n_total / world)]")
    out = np.zeros((4, P, D, int(n_mem) * int(n_total), Y), dtype=np.int64)
    _lib.check(lib.hdp_metrics_planes_i64_regroup(_ptr(g), int(world), int(n_mem), int(n_total), P, D, Y, _ptr(out)))
    return out


def compute_heatwave_metric_planes(x, thresholds, doy_map, hw_definitions, north, south, is_south):
    """The same pass, returned as int64 [4, P, D, n_series, Y]: one contiguous plane per output variable
    (HWF, HWN, HWD, HWA) in the dims and dtype compute_individual_metrics hands to xarray (metric.py:418-431)."""
    return compute_heatwave_metrics(x, thresholds, doy_map, hw_definitions, north, south, is_south, planes=True)


# ---- unit-level mirrors of the njit helpers (metric.py:11-172, 280-301) ----------------

def index_heatwaves(hot_days_ts, min_duration, max_break, max_subs):
    lib = _lib.ensure_device()
    hot = np.ascontiguousarray(np.asarray(hot_days_ts).astype(bool).astype(np.uint8))
    one = hot.ndim == 1
    hot2 = hot.reshape(1, -1) if one else hot
    ids = np.zeros(hot2.shape, dtype=np.int64)
    _lib.check(lib.hdp_index_heatwaves(_ptr(hot2), hot2.shape[0], hot2.shape[1], int(min_duration),
                                       int(max_break), int(max_subs), _ptr(ids)))
    return ids[0] if one else ids


def _season_metrics(hw_ts, season_ranges):
    lib = _lib.ensure_device()
    ids = np.ascontiguousarray(np.asarray(hw_ts).astype(np.int64))
    one = ids.ndim == 1
    ids2 = ids.reshape(1, -1) if one else ids
    rng = np.ascontiguousarray(np.asarray(season_ranges, dtype=np.int64).reshape(-1, 2))
    Y = rng.shape[0]
    out = np.zeros((ids2.shape[0], 4, Y), dtype=np.int64)
    hwa = np.zeros((ids2.shape[0], Y), dtype=np.float64)
    _lib.check(lib.hdp_season_metrics(_ptr(ids2), ids2.shape[0], ids2.shape[1], _ptr(rng), Y, _ptr(out),
                                      _ptr(hwa)))
    return (out[0], hwa[0]) if one else (out, hwa)


def heatwave_frequency(hw_ts, season_ranges):
    return _season_metrics(hw_ts, season_ranges)[0][..., 0, :]


def heatwave_number(hw_ts, season_ranges):
    return _season_metrics(hw_ts, season_ranges)[0][..., 1, :]


def heatwave_duration(hw_ts, season_ranges):
    return _season_metrics(hw_ts, season_ranges)[0][..., 2, :]


def heatwave_average(hw_ts, season_ranges):
    return _season_metrics(hw_ts, season_ranges)[1]


def indicate_hot_days(measure, threshold, doy_map):
    lib = _lib.ensure_device()
    x = np.ascontiguousarray(measure, dtype=np.float32)
    one = x.ndim == 1
    x2 = x.reshape(1, -1) if one else x
    thr = np.ascontiguousarray(threshold, dtype=np.float64).reshape(x2.shape[0], -1)
    dm = np.ascontiguousarray(doy_map, dtype=np.int64)
    hot = np.zeros(x2.shape, dtype=np.uint8)
    _lib.check(lib.hdp_indicate_hot_days(_ptr(x2), x2.shape[0], x2.shape[1], _ptr(thr), thr.shape[1],
                                         _ptr(dm), _ptr(hot)))
    hot = hot.astype(bool)
    return hot[0] if one else hot


def heat_index(temp, rel_humid):
    """NWS heat index, element-wise (the ufunc hdp.measure.heat_index, measure.py:61-94):
    temp [deg F] and rel_humid [%] float32 arrays of one shape -> float32 [deg F]."""
    lib = _lib.ensure_device()
    t = np.ascontiguousarray(temp, dtype=np.float32)
    r = np.ascontiguousarray(np.broadcast_to(np.asarray(rel_humid, dtype=np.float32), t.shape))
    out = np.empty(t.shape, dtype=np.float32)
    _lib.check(lib.hdp_heat_index_f32(_ptr(t), _ptr(r), t.size, _ptr(out)))
    return out


def weighted_row_mean(values, weights):
    """values [n_rows, n] (any real dtype; NaNs are skipped), weights [n] -> float64 [n_rows]:
    sum_c w[c] v[r][c] / sum_c w[c] over the valid values of row r -- the arithmetic of
    xarray's ``da.weighted(w).mean(...)`` as the reference's compute_weighted_spatial_mean uses it
    (hdp/graphics/figure.py:14-15)."""
    lib = _lib.ensure_device()
    v = np.ascontiguousarray(values, dtype=np.float64)
    if v.ndim != 2:
        raise ValueError("values must be [n_rows, n]")
    w = np.ascontiguousarray(weights, dtype=np.float64)
    if w.shape != (v.shape[1],):
        raise ValueError("weights must be [n]")
    out = np.empty(v.shape[0], dtype=np.float64)
    _lib.check(lib.hdp_weighted_mean_f64(_ptr(v), v.shape[0], v.shape[1], _ptr(w), _ptr(out)))
    return out


# ---- device-resident interface (bench.py, sharded runs) ----------------------------------

class DeviceArray:
    """A raw hipMalloc'd buffer with a numpy-style shape/dtype tag."""

    def __init__(self, shape, dtype):
        self.lib = _lib.ensure_device()
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self.ptr = self.lib.hdp_dev_alloc(self.nbytes)
        if not self.ptr:
            _lib.check(-4)

    @classmethod
    def from_host(cls, a):
        a = np.ascontiguousarray(a)
        d = cls(a.shape, a.dtype)
        _lib.check(d.lib.hdp_memcpy_h2d(d.ptr, _ptr(a), a.nbytes))
        return d

    def to_host(self):
        out = np.empty(self.shape, dtype=self.dtype)
        _lib.check(self.lib.hdp_memcpy_d2h(_ptr(out), self.ptr, self.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.lib.hdp_dev_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class ThresholdPlan:
    def __init__(self, time_index, cols, percentiles, T):
        self.lib = _lib.ensure_device()
        ti = np.ascontiguousarray(time_index, dtype=np.int64)
        cl = np.ascontiguousarray(cols, dtype=np.int32)
        q = np.ascontiguousarray(np.asarray(percentiles, dtype=np.float64).ravel())
        self.n_doy, self.S = ti.shape
        self.W, self.P, self.T = cl.shape[1], q.size, int(T)
        h = C.c_void_p()
        _lib.check(self.lib.hdp_threshold_plan_create(_ptr(ti), self.n_doy, self.S, _ptr(cl), self.W,
                                                      _ptr(q), self.P, self.T, C.byref(h)))
        self.handle = h

    def reserve(self, n_cells, time_major=False):
        """Allocate the plan's launch-time scratch up front (keeps run() free of synchronisation and hipMalloc)."""
        _lib.check(self.lib.hdp_threshold_plan_reserve(self.handle, int(n_cells), int(bool(time_major))))

    def run(self, x_ptr, n_cells, out_ptr, stream=None):
        """Device pointers; out is [n_cells][P][n_doy] float64 (percentile-major, what MetricsPlan.run reads)."""
        _lib.check(self.lib.hdp_thresholds_f32_dev(self.handle, x_ptr, int(n_cells), out_ptr, stream))

    def run_time_major(self, x_tm_ptr, pitch_cells, n_cells, out_ptr, stream=None):
        """x_tm is [T][pitch_cells] float32 on the device (CMIP order); output as run()."""
        _lib.check(self.lib.hdp_thresholds_f32_tm_dev(self.handle, x_tm_ptr, int(pitch_cells), int(n_cells), out_ptr,
                                                      stream))

    def describe(self):
        """The kernel a launch runs (plan choice under the HDP_THR_* switches at plan creation)."""
        return self.lib.hdp_threshold_plan_describe(self.handle).decode()

    def __del__(self):
        try:
            if self.handle:
                self.lib.hdp_threshold_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class MetricsPlan:
    def __init__(self, doy_map, n_doy, hw_definitions, north, south, P):
        self.lib = _lib.ensure_device()
        dm = np.ascontiguousarray(doy_map, dtype=np.int64)
        defs = np.ascontiguousarray(np.asarray(hw_definitions, dtype=np.int64).reshape(-1, 3))
        north = np.ascontiguousarray(north, dtype=np.int64).reshape(-1, 2)
        south = np.ascontiguousarray(south, dtype=np.int64).reshape(-1, 2)
        self.T, self.n_doy, self.D, self.Y, self.P = dm.size, int(n_doy), defs.shape[0], north.shape[0], int(P)
        h = C.c_void_p()
        _lib.check(self.lib.hdp_metrics_plan_create(_ptr(dm), self.T, self.n_doy, _ptr(defs), self.D,
                                                    _ptr(north), _ptr(south), self.Y, self.P, C.byref(h)))
        self.handle = h
        self.year_pitch = int(self.lib.hdp_metrics_year_pitch(h))

    def out_shape(self, n_cells):
        """Device layout of run()'s output: (metric, percentile, definition, year, series), int16."""
        return (4, self.P, self.D, self.year_pitch, int(n_cells))

    def reserve(self, n_cells):
        """Allocate the exceedance scratch up front (keeps run() free of allocations)."""
        _lib.check(self.lib.hdp_metrics_plan_reserve(self.handle, int(n_cells)))

    def batch_cells(self, n_cells):
        """Series per batch of the split path for a call of n_cells series."""
        return int(self.lib.hdp_metrics_plan_batch_cells(self.handle, int(n_cells)))

    def run(self, x_ptr, thr_ptr, n_thr_cells, is_south_ptr, n_cells, out_ptr, stream=None):
        """Device pointers; thr is [n_thr_cells][P][n_doy] float64 as written by ThresholdPlan.run."""
        _lib.check(self.lib.hdp_metrics_f32_dev(self.handle, x_ptr, thr_ptr, int(n_thr_cells), is_south_ptr,
                                                int(n_cells), out_ptr, stream))

    def run_time_major(self, x_tm_ptr, pitch_cells, thr_ptr, n_thr_cells, is_south_ptr, n_cells, out_ptr, stream=None):
        """x_tm is [T][pitch_cells] float32 on the device (CMIP order); output as run()."""
        _lib.check(self.lib.hdp_metrics_f32_tm_dev(self.handle, x_tm_ptr, int(pitch_cells), thr_ptr, int(n_thr_cells),
                                                   is_south_ptr, int(n_cells), out_ptr, stream))

    def describe(self):
        return self.lib.hdp_metrics_plan_describe(self.handle).decode()

    def __del__(self):
        try:
            if self.handle:
                self.lib.hdp_metrics_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass
