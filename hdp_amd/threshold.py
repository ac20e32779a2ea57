"""Drop-in for ``hdp.threshold`` (the hot half): same function names, arguments,
output variables, dims, coords, dtypes and attrs; the per-cell quantile loop runs in
the HIP kernels of libhdp_hip.so instead of the Numba gufunc.

  compute_threshold   <- hdp/threshold.py:96-204
  compute_thresholds  <- hdp/threshold.py:207-229
  compute_threshold_io<- hdp/threshold.py:232-289  (adds latitude-band streaming)
  datetimes_to_windows<- hdp/threshold.py:12-49   (re-exported from hdp_amd.calendar)

Differences, by design: results are eager (numpy-backed) where the reference returns a
lazy dask graph -- ``.compute()`` is a no-op on them; numpy-backed inputs are accepted
(the reference requires dask-backed input, threshold.py:144).  A dask-backed input is walked
block by block along its first chunked non-time dimension, so only one
block is materialised at a time.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

from . import core
from . import dist as hdist
from . import io as hio
from ._xr import backend
from .calendar import datetimes_to_windows, window_columns  # noqa: F401
from .utils import add_history, get_version


def _series_matrix(values, dims, time_dim="time"):
    """[..., time, ...] -> ([n_series, T] view when the layout allows, non-time dims, shape)."""
    t_axis = dims.index(time_dim)
    moved = np.moveaxis(values, t_axis, -1)
    other_dims = [d for d in dims if d != time_dim]
    other_shape = moved.shape[:-1]
    return moved.reshape(-1, moved.shape[-1]), other_dims, other_shape


def compute_threshold(baseline_data, percentiles, no_season: bool = False, rolling_window_size: int = 7,
                      fixed_value: float = None, shard=None):
    """Percentile thresholds per grid cell and day of year (threshold.py:96-204).

    ``rolling_window_size`` is a radius: the window is ``2*rolling_window_size + 1`` days
    (threshold.py:41).  ``no_season`` and ``fixed_value`` are recorded in attrs only, as
    in the reference (:182-184).

    ``shard`` (not in the reference, whose split over cells is the dask graph of :161-169): ``(rank, world)`` or
    ``"auto"`` -- this process computes only its contiguous range of grid cells on its GPU and the thresholds of
    all ranks are all-gathered (hdp_amd.dist), so every rank returns the complete Dataset."""
    blocks = hio.block_slices(baseline_data, skip=("time", "member"))
    if blocks is not None:   # lazily chunked input: one block at a time, as the reference's map_blocks does
        dim, edges = blocks
        # block k + 1 is read (dask: computed) on a helper thread while block k is on the GPU
        parts = [compute_threshold(blk, percentiles, no_season, rolling_window_size, fixed_value, shard)
                 for (blk,) in hio.iter_bands((baseline_data,), edges, dim)]
        return hio.concat_dim(parts, dim)
    xr = backend()
    dims = list(baseline_data.dims)
    values = np.asarray(baseline_data.values)
    times = np.asarray(baseline_data.coords["time"].values)
    if "member" in dims:
        # threshold.py:114-119: every member's series is appended along time
        m_axis, t_axis = dims.index("member"), dims.index("time")
        values = np.moveaxis(values, (m_axis, t_axis), (-2, -1))
        n_member = values.shape[-2]
        values = values.reshape(values.shape[:-2] + (n_member * values.shape[-1],))
        dims = [d for d in dims if d not in ("member", "time")] + ["time"]
        times = np.concatenate([times] * n_member)
    values = values.astype(np.float32, copy=False)       # threshold.py:121
    percentiles = np.array(percentiles)

    time_index, cols = window_columns(times, rolling_window_size)
    x2d, other_dims, other_shape = _series_matrix(values, dims)
    if shard is None:
        thr = core.compute_percentiles(x2d, time_index, cols, percentiles)
    else:
        thr = hdist.sharded_over_cells(lambda lo, hi: core.compute_percentiles(x2d[lo:hi], time_index, cols, percentiles),
                                       x2d.shape[0], 0, shard)
    n_doy = time_index.shape[0]
    thr = thr.reshape(tuple(other_shape) + (n_doy, percentiles.size))

    coords = {k: np.asarray(baseline_data.coords[k].values) for k in baseline_data.coords
              if k not in ("time", "member") and k in other_dims}
    coords["doy"] = np.arange(n_doy)
    coords["percentile"] = percentiles
    attrs = dict(baseline_data.attrs)
    da = xr.DataArray(thr, dims=other_dims + ["doy", "percentile"], coords=coords, attrs=attrs)
    add_history(da, f"Threshold data computed by HDP v{get_version()}.\n")
    if "long_name" in da.attrs:
        add_history(da, f"Metadata updated: 'long_name' value '{da.attrs['long_name']}' overwritten by HDP.\n")
    da.attrs.update({
        "long_name": f"Percentile threshold values for baseline variable '{baseline_data.name}'",
        "baseline_variable": baseline_data.name,
        "baseline_start_time": f"{str(times[0])}",
        "baseline_end_time": f"{str(times[-1])}",
        "baseline_calendar": f"{str(times[-1].calendar)}",
        "param_percentiles": str(percentiles),
        "param_noseason": str(no_season),
        "param_rolling_window_size": str(rolling_window_size),
        "param_fixed_value": str(fixed_value),
        "hdp_type": "threshold",
    })
    ds = xr.Dataset(
        data_vars={f"{baseline_data.name}_threshold": da},
        coords=dict(lon=(["lon"], coords["lon"]), lat=(["lat"], coords["lat"]), doy=np.arange(0, n_doy),
                    percentile=percentiles),
        attrs=dict(
            description=f"Extreme heat threshold dataset generated by Heatwave Diagnostics Package (HDP v{get_version()})",
            hdp_version=get_version(),
        ),
    )
    ds["doy"].attrs = dict(units="day_of_year", baseline_calendar=str(times[0].calendar))
    return ds


def compute_thresholds(baseline_dataset, percentiles, no_season: bool = False, rolling_window_size: int = 7,
                       fixed_value: float = None, shard=None):
    """One threshold variable per data variable of the Dataset, merged (threshold.py:207-229).
    ``shard``: see compute_threshold."""
    xr = backend()
    parts = [compute_threshold(baseline_dataset[name], percentiles, no_season, rolling_window_size, fixed_value, shard)
             for name in baseline_dataset]
    return xr.merge(parts)


def compute_threshold_io(baseline_path: str, baseline_var: str, output_path: str, percentiles, no_season: bool = False,
                         rolling_window_size: int = 7, fixed_value: float = None, overwrite: bool = False,
                         lat_band: int = None) -> None:
    """Thresholds from a netCDF file / zarr store, written to ``output_path`` (``.zarr`` or ``.nc``)
    (threshold.py:232-289: same arguments, checks and exceptions; the opened variable gets a
    ``baseline_source`` attr).

    ``lat_band`` (not in the reference) streams the grid through the GPU ``lat_band`` latitude rows
    at a time, so a variable larger than host memory or HBM can be processed; the result is the same."""
    output_path = hio.prepare_output(output_path, overwrite)
    baseline_path = Path(baseline_path)
    baseline_data = hio.open_dataset(baseline_path)[baseline_var]
    baseline_data.attrs["baseline_source"] = str(baseline_path)
    if lat_band and "lat" in baseline_data.dims:
        n_lat = baseline_data.shape[list(baseline_data.dims).index("lat")]
        parts = []
        for (band,) in hio.iter_bands((baseline_data,), hio.lat_slices(n_lat, lat_band)):
            parts.append(compute_threshold(band, percentiles, no_season, rolling_window_size, fixed_value))
        threshold_ds = hio.concat_lat(parts)
    else:
        threshold_ds = compute_threshold(baseline_data, percentiles, no_season, rolling_window_size, fixed_value)
    hio.write_dataset(threshold_ds, output_path)
