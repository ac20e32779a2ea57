"""Drop-in for ``hdp.metric`` (the hot half): same function names, arguments, output
variables, dims, coords, dtypes and attrs; the per-(cell, percentile, definition)
heatwave detection runs in the fused HIP kernel instead of Numba + apply_ufunc loops.

  compute_individual_metrics <- hdp/metric.py:372-506
  compute_group_metrics      <- hdp/metric.py:509-523
  compute_metrics_io         <- hdp/metric.py:526-590 (adds latitude-band streaming)
  index_heatwaves, heatwave_frequency/number/duration/average, indicate_hot_days
                             <- hdp/metric.py:11-172,280-301 (GPU mirrors, same signatures)
  get_range_indices, build_doy_map, compute_hemisphere_ranges
                             <- hdp/metric.py:175-277 (host tables)
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

from . import core
from . import dist as hdist
from . import io as hio
from ._xr import backend, jan1_stamps
from .calendar import build_doy_map, get_range_indices, hemisphere_season_tables  # noqa: F401
from .core import (heatwave_average, heatwave_duration, heatwave_frequency, heatwave_number,  # noqa: F401
                   index_heatwaves, indicate_hot_days)
from .utils import add_history, get_version


def compute_hemisphere_ranges(measure):
    """metric.py:212-262: int64 [year, end_points, lat, lon] season index table."""
    xr = backend()
    north, south, years = hemisphere_season_tables(np.asarray(measure.coords["time"].values))
    lat = np.asarray(measure.coords["lat"].values)
    lon = np.asarray(measure.coords["lon"].values)
    ranges = np.where((lat < 0)[None, None, :, None], south[:, :, None, None], north[:, :, None, None])
    ranges = np.broadcast_to(ranges, (north.shape[0], 2, lat.size, lon.size)).astype(np.int64)
    return xr.DataArray(ranges, dims=["year", "end_points", "lat", "lon"],
                        coords={"year": years, "end_points": ["start", "finish"], "lat": lat, "lon": lon})


def compute_heatwave_metrics(measure, threshold, doy_map, min_duration, max_break, max_subs, season_ranges):
    """metric.py:304-341 for ONE series: -> int64 [4, Y] rows HWF, HWN, HWD, HWA."""
    seasons = np.asarray(season_ranges, dtype=np.int64).reshape(-1, 2)
    out = core.compute_heatwave_metrics(np.asarray(measure, dtype=np.float32)[None, :],
                                        np.asarray(threshold, dtype=np.float64)[None, :, None], doy_map,
                                        [[min_duration, max_break, max_subs]], seasons, seasons,
                                        np.zeros(1, dtype=np.uint8))
    return out[0, 0, 0].astype(np.int64)


def compute_individual_metrics(measure, threshold, hw_definitions, include_threshold: bool = True,
                               check_variables: bool = True, shard=None):
    """HWF/HWN/HWD/HWA for one (measure, threshold) pair (metric.py:372-506).

    Output variables are int64 with dims (percentile, definition, <non-time dims of the
    measure in order>, time) where time holds one Jan-1 stamp per season year.

    ``shard`` (not in the reference, whose split over cells is the dask graph of :444-452): ``(rank, world)`` or
    ``"auto"`` -- this process computes the metrics of its contiguous range of grid cells only (all members of a cell
    stay on one GPU, with that cell's thresholds) and the ranks' results are all-gathered (hdp_amd.dist: RCCL over
    xGMI through the library's communicator), so every rank returns the complete Dataset."""
    blocks = hio.block_slices(measure, skip=("time", "member"))
    if blocks is not None and blocks[0] in threshold.dims:
        # lazily chunked measure: one block at a time with the matching slice of the thresholds (metric.py:444)
        dim, edges = blocks
        parts = [compute_individual_metrics(m_blk, t_blk, hw_definitions, include_threshold, check_variables, shard)
                 for m_blk, t_blk in hio.iter_bands((measure, threshold), edges, dim)]
        return hio.concat_dim(parts, dim)
    xr = backend()
    times = np.asarray(measure.coords["time"].values)
    if check_variables:
        assert "hdp_type" in threshold.attrs
        assert threshold.attrs["hdp_type"] == "threshold"
        assert threshold.attrs["baseline_variable"] == measure.attrs["baseline_variable"]
        assert threshold.attrs["baseline_calendar"] == times[0].calendar

    combined_history = ""
    for label, obj in (("Measure", measure), ("Threshold", threshold)):
        for entry in obj.attrs.get("history", "").split("\n"):
            if entry != "":
                combined_history += f"({label}) {entry}\n"

    north, south, years = hemisphere_season_tables(times)
    doy_map = build_doy_map(times)

    m_dims = list(measure.dims)
    other_dims = [d for d in m_dims if d != "time"]
    # members share their cell's thresholds: process member-major so series c uses row c % n_cells
    proc_dims = (["member"] if "member" in other_dims else []) + [d for d in other_dims if d != "member"]
    m_vals = np.asarray(measure.values)
    m_vals = np.moveaxis(m_vals, [m_dims.index(d) for d in proc_dims + ["time"]], range(len(m_dims)))
    proc_shape = m_vals.shape[:-1]
    x2d = m_vals.reshape(-1, m_vals.shape[-1])
    if x2d.dtype != np.float32:
        x2d = x2d.astype(np.float32)

    cell_dims = [d for d in proc_dims if d != "member"]
    t_dims = list(threshold.dims)
    thr = np.asarray(threshold.values, dtype=np.float64)
    thr = np.moveaxis(thr, [t_dims.index(d) for d in cell_dims + ["doy", "percentile"]], range(len(t_dims)))
    n_doy, P = thr.shape[-2], thr.shape[-1]
    thr3 = np.ascontiguousarray(thr.reshape(-1, n_doy, P))

    lat = np.asarray(measure.coords["lat"].values)
    shape_cells = [measure.shape[m_dims.index(d)] for d in proc_dims]
    lat_b = (lat < 0).reshape([-1 if d == "lat" else 1 for d in proc_dims])   # metric.py:249: lat == 0 is north
    is_south = np.broadcast_to(lat_b, shape_cells).reshape(-1).astype(np.uint8)

    # int64 planes [metric][P][D][series][Y] (int, as test_workflow.py:57), widened and regrouped on the device
    D, Y = len(hw_definitions), north.shape[0]
    if shard is None or hdist.check_shard(shard)[1] == 1:
        planes = core.compute_heatwave_metric_planes(x2d, thr3, doy_map, hw_definitions, north, south, is_south)
    else:
        # this rank's cells only (all members of a cell stay here, with the cell's thresholds); the collective moves
        # the int16 device layout -- never the widened int64 planes -- and the result is widened once, after it
        rank, world = hdist.check_shard(shard)
        n_thr = thr3.shape[0]
        n_mem = x2d.shape[0] // n_thr          # series are member-major: [member][cell]
        lo, hi = hdist.shard_bounds(n_thr, world, rank)
        x_loc = np.ascontiguousarray(x2d.reshape(n_mem, n_thr, x2d.shape[-1])[:, lo:hi]).reshape(n_mem * (hi - lo), x2d.shape[-1])
        south_loc = np.ascontiguousarray(is_south.reshape(n_mem, n_thr)[:, lo:hi]).reshape(-1)
        if hdist.comm_ready():     # library communicator: gather and widen on the device (RCCL over xGMI)
            planes, wire = core.compute_heatwave_metric_planes_sharded(x_loc, thr3[lo:hi], doy_map, hw_definitions, north,
                                                                       south, south_loc, n_mem, n_thr)
            hdist._note_wire(wire)
        else:                      # torch.distributed group (gloo in the CPU tests): int16 through the group
            err, lay = None, None
            try:
                lay = (core.compute_heatwave_metrics_layout(x_loc, thr3[lo:hi], doy_map, hw_definitions, north, south,
                                                            south_loc)
                       if hi > lo else np.zeros((4, P, D, Y, 0), dtype=np.int16))
            except Exception as e:        # noqa: BLE001 -- raised on every rank by agree()
                err = e
            try:
                hdist.agree(err is None, f"{type(err).__name__}: {err}" if err else "")
            except RuntimeError:
                if err is not None:
                    raise err
                raise
            planes = hdist.gather_metric_planes(lay, n_mem, n_thr, shard)
    planes = planes.reshape((4, P, D) + tuple(proc_shape) + (Y,))
    # back to the measure's own dim order (a view; only the member dim ever moves)
    src = ["metric", "percentile", "definition"] + proc_dims + ["year"]
    dst = ["metric", "percentile", "definition"] + other_dims + ["year"]
    planes = np.moveaxis(planes, [src.index(d) for d in dst], range(len(dst)))

    stamps = np.array(jan1_stamps(years, times[0]), dtype=object)
    coords = {k: np.asarray(measure.coords[k].values) for k in measure.coords if k != "time" and k in other_dims}
    coords["time"] = stamps
    coords["definition"] = [f"{d[0]}-{d[1]}-{d[2]}" for d in hw_definitions]
    coords["percentile"] = np.asarray(threshold.coords["percentile"].values)
    out_dims = ["percentile", "definition"] + other_dims + ["time"]
    ds = xr.Dataset(
        {name: xr.DataArray(planes[i], dims=out_dims, coords=coords)
         for i, name in enumerate(("HWF", "HWN", "HWD", "HWA"))})
    ds.attrs.update({
        "description": f"Heatwave metric dataset generated by Heatwave Diagnostics Package (HDP v{get_version()})",
        "hdp_version": get_version(),
        "hdp_type": "metric",
    })
    ds["HWF"].attrs.update({"units": "heatwave days", "long_name": "Heatwave Frequency",
                            "description": "Number of days that fall within heatwave during a heatwave season"})
    ds["HWD"].attrs.update({"units": "heatwave days", "long_name": "Heatwave Duration",
                            "description": "Length of longest heatwave during a heatwave season"})
    ds["HWN"].attrs.update({"units": "heatwave events", "long_name": "Heatwave Number",
                            "description": "Number of distinct heatwaves during a heatwave season"})
    ds["HWA"].attrs.update({"units": "heatwave events", "long_name": "Heatwave Average",
                            "description": "Average length of heatwaves during a heatwave season"})
    ds["percentile"].attrs.update({"range": "(0, 1)"})
    ds["definition"].attrs.update({
        "first_number": "Minimum number of consecutively hot days",
        "second_number": "Maximum number of break days after first wave",
        "third_number": "Minimum number of consecutively hot days after the break",
    })
    for variable in ds:
        ds[variable].attrs["history"] = combined_history
        add_history(ds[variable], f"Heatwave metrics generated by HDP v{get_version()}")
    return ds


def compute_group_metrics(measures, thresholds, hw_definitions, include_threshold: bool = False,
                          check_variables: bool = True, shard=None):
    """Every (measure, threshold) pair with matching ``baseline_variable``; variables are
    renamed ``{measure}.{threshold}.{metric}`` and merged (metric.py:509-523).
    ``shard``: see compute_individual_metrics."""
    xr = backend()
    metric_sets = []
    for measure_name in list(measures.keys()):
        measure = measures[measure_name]
        for threshold_name in list(thresholds.keys()):
            threshold = thresholds[threshold_name]
            if threshold.attrs["baseline_variable"] == measure.attrs["baseline_variable"]:
                hw = compute_individual_metrics(measure, threshold, hw_definitions, include_threshold,
                                                check_variables, shard)
                metric_sets.append(hw.rename({n: f"{measure_name}.{threshold_name}.{n}" for n in list(hw.keys())}))
    aggr = xr.merge(metric_sets)
    aggr.attrs["variable_naming_desc"] = "(heat measure).(threshold used).(heatwave metric)"
    aggr.attrs["variable_naming_delimeter"] = "."
    return aggr


def compute_metrics_io(output_path: str, measure_path: str, measure_var: str, threshold_path: str, hw_definitions: list,
                       include_threshold: bool = False, override_threshold_var: str = None, overwrite: bool = False,
                       lat_band: int = None) -> None:
    """Heatwave metrics from a measure file / store and a threshold file / store, written to
    ``output_path`` (``.zarr`` or ``.nc``) (metric.py:526-590: same arguments, checks and exceptions).

    The threshold variable is ``override_threshold_var`` when given (variables are then checked like
    ``compute_individual_metrics`` does); otherwise the documented default ``threshold_{measure_var}``
    or, when the dataset does not hold it, the name ``compute_threshold`` writes, ``{measure_var}_threshold``
    -- without the attribute checks, as in the reference (:558-560).  ``overwrite`` is what the
    reference's body refers to without declaring it; ``lat_band`` (not in the reference) streams the
    grid ``lat_band`` latitude rows at a time."""
    output_path = hio.prepare_output(output_path, overwrite)
    measure_ds = hio.open_dataset(Path(measure_path))
    threshold_ds = hio.open_dataset(Path(threshold_path))
    check_variables = override_threshold_var is not None
    if override_threshold_var is not None:
        threshold_var = override_threshold_var
    else:
        threshold_var = f"threshold_{measure_var}"
        if threshold_var not in threshold_ds and f"{measure_var}_threshold" in threshold_ds:
            threshold_var = f"{measure_var}_threshold"
    measure_data = measure_ds[measure_var]
    threshold_data = threshold_ds[threshold_var]
    if lat_band and "lat" in measure_data.dims:
        n_lat = measure_data.shape[list(measure_data.dims).index("lat")]
        parts = []
        for m_band, t_band in hio.iter_bands((measure_data, threshold_data), hio.lat_slices(n_lat, lat_band)):
            parts.append(compute_individual_metrics(m_band, t_band, hw_definitions,
                                                    include_threshold=include_threshold,
                                                    check_variables=check_variables))
        metric_ds = hio.concat_lat(parts)
    else:
        metric_ds = compute_individual_metrics(measure_data, threshold_data, hw_definitions,
                                               include_threshold=include_threshold, check_variables=check_variables)
    hio.write_dataset(metric_ds, output_path)
