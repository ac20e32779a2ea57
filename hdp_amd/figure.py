"""The reduction the reference's figure deck starts from, on the GPU:

  compute_weighted_spatial_mean <- hdp/graphics/figure.py:14-15
      ``da.weighted(np.cos(np.deg2rad(da.lat))).mean(dim=["lat", "lon"])``

Everything else in hdp.graphics (matplotlib / cartopy figures, notebooks) is out of scope.  xarray's weighted
mean is ``sum(w * x) / sum(w)`` over the non-NaN ``x``, the weights broadcast along ``lat``; the kernel
accumulates both sums in float64 in a fixed order (hdp_hip.h: hdp_weighted_mean_*).  xarray is not importable in
the build image, so parity with it is unpinned: the tests check a NumPy restatement of that formula to 1e-12.
"""
from __future__ import annotations

import numpy as np

from . import core
from ._xr import backend


def compute_weighted_spatial_mean(da):
    """Latitude-weighted mean over ``lat`` and ``lon``; the other dims keep their order and coords."""
    xr = backend()
    dims = list(da.dims)
    if "lat" not in dims or "lon" not in dims:
        raise ValueError("compute_weighted_spatial_mean needs 'lat' and 'lon' dimensions")
    keep = [d for d in dims if d not in ("lat", "lon")]
    values = np.asarray(da.values)
    moved = np.moveaxis(values, [dims.index(d) for d in keep + ["lat", "lon"]], range(len(dims)))
    n_lat, n_lon = moved.shape[-2], moved.shape[-1]
    keep_shape = moved.shape[:-2]
    lat = np.asarray(da.coords["lat"].values, dtype=np.float64)
    weights = np.repeat(np.cos(np.deg2rad(lat)), n_lon)            # broadcast along lon, row-major (lat, lon)
    rows = np.ascontiguousarray(moved.reshape(-1, n_lat * n_lon), dtype=np.float64)
    mean = core.weighted_row_mean(rows, weights).reshape(keep_shape)
    coords = {k: np.asarray(da.coords[k].values) for k in da.coords if k in keep}
    return xr.DataArray(mean, dims=keep, coords=coords, name=da.name, attrs=dict(da.attrs))
