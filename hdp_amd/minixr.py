"""A very small labelled-array container used when xarray is not installed.

The drop-in functions in ``hdp_amd.threshold`` / ``hdp_amd.metric`` take and return
xarray objects when xarray is importable.  The build container and the GPU test boxes
have no xarray (and no network to install it), so the same code paths are exercised
there with these stand-ins, which implement exactly the subset of the xarray API that
the adapters and the reference's own workflow test touch: ``dims``, ``shape``,
``dtype``, ``values``, ``coords``, ``attrs``, ``name``, ``rename``, ``mean``,
``compute``, item/attribute access on a Dataset, ``data_vars``, ``merge``, and -- for the latitude-band
streaming of the ``*_io`` wrappers -- ``isel`` with slices and ``concat`` along an existing dimension.
It does no file I/O (``open_dataset`` / ``to_netcdf`` / ``to_zarr`` need the real xarray).
"""
from __future__ import annotations

import numpy as np


class DataArray:
    chunks = None   # xarray: per-dimension block lengths of a dask-backed array; tests may set it on an instance

    def __init__(self, data, dims=None, coords=None, name=None, attrs=None):
        self.values = np.asarray(data)
        if dims is None:
            dims = tuple(coords.keys()) if coords is not None and self.values.ndim else ()
        self.dims = tuple(dims)
        if len(self.dims) != self.values.ndim:
            raise ValueError(f"dims {self.dims} do not match array of rank {self.values.ndim}")
        self.coords = {}
        for k, v in (coords or {}).items():
            if isinstance(v, DataArray):
                v = v.values
            elif isinstance(v, tuple) and len(v) == 2:   # (dims, values) form
                v = v[1]
            self.coords[k] = _Coord(k, v)
        self.name = name
        self.attrs = dict(attrs or {})

    # -- array protocol -------------------------------------------------------------
    shape = property(lambda self: self.values.shape)
    dtype = property(lambda self: self.values.dtype)
    size = property(lambda self: self.values.size)
    ndim = property(lambda self: self.values.ndim)

    def __array__(self, dtype=None, copy=None):
        return self.values if dtype is None else self.values.astype(dtype)

    def __getattr__(self, item):
        coords = self.__dict__.get("coords", {})
        if item in coords:
            return coords[item]
        raise AttributeError(item)

    def __getitem__(self, key):
        if isinstance(key, str):
            return self.coords[key]
        raise TypeError("positional indexing is not part of the minixr subset")

    def compute(self):
        return self

    def isel(self, **indexers):
        """Slices only (what the band streaming needs); coordinates of the sliced dims follow."""
        index = [slice(None)] * self.values.ndim
        coords = self._coord_dict()
        for dim, sl in indexers.items():
            if not isinstance(sl, slice):
                raise TypeError("minixr.isel takes slices")
            index[self.dims.index(dim)] = sl
            if dim in coords:
                coords[dim] = coords[dim][sl]
        return DataArray(self.values[tuple(index)], self.dims, coords, self.name, self.attrs)

    def astype(self, dtype):
        return DataArray(self.values.astype(dtype), self.dims, self._coord_dict(), self.name, self.attrs)

    def rename(self, name):
        return DataArray(self.values, self.dims, self._coord_dict(), name, self.attrs)

    def mean(self, *a, **k):
        return float(np.mean(self.values))

    def _coord_dict(self):
        return {k: v.values for k, v in self.coords.items()}

    def __float__(self):
        return float(self.values)

    def __repr__(self):
        return f"<minixr.DataArray {self.name!r} {dict(zip(self.dims, self.shape))} {self.dtype}>"


class _Coord(DataArray):
    def __init__(self, name, values):
        self.values = np.asarray(values)
        self.dims = (name,) if self.values.ndim == 1 else ()
        self.coords = {}
        self.name = name
        self.attrs = {}


class Dataset:
    def __init__(self, data_vars=None, coords=None, attrs=None):
        self.data_vars = {}
        self.coords = {}
        self.attrs = dict(attrs or {})
        for k, v in (coords or {}).items():
            if isinstance(v, DataArray):
                v = v.values
            elif isinstance(v, tuple) and len(v) == 2:
                v = v[1]
            self.coords[k] = _Coord(k, v)
        for k, v in (data_vars or {}).items():
            self[k] = v

    def __setitem__(self, key, value):
        if not isinstance(value, DataArray):
            raise TypeError("minixr.Dataset holds DataArrays")
        da = DataArray(value.values, value.dims, value._coord_dict(), key, value.attrs)
        for ck, cv in da.coords.items():
            if ck not in self.coords:
                self.coords[ck] = _Coord(ck, cv.values)
                self.coords[ck].attrs = dict(cv.attrs)
        self.data_vars[key] = da

    def __getitem__(self, key):
        if key in self.data_vars:
            return self.data_vars[key]
        return self.coords[key]

    def __getattr__(self, item):
        d = self.__dict__
        if item in d.get("data_vars", {}):
            return d["data_vars"][item]
        if item in d.get("coords", {}):
            return d["coords"][item]
        raise AttributeError(item)

    def __iter__(self):
        return iter(self.data_vars)

    def __contains__(self, key):
        return key in self.data_vars or key in self.coords

    def keys(self):
        return self.data_vars.keys()

    def __len__(self):
        return len(self.data_vars)

    def compute(self):
        return self

    def rename(self, mapping):
        out = Dataset(coords={k: v.values for k, v in self.coords.items()}, attrs=self.attrs)
        for ck, cv in self.coords.items():
            out.coords[ck].attrs = dict(cv.attrs)
        for k, v in self.data_vars.items():
            out[mapping.get(k, k)] = v
        return out

    def mean(self, *a, **k):
        return {name: float(np.mean(v.values)) for name, v in self.data_vars.items()}

    def __repr__(self):
        return f"<minixr.Dataset vars={list(self.data_vars)} coords={list(self.coords)}>"


def merge(datasets):
    out = Dataset()
    for ds in datasets:
        for ck, cv in ds.coords.items():
            if ck in out.coords:
                if out.coords[ck].values.shape != cv.values.shape or not np.all(out.coords[ck].values == cv.values):
                    raise ValueError(f"conflicting values for coordinate {ck!r}")
            else:
                out.coords[ck] = _Coord(ck, cv.values)
                out.coords[ck].attrs = dict(cv.attrs)
        for k, v in ds.data_vars.items():
            out[k] = v
        for ak, av in ds.attrs.items():   # combine_attrs="override"-like: first one wins
            out.attrs.setdefault(ak, av)
    return out


def concat(datasets, dim):
    """Datasets along an existing dimension: variables (and the coordinate) that carry ``dim`` are
    concatenated in order, everything else and all attrs come from the first Dataset."""
    first = datasets[0]
    coords = {k: v.values for k, v in first.coords.items()}
    if dim in coords:
        coords[dim] = np.concatenate([np.asarray(ds.coords[dim].values) for ds in datasets])
    out = Dataset(coords=coords, attrs=first.attrs)
    for ck, cv in first.coords.items():
        out.coords[ck].attrs = dict(cv.attrs)
    for name, da in first.data_vars.items():
        if dim in da.dims:
            axis = da.dims.index(dim)
            values = np.concatenate([ds.data_vars[name].values for ds in datasets], axis=axis)
            c = da._coord_dict()
            c[dim] = coords[dim]
            out[name] = DataArray(values, da.dims, c, name, da.attrs)
        else:
            out[name] = da
    return out
