"""Path-in / path-out plumbing shared by ``compute_threshold_io`` (hdp/threshold.py:232-289) and
``compute_metrics_io`` (hdp/metric.py:526-590): output-path checks with the reference's exceptions,
opening a variable from a netCDF file or zarr store, and latitude-band streaming.

The reference opens the whole variable lazily (dask) and lets the scheduler walk its chunks; here the
same effect comes from slicing the lazily opened variable into bands of latitude rows, each of which
is materialised, pushed through the HIP kernels and released before the next -- grid cells are
independent, so bands need no halo.  Band k + 1 is read on a helper thread while band k computes (``iter_bands``), and
inside a band the library uploads chunk c + 1 under the kernels of chunk c (AsyncUpload in hdp_api.hip).  Reading and writing need xarray (with netCDF4 / zarr); neither
is in the build image, so the flow is tested with an in-memory stand-in (tests/test_io_cpu.py).
"""
from __future__ import annotations

import os
from pathlib import Path

from ._xr import backend

SUPPORTED_SUFFIXES = (".zarr", ".nc")


def prepare_output(output_path, overwrite: bool) -> Path:
    """The reference's checks, in its order (threshold.py:266-277): an existing output or a missing
    parent directory without ``overwrite`` -> FileExistsError; unknown suffix -> ValueError.
    With ``overwrite`` the missing parent directory is created."""
    output_path = Path(output_path)
    if output_path.exists() and not overwrite:
        raise FileExistsError(f"Overwrite parameter set to False and file exists at '{output_path}'.")
    if not output_path.parent.exists():
        if overwrite:
            os.makedirs(output_path.parent, exist_ok=True)
        else:
            raise FileExistsError(
                f"Overwrite parameter set to False and directory '{output_path.parent}' does not exist.")
    if output_path.suffix not in SUPPORTED_SUFFIXES:
        raise ValueError(f"File type '{output_path.suffix}' from '{output_path}' not supported.")
    return output_path


def _require(xr, name):
    fn = getattr(xr, name, None)
    if fn is None:
        raise ImportError(f"{name} needs xarray (with netCDF4 or zarr); it is not importable here and "
                          f"hdp_amd's stand-in container does no file I/O")
    return fn


def open_dataset(path):
    """zarr store (a ``.zarr`` directory) or anything ``xarray.open_dataset`` reads (threshold.py:279-282)."""
    xr = backend()
    path = Path(path)
    if path.suffix == ".zarr" and path.is_dir():
        return _require(xr, "open_zarr")(path)
    return _require(xr, "open_dataset")(path)


def write_dataset(ds, output_path: Path):
    """``.zarr`` -> to_zarr, ``.nc`` -> to_netcdf (threshold.py:286-289)."""
    if output_path.suffix == ".zarr":
        ds.to_zarr(output_path)
    else:
        ds.to_netcdf(output_path)


def lat_slices(n_lat: int, lat_band):
    """[a, b) row ranges covering the latitude axis; one range when ``lat_band`` is None or too large."""
    if not lat_band or lat_band >= n_lat:
        return [(0, n_lat)]
    lat_band = int(lat_band)
    if lat_band < 1:
        raise ValueError("lat_band must be a positive number of latitude rows")
    return [(a, min(a + lat_band, n_lat)) for a in range(0, n_lat, lat_band)]


def iter_bands(variables, slices, dim="lat"):
    """Yield, for every [a, b) of ``slices``, the tuple of ``v.isel(dim=slice(a, b))`` for v in ``variables`` -- with
    band k + 1 being read from its file / store (``.load()`` of a lazily opened variable) on a helper thread while the
    caller pushes band k through the GPU.  ctypes releases the GIL for the library calls, so the read of the next band,
    the upload of the next chunk (the library's own upload thread) and the kernels of the current chunk all overlap;
    at most two bands are in host memory at a time."""
    import threading

    def fetch(a, b, box):
        try:
            band = tuple(v.isel(**{dim: slice(a, b)}) for v in variables)
            box.append(tuple(x.load() if hasattr(x, "load") else x for x in band))
        except BaseException as e:      # noqa: BLE001 -- re-raised by the consumer
            box.append(e)

    slices = list(slices)
    nxt, th = [], None
    if slices:
        fetch(*slices[0], nxt)
    for k in range(len(slices)):
        cur = nxt[0]
        if isinstance(cur, BaseException):
            raise cur
        nxt = []
        if k + 1 < len(slices):
            th = threading.Thread(target=fetch, args=(*slices[k + 1], nxt), daemon=True)
            th.start()
        yield cur
        if th is not None:
            th.join()
            th = None


def concat_lat(parts):
    """Datasets of consecutive latitude bands -> one Dataset (attrs of the first band)."""
    if len(parts) == 1:
        return parts[0]
    return backend().concat(parts, dim="lat")


def block_slices(da, skip=("time",)):
    """(dim, [(a, b), ...]) for the first dimension (other than ``skip``) along which ``da`` is stored in more than
    one chunk -- a dask-backed xarray object exposes ``.chunks`` as per-dimension tuples of block lengths -- else
    None.  The reference hands such blocks to ``map_blocks`` (threshold.py:138-161, metric.py:420-444); the adapters
    walk them in order, so only one block of a lazily loaded variable is in host memory at a time."""
    chunks = getattr(da, "chunks", None)
    if not chunks:
        return None
    for dim, lens in zip(da.dims, chunks):
        if dim in skip or lens is None or len(lens) < 2:
            continue
        edges, a = [], 0
        for n in lens:
            edges.append((a, a + int(n)))
            a += int(n)
        return dim, edges
    return None


def concat_dim(parts, dim):
    """Datasets of consecutive blocks along ``dim`` -> one Dataset (attrs of the first block)."""
    if len(parts) == 1:
        return parts[0]
    return backend().concat(parts, dim=dim)
