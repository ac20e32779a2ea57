"""Grid-cell sharding across the GPUs of a node (one process per GPU).

Every series is independent in both passes (reference docs/testing.rst:21; gufunc core
dims threshold.py:57, metric.py:364), so rank r owns the contiguous cell range
[r*ceil(n/W), (r+1)*ceil(n/W)) and no collective runs during compute -- the split the
reference expresses as dask chunks (threshold.py:161-169, metric.py:444-452).  The only
exchange is the all-gather that reassembles the (int16/int64) metrics, and the thresholds,
on every rank.

Transport, in order of preference:
  * the library's own RCCL communicator (``hdp_comm_*`` / ``hdp_allgather_dev`` in
    include/hdp_hip.h: ncclAllGather over xGMI, no torch involved) once ``comm_init_rank``
    or ``init_from_env`` has run;
  * ``torch.distributed`` if the caller initialised a process group (backend "nccl" is RCCL
    on ROCm; "gloo" in the CPU tests);
  * none for a single process.
"""
from __future__ import annotations

import ctypes as C
import os
import socket
import time

import numpy as np

from . import _lib

COMM_ID_BYTES = 128   # HDP_COMM_ID_BYTES (sizeof(ncclUniqueId))


def shard_size(n_cells: int, world: int) -> int:
    return (int(n_cells) + world - 1) // world


def shard_bounds(n_cells: int, world: int, rank: int):
    """[start, stop) of `rank`'s cells; trailing ranks may own fewer (or zero) cells."""
    s = shard_size(n_cells, world)
    return min(n_cells, rank * s), min(n_cells, (rank + 1) * s)


def pad_cells(local: np.ndarray, n_pad: int, axis: int) -> np.ndarray:
    """Zero-pad the cell axis to the common shard size (all-gather needs equal shards)."""
    if local.shape[axis] == n_pad:
        return local
    widths = [(0, 0)] * local.ndim
    widths[axis] = (0, n_pad - local.shape[axis])
    return np.pad(local, widths)


# ---- the library's RCCL communicator (C ABI; no torch) ---------------------------------------

def comm_unique_id() -> bytes:
    """Rank 0: the 128 bytes every other rank needs for comm_init_rank (ship them by any channel)."""
    lib = _lib.ensure_device()
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _lib.check(lib.hdp_comm_unique_id(buf))
    return buf.raw


def comm_init_rank(ident: bytes, rank: int, world: int) -> None:
    """Collective: every rank calls this with rank 0's id (ncclCommInitRank)."""
    lib = _lib.ensure_device()
    if len(ident) != COMM_ID_BYTES:
        raise ValueError(f"communicator id must be {COMM_ID_BYTES} bytes")
    _lib.check(lib.hdp_comm_init_rank(C.create_string_buffer(bytes(ident), COMM_ID_BYTES), int(rank), int(world)))


def comm_destroy() -> None:
    _lib.load().hdp_comm_destroy()


def comm_world() -> int:
    return int(_lib.load().hdp_comm_world())


def comm_rank() -> int:
    return int(_lib.load().hdp_comm_rank())


def comm_ready() -> bool:
    return comm_world() > 0


def init_from_env(port_offset: int = 29, timeout: float = 120.0):
    """Torch-free start-up under any launcher that sets RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR /
    MASTER_PORT (torchrun does): rank 0 serves the RCCL unique id on MASTER_PORT + port_offset, the other
    ranks fetch it, then every rank joins the communicator.  Returns (rank, world)."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    _lib.ensure_device(int(os.environ.get("LOCAL_RANK", "0")))
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("MASTER_PORT", "29500")) + port_offset
    if rank == 0:
        ident = comm_unique_id()
        if world > 1:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(world)
            srv.settimeout(timeout)
            for _ in range(world - 1):
                conn, _peer = srv.accept()
                conn.sendall(ident)
                conn.close()
            srv.close()
    else:
        deadline = time.time() + timeout
        ident = b""
        while True:
            try:
                with socket.create_connection((addr, port), timeout=5.0) as c:
                    while len(ident) < COMM_ID_BYTES:
                        chunk = c.recv(COMM_ID_BYTES - len(ident))
                        if not chunk:
                            break
                        ident += chunk
                if len(ident) == COMM_ID_BYTES:
                    break
                ident = b""
            except OSError:
                pass
            if time.time() > deadline:
                raise TimeoutError(f"rank {rank}: no communicator id from {addr}:{port}")
            time.sleep(0.2)
    comm_init_rank(ident, rank, world)
    return rank, world


# bytes this rank handed to the transport in its most recent gather (tests assert the int16 volume on the wire)
last_wire_bytes = 0


def _note_wire(nbytes: int) -> None:
    global last_wire_bytes
    last_wire_bytes = int(nbytes)


def _allgather_bytes_rccl(local: np.ndarray) -> np.ndarray:
    """local: C-contiguous array, equal size on every rank -> [world, *local.shape] through hdp_allgather_dev."""
    lib = _lib.ensure_device()
    world = comm_world()
    nbytes = local.nbytes
    out = np.empty((world,) + local.shape, dtype=local.dtype)
    _note_wire(nbytes)
    if nbytes == 0:
        return out
    send, recv = lib.hdp_dev_alloc(nbytes), lib.hdp_dev_alloc(nbytes * world)
    if not send or not recv:
        _lib.check(-4)
    try:
        _lib.check(lib.hdp_memcpy_h2d(send, local.ctypes.data_as(C.c_void_p), nbytes))
        _lib.check(lib.hdp_allgather_dev(send, nbytes, recv, None))
        _lib.check(lib.hdp_memcpy_d2h(out.ctypes.data_as(C.c_void_p), recv, nbytes * world))
    finally:
        lib.hdp_dev_free(send)
        lib.hdp_dev_free(recv)
    return out


def _torch_group():
    try:
        import torch.distributed as tdist
        if tdist.is_available() and tdist.is_initialized():
            return tdist
    except ImportError:
        pass
    return None


def current(shard=None):
    """(rank, world) of this process: an explicit ``shard=(rank, world)``, else the library communicator, else an
    initialised torch.distributed group, else (0, 1)."""
    if shard is not None and shard != "auto":
        rank, world = shard
        return int(rank), int(world)
    return transport()


def transport():
    """(rank, world) of the transport a gather would use: the library communicator, else torch's group, else (0, 1)."""
    if comm_ready():
        return comm_rank(), comm_world()
    tdist = _torch_group()
    if tdist is not None:
        return tdist.get_rank(), tdist.get_world_size()
    return 0, 1


def check_shard(shard):
    """(rank, world) for a ``shard=`` argument, checked against the transport that will carry the gather: shard sizes and
    padding come from the caller's (rank, world), the collective from the communicator -- if the two disagree the result
    would be silently mis-shaped, so that is an error here."""
    rank, world = current(shard)
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad shard {shard!r}")
    if world > 1:
        t_rank, t_world = transport()
        if (t_rank, t_world) != (rank, world):
            raise ValueError(f"shard=({rank}, {world}) but the initialised transport is rank {t_rank} of {t_world}: "
                             "initialise hdp_amd.dist (comm_init_rank / init_from_env) or torch.distributed with the "
                             "same ranks before calling a sharded adapter")
    return rank, world


def allgather_cells(local, n_cells: int, axis: int, group=None, device=None, world=None):
    """All-gather per-rank results along the cell axis and strip the padding.

    `local` is this rank's array (numpy, or a torch tensor already on the right device)
    whose `axis` has shard_size(n_cells, world) entries (pad with pad_cells).  Returns the
    reassembled array with n_cells entries along `axis`, same type as the input."""
    is_np = isinstance(local, np.ndarray)
    if is_np and comm_ready() and group is None:
        t = np.ascontiguousarray(np.moveaxis(local, axis, 0))          # cell axis first: shards concatenate
        g = _allgather_bytes_rccl(t)                                   # [world, shard, ...]
        g = g.reshape((-1,) + t.shape[1:])[:n_cells]
        return np.moveaxis(g, 0, axis)
    if world == 1:
        return local
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    t = torch.from_numpy(np.ascontiguousarray(local)) if is_np else local.contiguous()
    if device is not None:
        t = t.to(device)
    # cell axis first so that the gathered buffer is a plain concatenation of shards
    t = t.movedim(axis, 0).contiguous()
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    # bytes on the wire: neither RCCL nor gloo has an int16 datatype
    _note_wire(t.numel() * t.element_size())
    dist.all_gather_into_tensor(out.view(torch.uint8).reshape(-1), t.view(torch.uint8).reshape(-1), group=group)
    out = out[:n_cells].movedim(0, axis)
    return out.cpu().numpy() if is_np else out


def agree(ok: bool, what: str = "") -> None:
    """Collective: raise on EVERY rank if any rank reports a failure (one byte per rank through the transport), so a
    rank whose local computation raised never leaves the others blocked in the data collective."""
    _, world = transport()
    if world == 1:
        if not ok:
            raise RuntimeError(what or "local computation failed")
        return
    flag = np.array([0 if ok else 1], dtype=np.uint8)
    if comm_ready():
        flags = _allgather_bytes_rccl(flag).reshape(-1)
    else:
        import torch
        tdist = _torch_group()
        out = torch.empty(world, dtype=torch.uint8)
        tdist.all_gather_into_tensor(out, torch.from_numpy(flag))
        flags = out.numpy()
    bad = [int(r) for r in np.nonzero(flags)[0]]
    if bad:
        raise RuntimeError(f"sharded call failed on rank(s) {bad}" + (f": {what}" if what and not ok else ""))


def sharded_over_cells(fn, n_cells: int, axis: int, shard, empty=None):
    """Run ``fn(lo, hi)`` on this rank's cell range and all-gather the results along ``axis`` (the output's cell
    axis): what the ``shard=`` argument of the hdp_amd.threshold / hdp_amd.metric adapters does.

    A rank that owns no cells (``shard_bounds`` gives trailing ranks an empty range when the grid is small) does not
    call ``fn``: it contributes ``empty(0)`` -- an array with a zero-length cell axis (default: ``fn``'s result shape is
    learned from the other ranks being unnecessary, the padding supplies it) -- and still takes part in the collectives.
    Every rank then agrees on success BEFORE the data collective, so an exception on one rank is raised on all."""
    rank, world = check_shard(shard)
    if world == 1 or n_cells == 0:   # (an empty grid: nothing to shard, every rank takes fn's own empty result)
        return fn(0, n_cells)
    lo, hi = shard_bounds(n_cells, world, rank)
    local, err = None, None
    try:
        if hi > lo:
            local = np.asarray(fn(lo, hi))
        elif empty is not None:
            local = np.asarray(empty(0))
    except Exception as e:        # noqa: BLE001 -- re-raised below, on every rank
        err = e
    try:
        agree(err is None, f"{type(err).__name__}: {err}" if err else "")
    except RuntimeError:
        if err is not None:
            raise err
        raise
    # shape and dtype of a shard: from this rank's result, or -- where the rank owns nothing and no `empty` was
    # given -- from a rank that does (rank 0 always owns cells when n_cells > 0)
    meta = _share_meta(local, axis)
    if local is None:
        shape = list(meta[0])
        shape[axis] = 0
        local = np.zeros(shape, dtype=meta[1])
    local = pad_cells(local, shard_size(n_cells, world), axis)
    return allgather_cells(local, n_cells, axis, world=world)


def _share_meta(local, axis):
    """(shape, dtype) of rank 0's local result, broadcast as a small fixed-size record (rank 0 owns cells whenever the
    grid has any)."""
    rec = np.zeros(16, dtype=np.int64)
    if local is not None:
        rec[0] = local.ndim
        rec[1:1 + local.ndim] = local.shape
        rec[15] = np.dtype(local.dtype).num
    if comm_ready():
        allr = _allgather_bytes_rccl(rec)
    else:
        import torch
        tdist = _torch_group()
        _, world = transport()
        out = torch.empty(world * 16, dtype=torch.int64)
        tdist.all_gather_into_tensor(out, torch.from_numpy(rec))
        allr = out.numpy().reshape(world, 16)
    r0 = allr[0]
    nd = int(r0[0])
    dt = [np.dtype(t) for t in (np.int16, np.int32, np.int64, np.float32, np.float64, np.uint8) if np.dtype(t).num == int(r0[15])]
    return tuple(int(v) for v in r0[1:1 + nd]), (dt[0] if dt else np.dtype(np.float64))


def gather_metric_planes(layout_local: np.ndarray, n_mem: int, n_cells: int, shard) -> np.ndarray:
    """int16 device-layout result of this rank's cells, [4, P, D, Y, n_mem * n_loc] (series member-major), -> the int64
    planes of the whole grid [4, P, D, n_mem, n_cells, Y] on every rank.  The collective moves the int16 values (2 bytes
    each; widening first would put four times the bytes on the wire); widening and the (year, series) regrouping happen
    once, on the gathered array.  Used with a torch.distributed group; with the library communicator the same happens
    on the device (core.compute_heatwave_metric_planes_sharded)."""
    rank, world = check_shard(shard)
    lo, hi = shard_bounds(n_cells, world, rank)
    n_loc = hi - lo
    four, P, D, Y = layout_local.shape[:4]
    loc = layout_local.reshape(four, P, D, Y, n_mem, n_loc)
    loc = pad_cells(loc, shard_size(n_cells, world), 5)
    full = allgather_cells(np.ascontiguousarray(loc), n_cells, 5, world=world)      # int16 [4, P, D, Y, n_mem, n_cells]
    return np.ascontiguousarray(np.moveaxis(full, 3, 5)).astype(np.int64)          # [4, P, D, n_mem, n_cells, Y]
