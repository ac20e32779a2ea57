"""Grid-cell sharding across the GPUs of a node (one process per GPU).

Every series is independent in both passes (reference docs/testing.rst:21; gufunc core
dims threshold.py:57, metric.py:364), so rank r owns the contiguous cell range
[r*ceil(n/W), (r+1)*ceil(n/W)) and no collective runs during compute.  The only exchange
is the all-gather that reassembles the (int16) metrics, and optionally the thresholds,
on every rank -- RCCL over xGMI when the tensors live on the GPU (torch.distributed
backend "nccl"), gloo in the CPU tests.
"""
from __future__ import annotations

import numpy as np


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


def allgather_cells(local, n_cells: int, axis: int, group=None, device=None):
    """All-gather per-rank results along the cell axis and strip the padding.

    `local` is this rank's array (numpy, or a torch tensor already on the right device)
    whose `axis` has shard_size(n_cells, world) entries (pad with pad_cells).  Returns the
    reassembled array with n_cells entries along `axis`, same type as the input."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    is_np = isinstance(local, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(local)) if is_np else local.contiguous()
    if device is not None:
        t = t.to(device)
    # cell axis first so that the gathered buffer is a plain concatenation of shards
    t = t.movedim(axis, 0).contiguous()
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    # bytes on the wire: neither RCCL nor gloo has an int16 datatype
    dist.all_gather_into_tensor(out.view(torch.uint8).reshape(-1), t.view(torch.uint8).reshape(-1), group=group)
    out = out[:n_cells].movedim(0, axis)
    return out.cpu().numpy() if is_np else out
