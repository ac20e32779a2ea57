"""N > 1 path on CPU: two gloo ranks shard the grid cells, compute their shard (oracle as the
per-shard compute so no GPU is needed), all-gather, and must reproduce the unsharded result."""
import os
import socket

import numpy as np
import pytest

from hdp_amd import dist as hd


def test_shard_bounds_cover_exactly():
    for n in (1, 2, 7, 8, 64800, 1036800):
        for w in (1, 2, 3, 4, 8):
            b = [hd.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert all(e - s <= hd.shard_size(n, w) for s, e in b)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from oracle import hdp_oracle as orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)                     # same data on every rank
        dates = orc.noleap_date_range("2001-01-01", "2003-12-31")
        n_cells = 7                                        # odd: the last shard is ragged
        x = rng.normal(size=(n_cells, dates.size)).astype(np.float32)
        thr_q = [0.9, 0.95]
        defs = [[3, 0, 0], [2, 1, 1]]
        win = orc.datetimes_to_windows(dates, 7)
        north, south, _ = orc.hemisphere_ranges(dates)
        dm = orc.build_doy_map(dates)
        hemi = (np.arange(n_cells) % 2).astype(np.uint8)
        s, e = hd.shard_bounds(n_cells, world, rank)
        thr_local = orc.compute_thresholds_cells(x[s:e], win, thr_q) if e > s else np.zeros((0, 365, 2))
        met_local = (orc.compute_metrics_cells(x[s:e], thr_local, dm, defs, north, south, hemi[s:e])
                     if e > s else np.zeros((2, 2, 0, 4, north.shape[0]), dtype=np.int64)).astype(np.int16)
        pad = hd.shard_size(n_cells, world)
        thr_all = hd.allgather_cells(hd.pad_cells(thr_local, pad, 0), n_cells, 0)
        met_all = hd.allgather_cells(hd.pad_cells(met_local, pad, 2), n_cells, 2)
        if rank == 0:
            want_thr = orc.compute_thresholds_cells(x, win, thr_q)
            want = orc.compute_metrics_cells(x, want_thr, dm, defs, north, south, hemi)
            q.put((bool(np.array_equal(thr_all, want_thr)), bool(np.array_equal(met_all.astype(np.int64), want))))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_metrics_match_unsharded():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok_thr, ok_met = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ok_thr and ok_met
