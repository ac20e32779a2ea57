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


def _adapter_worker(rank, world, port, q):
    """The product adapters with shard=(rank, world): every rank computes its cells (oracle in place of the GPU calls,
    as in tests/test_adapter_cpu.py) and must end up with the complete, unsharded Datasets."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    import hdp_amd.metric
    import hdp_amd.threshold
    from hdp_amd import calendar as cal
    from hdp_amd import core, utils
    from oracle import hdp_oracle as orc
    from tests.helpers import measure_dataset

    calls = []

    def fake_percentiles(x, time_index, cols, qq):
        calls.append(x.shape[0])
        return orc.compute_thresholds_cells(np.ascontiguousarray(x), cal.expand_window_table(time_index, cols), qq)

    def fake_planes(x, thr, doy_map, defs, north, south, is_south):
        x = np.ascontiguousarray(x)
        full = thr[np.arange(x.shape[0]) % thr.shape[0]]
        m = orc.compute_metrics_cells(x, full, doy_map, defs, north, south, is_south)
        return np.ascontiguousarray(np.moveaxis(m.astype(np.int64), 3, 0))

    def fake_layout(x, thr, doy_map, defs, north, south, is_south):   # int16 device layout [4, P, D, Y, n]
        calls_layout.append(x.shape[0])
        x = np.ascontiguousarray(x)
        full = thr[np.arange(x.shape[0]) % thr.shape[0]]
        m = orc.compute_metrics_cells(x, full, doy_map, defs, north, south, is_south)   # [P, D, n, 4, Y]
        return np.ascontiguousarray(np.transpose(m, (3, 0, 1, 4, 2))).astype(np.int16)

    calls_layout = []
    core.compute_percentiles = fake_percentiles
    core.compute_heatwave_metric_planes = fake_planes
    core.compute_heatwave_metrics_layout = fake_layout
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        base, lon, lat, bdates = utils.generate_control_array(start_date="1700-01-01", end_date="1703-12-31", add_noise=True)
        warm, _, _, mdates = utils.generate_warming_array(start_date="2000-01-01", end_date="2003-12-31", add_noise=True)
        qv = [0.9, 0.95]
        defs = [[3, 0, 0], [2, 1, 1]]
        thr_s = hdp_amd.threshold.compute_thresholds(measure_dataset(base, lon, lat, bdates), qv, shard=(rank, world))
        n_local = list(calls)
        met_s = hdp_amd.metric.compute_group_metrics(measure_dataset(warm, lon, lat, mdates), thr_s, defs, shard="auto")
        wire = hd.last_wire_bytes       # bytes this rank handed to the transport for the metrics: the int16 layout
        thr_1 = hdp_amd.threshold.compute_thresholds(measure_dataset(base, lon, lat, bdates), qv)
        met_1 = hdp_amd.metric.compute_group_metrics(measure_dataset(warm, lon, lat, mdates), thr_1, defs)
        same_thr = bool(np.array_equal(thr_s["temp_threshold"].values, thr_1["temp_threshold"].values))
        same_met = all(np.array_equal(met_s[v].values, met_1[v].values) and met_s[v].dims == met_1[v].dims
                       for v in met_1.data_vars)
        n_years = met_1["temp.temp_threshold.HWF"].shape[-1]
        want_wire = 4 * len(qv) * len(defs) * n_years * hd.shard_size(6, world) * 2
        q.put((rank, n_local, same_thr, bool(same_met), wire == want_wire, list(calls_layout)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_adapters_return_the_complete_datasets_on_every_rank():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_adapter_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # 6 grid cells (lon 2 x lat 3): 3 per rank
    assert got[0][1] == [3] and got[1][1] == [3]
    assert all(g[2] and g[3] for g in got)
    # the collective carried int16: 2 bytes per (metric, percentile, definition, year, cell of the padded shard)
    assert all(g[4] for g in got)


@pytest.mark.timeout(300)
def test_sharded_adapters_with_a_rank_that_owns_no_cells():
    """6 cells on 4 ranks: shard_bounds gives (0,2) (2,4) (4,6) (6,6) -- the last rank computes nothing, must still take
    part in every collective, and must still return the complete Datasets."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_adapter_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=240) for _ in range(4))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [g[1] for g in got] == [[2], [2], [2], []]          # thresholds: the empty rank never calls the kernel
    assert [g[5] for g in got] == [[2], [2], [2], []]          # metrics likewise
    assert all(g[2] and g[3] and g[4] for g in got)


def _failing_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def fn(lo, hi):
            if rank == 1:
                raise MemoryError("rank 1 ran out of memory")
            return np.ones((hi - lo, 3))
        try:
            hd.sharded_over_cells(fn, 7, 0, (rank, world))
            q.put((rank, "returned"))
        except MemoryError as e:
            q.put((rank, f"MemoryError: {e}"))
        except RuntimeError as e:
            q.put((rank, f"RuntimeError: {e}"))
        # a shard= that disagrees with the transport is refused before anything is computed
        try:
            hd.sharded_over_cells(lambda lo, hi: np.ones((hi - lo, 3)), 7, 0, (rank, world + 1))
            q.put((rank, "returned"))
        except ValueError:
            q.put((rank, "ValueError"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_an_exception_on_one_rank_is_raised_on_every_rank():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(4))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert (0, "RuntimeError: sharded call failed on rank(s) [1]") in got
    assert (1, "MemoryError: rank 1 ran out of memory") in got
    assert got.count((0, "ValueError")) == 1 and got.count((1, "ValueError")) == 1


def test_comm_entry_points_fail_loudly_without_a_device():
    """The RCCL communicator lives behind the C ABI (hdp_comm_*, hdp_allgather_dev): exported, bound, and -- like every
    compute entry point -- HDP_ENODEV without a usable HIP device."""
    from hdp_amd import _lib
    lib = _lib.load()
    assert lib.hdp_comm_world() == 0 and lib.hdp_comm_rank() == -1
    if lib.hdp_device_count() > 0:
        pytest.skip("a HIP device is visible; the no-device behaviour cannot be observed here")
    import ctypes as C
    buf = C.create_string_buffer(hd.COMM_ID_BYTES)
    assert lib.hdp_comm_unique_id(buf) == -2
    assert lib.hdp_comm_init_rank(buf, 0, 1) == -2
    assert lib.hdp_allgather_dev(None, 8, None, None) == -2
    assert lib.hdp_allgather_direct_dev(None, 8, None, None) == -2
    assert hd.current() == (0, 1)
