"""Round-3 GPU tests (through the C ABI, checked against the C oracle):

* the int16 device-layout result and the SHARDED metrics call (int16 all-gather on the device, widened once on the
  gathered buffer) behind the library communicator -- a world of one on this box, members sharing thresholds;
* import order: hdp_amd first, torch afterwards, must leave torch with its GPU (one HIP runtime per process);
* BASELINE config 5 beyond a handful of cells: 4 096 cells x 10 members (S = 1000 samples per day of year, 20
  percentiles x 12 definitions), properties over every cell and C-oracle equality on 64 of them.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from hdp_amd import _lib, calendar as cal, core, utils  # noqa: E402
from hdp_amd import dist as hdist  # noqa: E402
from oracle import c_oracle  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _small_case(seed=3, n_cells=37, members=3, years=6, P=4):
    rng = np.random.default_rng(seed)
    dates = utils.noleap_date_range("2000-01-01", f"{2000 + years - 1}-12-31")
    T = dates.size
    x = rng.normal(0, 1, size=(members, n_cells, T)).astype(np.float32)
    x += (np.arange(T, dtype=np.float32) / np.float32(T))[None, None, :]
    thr = np.sort(rng.normal(0.8, 0.3, size=(n_cells, 365, P)), axis=2)
    dm = cal.build_doy_map(dates)
    north, south, _ = cal.hemisphere_season_tables(dates)
    defs = [[3, 0, 0], [2, 1, 1], [4, 2, 1]]
    hemi = np.tile((np.arange(n_cells) % 2).astype(np.uint8), members)
    return x.reshape(members * n_cells, T), thr, dm, defs, north, south, hemi, members, n_cells


def test_device_layout_result_is_the_block_result_transposed():
    x, thr, dm, defs, north, south, hemi, members, n_cells = _small_case()
    blocks = core.compute_heatwave_metrics(x, thr, dm, defs, north, south, hemi)          # [P, D, n, 4, Y]
    layout = core.compute_heatwave_metrics_layout(x, thr, dm, defs, north, south, hemi)   # [4, P, D, Y, n]
    assert layout.dtype == np.int16 and layout.shape == (4,) + blocks.shape[:2] + (blocks.shape[4], blocks.shape[2])
    assert np.array_equal(np.transpose(layout, (1, 2, 4, 0, 3)), blocks)
    want = c_oracle.metrics(x, np.concatenate([thr] * members), dm, defs, north, south, hemi)
    assert np.array_equal(blocks.astype(np.int64), want)


def test_sharded_planes_call_gathers_int16_on_the_device():
    """hdp_metrics_f32_planes_i64_sharded with a communicator of one rank: the rank's cells are the whole grid, the
    all-gather runs (ncclAllGather on the int16 layout), the widening kernel regroups the gathered buffer; the bytes
    handed to the collective are the int16 layout's, a quarter of the int64 planes'."""
    x, thr, dm, defs, north, south, hemi, members, n_cells = _small_case(seed=5)
    planes = core.compute_heatwave_metric_planes(x, thr, dm, defs, north, south, hemi)    # [4, P, D, n, Y] int64
    with pytest.raises(_lib.HdpError):     # no communicator: refused, not silently unsharded
        core.compute_heatwave_metric_planes_sharded(x, thr, dm, defs, north, south, hemi, members, n_cells)
    hdist.comm_init_rank(hdist.comm_unique_id(), 0, 1)
    try:
        got, wire = core.compute_heatwave_metric_planes_sharded(x, thr, dm, defs, north, south, hemi, members, n_cells)
        assert got.dtype == np.int64 and np.array_equal(got, planes)
        assert wire == planes.size * 2 and wire * 4 == planes.nbytes
        with pytest.raises(_lib.HdpError):  # a cell count that is not this rank's range of the grid
            core.compute_heatwave_metric_planes_sharded(x, thr, dm, defs, north, south, hemi, members, n_cells + 1)
        # the adapter rides on it: shard="auto" with the library communicator
        import hdp_amd.metric
        import hdp_amd.threshold
        from tests.helpers import measure_dataset
        base, lon, lat, bdates = utils.generate_control_array(start_date="1700-01-01", end_date="1703-12-31", add_noise=True)
        warm, _, _, mdates = utils.generate_warming_array(start_date="2000-01-01", end_date="2003-12-31", add_noise=True)
        thr_ds = hdp_amd.threshold.compute_thresholds(measure_dataset(base, lon, lat, bdates), [0.9, 0.95])
        m1 = hdp_amd.metric.compute_group_metrics(measure_dataset(warm, lon, lat, mdates), thr_ds, defs)
        # world of one short-circuits to the unsharded call; force the sharded branch through the core call instead
        ms = hdp_amd.metric.compute_group_metrics(measure_dataset(warm, lon, lat, mdates), thr_ds, defs, shard="auto")
        for v in m1.data_vars:
            assert np.array_equal(ms[v].values, m1[v].values) and ms[v].dims == m1[v].dims
    finally:
        hdist.comm_destroy()


@pytest.mark.timeout(600)
def test_torch_keeps_its_gpu_when_hdp_amd_is_imported_first():
    """libhdp_hip.so loaded BEFORE torch (the order a user's script may well have): both must share one HIP runtime
    (hdp_amd._lib._share_hip_runtime), so torch still sees the device and both can use it in one process."""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from hdp_amd import _lib\n"
        "lib = _lib.ensure_device(0)\n"
        "p = lib.hdp_dev_alloc(1 << 20); assert p\n"
        "import torch\n"
        "assert torch.cuda.is_available(), 'torch lost the GPU: ' + str(_lib.hip_runtime)\n"
        "t = torch.arange(1024, device='cuda:0').sum().item(); assert t == 1023 * 512\n"
        "lib.hdp_dev_free(p)\n"
        "print('ok', _lib.hip_runtime)\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=540)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:]


@pytest.mark.timeout(900)
def test_c5_ensemble_4096_cells_properties_and_oracle_sample():
    """Config 5's per-cell shape on 4 096 cells x 10 members: thresholds from the members concatenated along time
    (threshold.py:114-119: S = 1000, windows of 15 000 samples), metrics of the 40 960 member series against their
    cell's thresholds; device-generated series, both hemispheres."""
    import torch
    lib = _lib.ensure_device()
    dev = torch.device("cuda", 0)
    ts = torch.cuda.Stream(dev)
    torch.cuda.set_stream(ts)
    stream = ts.cuda_stream
    years, M, n = 100, 10, 4096
    PERC = np.linspace(0.80, 0.99, 20)
    DEFS = [[d, b, b] for d in (3, 4, 5, 6) for b in (0, 1, 2)]
    dates = utils.noleap_date_range("2000-01-01", f"{2000 + years - 1}-12-31")
    T = dates.size
    ti, cols = cal.window_columns(np.concatenate([dates] * M), 7)
    assert ti.shape == (365, 1000)
    dm = cal.build_doy_map(dates)
    north, south, _ = cal.hemisphere_season_tables(dates)
    P, D, Y, n_doy = PERC.size, len(DEFS), north.shape[0], 365
    lat = np.linspace(-90.0, 90.0, n).astype(np.float32)
    lat_dev = torch.from_numpy(lat).to(dev)
    south_dev = torch.from_numpy(np.tile((lat < 0).astype(np.uint8), M)).to(dev)
    lat_m = torch.from_numpy(np.tile(lat, M)).to(dev)
    xb = torch.empty(n * M * T, dtype=torch.float32, device=dev)           # [cell][M * T]
    xm = torch.empty(M * n * T, dtype=torch.float32, device=dev)           # [member][cell][T]
    _lib.check(lib.hdp_generate_series_dev(xb.data_ptr(), n, M * T, 11, lat_dev.data_ptr(), 0, 0.7, 0.0, stream))
    _lib.check(lib.hdp_generate_series_dev(xm.data_ptr(), M * n, T, 11 * M, lat_m.data_ptr(), 1, 0.7, 1.0 / 36500.0, stream))
    tplan = core.ThresholdPlan(ti, cols, PERC, M * T)
    mplan = core.MetricsPlan(dm, n_doy, DEFS, north, south, P)
    thr = torch.empty((n, P, n_doy), dtype=torch.float64, device=dev)
    out = torch.empty((4, P, D, Y, M * n), dtype=torch.int16, device=dev)
    tplan.run(xb.data_ptr(), n, thr.data_ptr(), stream)
    mplan.run(xm.data_ptr(), thr.data_ptr(), n, south_dev.data_ptr(), M * n, out.data_ptr(), stream)
    torch.cuda.synchronize(dev)

    # ---- properties over every cell (on the device)
    assert not bool(torch.isnan(thr).any())
    assert bool((thr[:, 1:, :] >= thr[:, :-1, :]).all())                    # quantiles are monotone in q
    xb2 = xb.view(n, M * T)
    assert bool((thr.amax(dim=(1, 2)) <= xb2.amax(dim=1).double()).all())
    assert bool((thr.amin(dim=(1, 2)) >= xb2.amin(dim=1).double()).all())
    hwf, hwn, hwd, hwa = (out[i].to(torch.int32) for i in range(4))
    assert bool((hwf >= 0).all()) and bool((hwd <= hwf).all()) and bool((hwn <= hwf).all())
    assert bool((hwa == torch.where(hwn > 0, hwf // hwn.clamp(min=1), torch.zeros_like(hwf))).all())
    # a higher percentile never has more heatwave days under a definition without breaks (b == 0: indices 0, 3, 6, 9)
    assert bool((hwf[1:, ::3] <= hwf[:-1, ::3]).all())
    assert int(hwf.sum()) > 0

    # ---- C-oracle equality on 64 cells strided over both hemispheres (all their members)
    idx = np.unique(np.linspace(0, n - 1, 64).astype(np.int64))
    it = torch.from_numpy(idx).to(dev)
    xs_b = xb2[it].cpu().numpy()
    xs_m = xm.view(M, n, T)[:, it].cpu().numpy().reshape(M * idx.size, T)
    th_g = thr[it].cpu().numpy().transpose(0, 2, 1)                         # [cells, n_doy, P]
    rows = [0, 1, 100, 182, 357, 358, 364]                                  # the oracle sorts 15 000 samples per row
    win = cal.expand_window_table(ti, cols)[rows]
    th_c = c_oracle.thresholds(xs_b, win, PERC)
    assert np.array_equal(th_g[:, rows], th_c, equal_nan=True)
    og = out.view(4, P, D, Y, M, n)[..., it].cpu().numpy()                  # [4, P, D, Y, M, 64]
    met_g = np.transpose(og, (1, 2, 4, 5, 0, 3)).reshape(P, D, M * idx.size, 4, Y).astype(np.int64)
    hemi = np.tile((lat[idx] < 0).astype(np.uint8), M)
    met_c = c_oracle.metrics(xs_m, np.concatenate([th_g] * M), dm, DEFS, north, south, hemi)
    assert np.array_equal(met_g, met_c)


# ---- the lock-step state machines on the cases written for the round-3 run-queue prototype (the prototype itself was
# measured slower -- profiles/r03_metrics_queue* -- and is no longer in the tree) -------------------------------------
from tests.test_round2_gpu import _regular_case  # noqa: E402


@pytest.mark.parametrize("n_doy,T,P,defs", [
    (365, 365 * 9, 10, [[3, 0, 0], [3, 1, 1], [4, 0, 0], [4, 1, 1], [5, 0, 0], [5, 1, 1]]),
    (365, 365 * 40 + 17, 10, [[3, 0, 0], [3, 1, 1], [5, 2, 2]]),
    (365, 365 * 8 + 63, 2, [[25, 1, 1], [30, 0, 0]]),
    (360, 360 * 7 + 200, 4, [[3, 0, 0], [40, 2, 1]]),
    (365, 365 * 26, 4, [[3, 0, 0], [4, 0, 0], [5, 0, 0], [6, 0, 0]]),      # every pair simple: nothing is ever "near"
    (365, 365 * 12, 3, [[1, 0, 0], [0, 3, 1], [2, 5, 0], [1, 1, 3]]),      # min_duration 0/1: every run is kept; long breaks
    (365, 365 * 30, 2, [[3, 40, 2], [6, 2, 0], [2, 0, 0]]),               # a 40-day max_break keeps far-apart short runs
])
def test_state_machines_match_the_oracle_on_year_aligned_and_day_aligned_words(n_doy, T, P, defs, monkeypatch):
    monkeypatch.setenv("HDP_METRICS_YEARS", "2")
    case = _regular_case(7000 + n_doy + P + len(defs), n_doy, T, 70, P, defs, long_runs=True)
    x, thr, doy_map, dfs, north, south, is_south = case
    want = c_oracle.metrics(x, thr, doy_map, dfs, north, south, is_south)
    lock = core.compute_heatwave_metrics(*case)
    assert np.array_equal(lock.astype(np.int64), want)
    monkeypatch.setenv("HDP_METRICS_YEARS", "0")      # day-aligned words (any calendar) through the same kernel
    assert np.array_equal(core.compute_heatwave_metrics(*case), lock)


@pytest.mark.parametrize("seed", range(10))
def test_state_machines_random_sweep_dense_exceedance(seed, monkeypatch):
    """Dense exceedance (many runs inside a year and inside a word), random definitions incl. large max_break."""
    rng = np.random.default_rng(12000 + seed)
    n_doy = int(rng.choice([365, 365, 360, 366, int(rng.integers(321, 385))]))
    years = int(rng.choice([3, 9, 26, 40]))
    T = min(65535, years * n_doy + int(rng.choice([0, 1, 63, 64, 200])))
    P = int(rng.integers(1, 12))
    D = int(rng.integers(1, 9))
    defs = [[int(rng.integers(0, 7)), 0 if rng.random() < 0.4 else int(rng.integers(1, 9)), int(rng.integers(0, 4))]
            for _ in range(D)]
    case = list(_regular_case(12100 + seed, n_doy, T, 67, P, defs, long_runs=bool(seed & 1)))
    case[1] = np.sort(rng.normal(-0.2 + 0.5 * rng.random(), 0.5, size=case[1].shape), axis=2)   # hot about half the time
    x, thr, doy_map, dfs, north, south, is_south = case
    if north.shape[0] == 0:
        pytest.skip("no complete season in this record")
    want = c_oracle.metrics(x, thr, doy_map, dfs, north, south, is_south)
    monkeypatch.setenv("HDP_METRICS_YEARS", "2")
    got = core.compute_heatwave_metrics(*case)
    assert np.array_equal(got.astype(np.int64), want), (n_doy, T, P, defs)


def test_threshold_plan_reserve_then_launch():
    """hdp_threshold_plan_reserve: the tiered image's tail (and, time-major, the staging) exist before the first launch; the
    launches that follow give the results of a plan that allocated on first use."""
    import torch
    dev = torch.device("cuda", 0)
    lib = _lib.ensure_device()
    dates = utils.noleap_date_range("2000-01-01", "2099-12-31")
    ti, cols = cal.window_columns(dates, 7)
    q = np.arange(0.9, 1.0, 0.01)
    T, n = dates.size, 700
    lat = torch.linspace(-60, 60, n, device=dev)
    x = torch.empty(n * T, dtype=torch.float32, device=dev)
    _lib.check(lib.hdp_generate_series_dev(x.data_ptr(), n, T, 3, lat.data_ptr(), 5, 0.7, 0.0, None))
    a = core.ThresholdPlan(ti, cols, q, T)
    b = core.ThresholdPlan(ti, cols, q, T)
    b.reserve(n)
    b.reserve(n, time_major=True)
    out_a = torch.empty(n * q.size * 365, dtype=torch.float64, device=dev)
    out_b = torch.empty_like(out_a)
    out_c = torch.empty_like(out_a)
    a.run(x.data_ptr(), n, out_a.data_ptr())
    b.run(x.data_ptr(), n, out_b.data_ptr())
    xt = x.view(n, T).t().contiguous()
    b.run_time_major(xt.data_ptr(), n, n, out_c.data_ptr())
    torch.cuda.synchronize(dev)
    _lib.check(lib.hdp_sync(None))
    assert bool(torch.equal(out_a.view(torch.int64), out_b.view(torch.int64)))
    assert bool(torch.equal(out_a.view(torch.int64), out_c.view(torch.int64)))
