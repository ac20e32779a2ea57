"""Round-4 GPU tests (through the C ABI):

* the multi-rank layout of the sharded metrics call, pinned on one GPU: a gathered buffer built on the host from the
  per-rank device layouts of a 3-rank split (a grid that does not divide by the world, three members) goes through
  the regrouping kernel and must equal the unsharded planes;
* a rank that fails before the data collective (allocation, its share of the grid, its own preparation) returns an
  error through the status exchange and leaves the communicator usable.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from hdp_amd import _lib, core  # noqa: E402
from hdp_amd import dist as hdist  # noqa: E402
from tests.test_round3_gpu import _small_case  # noqa: E402


@pytest.mark.parametrize("world,n_cells,members", [(3, 7, 3), (4, 6, 2), (2, 37, 1), (5, 3, 2)])
def test_regroup_of_a_multi_rank_gathered_buffer(world, n_cells, members):
    x, thr, dm, defs, north, south, hemi, members, n_cells = _small_case(seed=11, n_cells=n_cells, members=members)
    T = x.shape[1]
    want = core.compute_heatwave_metric_planes(x, thr, dm, defs, north, south, hemi)      # [4, P, D, members * n, Y]
    P, D, Y = want.shape[1], want.shape[2], want.shape[4]
    shard = hdist.shard_size(n_cells, world)
    gathered = np.zeros((world, 4, P, D, Y, members * shard), dtype=np.int16)
    x3, h2 = x.reshape(members, n_cells, T), hemi.reshape(members, n_cells)
    for r in range(world):
        lo, hi = hdist.shard_bounds(n_cells, world, r)
        if hi == lo:
            continue          # a rank without cells contributes zeros (small grids: the last ranks)
        lay = core.compute_heatwave_metrics_layout(np.ascontiguousarray(x3[:, lo:hi]).reshape(-1, T), thr[lo:hi], dm, defs,
                                                   north, south, np.ascontiguousarray(h2[:, lo:hi]).reshape(-1))
        for m in range(members):   # member m's cells at columns m * shard + c of the rank's padded shard buffer
            gathered[r, ..., m * shard: m * shard + (hi - lo)] = lay[..., m * (hi - lo): (m + 1) * (hi - lo)]
    got = core.regroup_gathered_planes(gathered, world, members, n_cells)
    assert got.dtype == np.int64 and got.shape == want.shape
    assert np.array_equal(got, want)
    with pytest.raises(ValueError):
        core.regroup_gathered_planes(gathered[:, :, :, :, :, :-1], world, members, n_cells)


def test_a_failing_rank_reports_through_the_status_exchange():
    x, thr, dm, defs, north, south, hemi, members, n_cells = _small_case(seed=7)
    planes = core.compute_heatwave_metric_planes(x, thr, dm, defs, north, south, hemi)
    hdist.comm_init_rank(hdist.comm_unique_id(), 0, 1)
    try:
        os.environ["HDP_FAULT_INJECT"] = "sharded_alloc"      # the gathered buffer's allocation "fails"
        try:
            with pytest.raises(_lib.HdpError) as ei:
                core.compute_heatwave_metric_planes_sharded(x, thr, dm, defs, north, south, hemi, members, n_cells)
            assert ei.value.code == -4 and "gathered" in str(ei.value)
        finally:
            del os.environ["HDP_FAULT_INJECT"]
        with pytest.raises(_lib.HdpError) as ei:               # not this rank's share of the grid
            core.compute_heatwave_metric_planes_sharded(x, thr, dm, defs, north, south, hemi, members, n_cells + 1)
        assert "owns" in str(ei.value)
        with pytest.raises(ValueError):                        # the rank's own preparation fails (tables do not fit)
            core.compute_heatwave_metric_planes_sharded(x, thr, dm[:-1], defs, north, south, hemi, members, n_cells)
        # the communicator and the entry point are still good afterwards
        got, wire = core.compute_heatwave_metric_planes_sharded(x, thr, dm, defs, north, south, hemi, members, n_cells)
        assert np.array_equal(got, planes) and wire == planes.size * 2
    finally:
        hdist.comm_destroy()


def test_rccl_version_and_mapped_runtime_are_reported():
    info = _lib.runtime_report()
    assert info["rccl_version"] > 20000
    for key in ("librccl", "libamdhip64", "libhsa-runtime64"):
        assert info["mapped"][key], info


# ---- thresholds: quantile SETS anywhere in [0, 1] (the former tools/dbg/fuzz_thresholds.py) -----------------------------
@pytest.mark.parametrize("seed", range(48))
def test_threshold_plans_fuzz_quantile_sets(seed, monkeypatch):
    """Random record lengths, window radii and quantile sets (tails, straddling the median, spread over the window, both
    tails, duplicates, 0 and 1), ties and special values -- the plan's own choice, its blocked form with the walks cut into
    runs that enter the merge at a pivot (HDP_THR_DUAL=1 forces the runs), and the classic blocked form, against the C
    oracle, bit for bit."""
    from hdp_amd import calendar as cal
    from oracle import c_oracle, hdp_oracle as orc
    rng = np.random.default_rng(50000 + seed)
    years = int(rng.choice([3, 9, 17, 40, 64, 65, 80, 100, 100, 100, 128]))
    dates = orc.noleap_date_range("0001-01-01", f"{years:04d}-12-31")
    T = dates.size
    radius = int(rng.choice([0, 1, 3, 7, 7, 7, 8]))
    kind = int(rng.integers(0, 6))
    P = int(rng.integers(1, 13))
    if kind == 0:
        q = np.sort(rng.uniform(0.85, 1.0, P))
    elif kind == 1:
        q = np.sort(rng.uniform(0.0, 0.15, P))
    elif kind == 2:
        q = np.sort(rng.uniform(0.4, 0.6, P))
    elif kind == 3:
        q = np.sort(rng.uniform(0.0, 1.0, P))
    elif kind == 4:
        q = np.sort(np.concatenate([rng.uniform(0.0, 0.1, P // 2 + 1), rng.uniform(0.9, 1.0, P // 2 + 1)]))
    else:
        q = np.sort(rng.choice([0.0, 0.5, 1.0, 0.25, 0.75, 0.5000001, 0.4999999], P))
    ncell = int(rng.integers(1, 6))
    x = (15 + 6 * np.sin(2 * np.pi * np.arange(T) / 365.0)[None, :] + rng.normal(0, 2, size=(ncell, T))).astype(np.float32)
    if rng.random() < 0.3:
        x = np.round(x)                          # many exact ties
    if rng.random() < 0.2:
        x[0, rng.integers(0, T, 3)] = np.inf
    if rng.random() < 0.2:
        x[0, rng.integers(0, T, 3)] = -np.inf
    if rng.random() < 0.1:
        x[-1, rng.integers(0, T)] = np.nan
    if rng.random() < 0.15:
        x[ncell // 2] = np.float32(3.25)         # a constant series: every sample ties with the pivot
    ti, cols = cal.window_columns(dates, radius)
    win = cal.expand_window_table(ti, cols)
    with np.errstate(invalid="ignore"):
        want = c_oracle.thresholds(x, win, q)

    def same(a, b):
        return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))

    for env in ({}, {"HDP_THR_WHOLE": "0"}, {"HDP_THR_WHOLE": "0", "HDP_THR_DUAL": "1"}, {"HDP_THR_WHOLE": "0", "HDP_THR_DUAL": "0"}):
        for k in ("HDP_THR_WHOLE", "HDP_THR_DUAL"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = core.compute_percentiles(x, ti, cols, q)
        assert same(got, want), (env, years, radius, kind, q)


@pytest.mark.timeout(1500)
def test_c5_full_grid_192x288_cells_x_10_members():
    """BASELINE config 5 at its full single-GPU size: thresholds of all 55 296 cells from the ten members concatenated along
    time (threshold.py:114-119: S = 1000, 80.7 GB of input), then -- the baseline freed -- the metrics of all 552 960 member
    series (20 percentiles x 12 definitions, 106 GB of int16 results).  Properties over EVERY cell and series, computed on
    the device in slabs; C-oracle equality on 64 cells strided over both hemispheres."""
    import torch
    from hdp_amd import calendar as cal, utils
    from oracle import c_oracle
    lib = _lib.ensure_device()
    dev = torch.device("cuda", 0)
    import gc
    gc.collect()
    torch.cuda.empty_cache()          # earlier tests' cached blocks: the metrics half needs ~200 GiB at once
    free_b, _ = torch.cuda.mem_get_info(dev)
    if free_b < 192 * (1 << 30):
        pytest.skip(f"needs ~190 GiB of free HBM, {free_b >> 30} GiB are free")
    ts = torch.cuda.Stream(dev)
    torch.cuda.set_stream(ts)
    stream = ts.cuda_stream
    years, M, n = 100, 10, 192 * 288
    PERC = np.linspace(0.80, 0.99, 20)
    DEFS = [[d, b, b] for d in (3, 4, 5, 6) for b in (0, 1, 2)]
    dates = utils.noleap_date_range("2000-01-01", f"{2000 + years - 1}-12-31")
    T = dates.size
    ti, cols = cal.window_columns(np.concatenate([dates] * M), 7)
    dm = cal.build_doy_map(dates)
    north, south, _ = cal.hemisphere_season_tables(dates)
    P, D, Y, n_doy = PERC.size, len(DEFS), north.shape[0], 365
    lat = np.repeat(np.linspace(-90.0, 90.0, 192), 288).astype(np.float32)       # row-major (lat, lon)
    lat_dev = torch.from_numpy(lat).to(dev)
    idx = np.unique(np.linspace(0, n - 1, 64).astype(np.int64))
    it = torch.from_numpy(idx).to(dev)

    # ---- thresholds of the whole grid
    xb = torch.empty(n * M * T, dtype=torch.float32, device=dev)                  # [cell][M * T]
    _lib.check(lib.hdp_generate_series_dev(xb.data_ptr(), n, M * T, 0, lat_dev.data_ptr(), 0, 0.7, 0.0, stream))
    tplan = core.ThresholdPlan(ti, cols, PERC, M * T)
    thr = torch.empty((n, P, n_doy), dtype=torch.float64, device=dev)
    tplan.run(xb.data_ptr(), n, thr.data_ptr(), stream)
    torch.cuda.synchronize(dev)
    assert not bool(torch.isnan(thr).any())
    assert bool((thr[:, 1:, :] >= thr[:, :-1, :]).all())                          # monotone in q, every cell and day
    xb2 = xb.view(n, M * T)
    assert bool((thr.amax(dim=(1, 2)) <= xb2.amax(dim=1).double()).all())
    assert bool((thr.amin(dim=(1, 2)) >= xb2.amin(dim=1).double()).all())
    xs_b = xb2[it].cpu().numpy()
    del xb2, xb
    torch.cuda.empty_cache()

    # ---- metrics of every member series against its cell's thresholds
    lat_m = torch.from_numpy(np.tile(lat, M)).to(dev)
    south_dev = torch.from_numpy(np.tile((lat < 0).astype(np.uint8), M)).to(dev)
    xm = torch.empty(M * n * T, dtype=torch.float32, device=dev)                  # [member][cell][T]
    _lib.check(lib.hdp_generate_series_dev(xm.data_ptr(), M * n, T, 0, lat_m.data_ptr(), 1, 0.7, 1.0 / 36500.0, stream))
    mplan = core.MetricsPlan(dm, n_doy, DEFS, north, south, P)
    out = torch.empty((4, P, D, Y, M * n), dtype=torch.int16, device=dev)
    mplan.run(xm.data_ptr(), thr.data_ptr(), n, south_dev.data_ptr(), M * n, out.data_ptr(), stream)
    torch.cuda.synchronize(dev)
    total = 0
    for p in range(P):                                                            # slabs of one percentile: 1.3 GB per metric
        hwf, hwn, hwd, hwa = (out[i, p] for i in range(4))
        assert bool((hwf >= 0).all()) and bool((hwd <= hwf).all()) and bool((hwn <= hwf).all())
        assert bool((hwa == torch.where(hwn > 0, torch.div(hwf, hwn.clamp(min=1), rounding_mode="floor"),
                                        torch.zeros_like(hwf))).all())
        if p:   # a higher percentile never has more heatwave days under a definition without breaks (b == 0)
            assert bool((hwf[::3] <= out[0, p - 1, ::3]).all())
        total += int(hwf.sum(dtype=torch.int64))
    assert total > 0

    # ---- C oracle on 64 cells (all their members)
    xs_m = xm.view(M, n, T)[:, it].cpu().numpy().reshape(M * idx.size, T)
    th_g = thr[it].cpu().numpy().transpose(0, 2, 1)                               # [cells, n_doy, P]
    rows = [0, 1, 100, 182, 357, 358, 364]                                        # the oracle sorts 15 000 samples per row
    win = cal.expand_window_table(ti, cols)[rows]
    assert np.array_equal(th_g[:, rows], c_oracle.thresholds(xs_b, win, PERC), equal_nan=True)
    og = out.view(4, P, D, Y, M, n)[..., it].cpu().numpy()                        # [4, P, D, Y, M, 64]
    met_g = np.transpose(og, (1, 2, 4, 5, 0, 3)).reshape(P, D, M * idx.size, 4, Y).astype(np.int64)
    hemi = np.tile((lat[idx] < 0).astype(np.uint8), M)
    assert np.array_equal(met_g, c_oracle.metrics(xs_m, np.concatenate([th_g] * M), dm, DEFS, north, south, hemi))


@pytest.mark.parametrize("seed", range(4))
def test_rank_selection_on_long_columns_with_ties_and_special_values(seed):
    """The ensemble shape's selection (S = 600 .. 1000 samples per day of year; value-pivot rounds, popped finish, the
    key-pivot loop behind them) on inputs it was not tuned on: heavy ties (rounded data: the secant stalls, whole tie
    groups straddle the wanted rank), constant cells, +-inf, a NaN, narrow and wide windows, shallow and deep ranks
    including q = 0 and q = 1, and the duplicate columns of the windows past the year's end -- against the C oracle."""
    from hdp_amd import calendar as cal
    from oracle import c_oracle, hdp_oracle as orc
    from tests.test_gpu_parity import same_f64
    rng = np.random.default_rng(4400 + seed)
    members = int(rng.choice([6, 8, 10]))
    dates = orc.noleap_date_range("2001-01-01", "2100-12-31")
    T = dates.size
    n_cells = 3
    x = rng.normal(10, 4, size=(members, n_cells, T)).astype(np.float32)
    mode = seed % 4
    if mode == 0:
        x = np.round(x)                          # ~30 distinct values: tie groups of hundreds
    elif mode == 1:
        x[:, 1] = np.float32(7.5)                # a constant cell
        x[:, 2] = np.round(x[:, 2], 1)
    elif mode == 2:
        x[0, 0, rng.integers(0, T, 40)] = np.inf
        x[1, 0, rng.integers(0, T, 40)] = -np.inf
        x[2, 1, rng.integers(0, T)] = np.nan
    cat = np.concatenate([x[m] for m in range(members)], axis=1)
    radius = int(rng.choice([3, 7]))
    ti, cols = cal.window_columns(np.concatenate([dates] * members), radius)
    q = np.sort(np.concatenate([rng.random(12), [0.0, 1.0, 0.5, 0.999]]))
    assert "select" in core.ThresholdPlan(ti, cols, q, cat.shape[1]).describe()      # the path under test
    with np.errstate(invalid="ignore"):
        thr = core.compute_percentiles(cat, ti, cols, q)
        rows = [0, 1, 180, 200, 358, 359, 362, 364]
        want = c_oracle.thresholds(cat, cal.expand_window_table(ti, cols)[rows], q)
    assert same_f64(thr[:, rows], want), (seed, members, radius)
