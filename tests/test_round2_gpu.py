"""Round-2 GPU tests (through the C ABI, checked against the C oracle):

* BASELINE config 3 at the size the bench runs per resident band (518 400 cells x 36 500 days, 10 percentiles x
  6 definitions), device-generated, both hemispheres: size-independent properties over EVERY cell and C-oracle
  equality on 512 strided cells;
* season tables that overlap / are unordered (the reference's compute_heatwave_metrics takes any ranges; its unit
  tests use overlapping ones) through the product entry points;
* time-major device inputs (CMIP order) against the series-major path;
* the RCCL communicator behind the C ABI (world of one: a 1-GPU box cannot host two ranks on one device).
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from hdp_amd import _lib, calendar as cal, core, utils  # noqa: E402
from hdp_amd import dist as hdist  # noqa: E402
from oracle import c_oracle  # noqa: E402
from oracle import hdp_oracle as orc  # noqa: E402

PERC = np.arange(0.9, 1.0, 0.01)
DEFS = [[3, 0, 0], [3, 1, 1], [4, 0, 0], [4, 1, 1], [5, 0, 0], [5, 1, 1]]


def _c3_tables(years=100):
    dates = utils.noleap_date_range("2000-01-01", f"{2000 + years - 1}-12-31")
    ti, cols = cal.window_columns(dates, 7)
    doy_map = cal.build_doy_map(dates)
    north, south, _ = cal.hemisphere_season_tables(dates)
    return dates, ti, cols, doy_map, north, south


@pytest.mark.timeout(900)
def test_c3_full_band_properties_and_oracle_sample():
    import torch
    lib = _lib.ensure_device()
    dev = torch.device("cuda", 0)
    ts = torch.cuda.Stream(dev)
    torch.cuda.set_stream(ts)
    stream = ts.cuda_stream
    dates, ti, cols, doy_map, north, south = _c3_tables()
    T, P, D, Y, n_doy = dates.size, PERC.size, len(DEFS), north.shape[0], ti.shape[0]
    n_lat, n_lon = 720, 1440
    n = n_lat * n_lon // 2                       # one resident band of the bench: 518 400 cells
    free_b, _ = torch.cuda.mem_get_info(dev)
    need = n * (2 * T * 4 + n_doy * P * 8 + 4 * P * D * Y * 2 + 8)
    if need > free_b * 0.9:
        pytest.skip(f"needs {need / 2**30:.0f} GiB of HBM")
    # every second cell of the 720 x 1440 grid: all 720 latitude rows, both hemispheres, in one band
    grid_id = np.arange(n, dtype=np.int64) * 2
    lat = np.linspace(-90.0, 90.0, n_lat)[grid_id // n_lon].astype(np.float32)
    lat_dev = torch.from_numpy(lat).to(dev)
    south_dev = torch.from_numpy((lat < 0).astype(np.uint8)).to(dev)
    xb = torch.empty(n * T, dtype=torch.float32, device=dev)
    xm = torch.empty(n * T, dtype=torch.float32, device=dev)
    _lib.check(lib.hdp_generate_series_dev(xb.data_ptr(), n, T, 7, lat_dev.data_ptr(), 0, 0.7, 0.0, stream))
    _lib.check(lib.hdp_generate_series_dev(xm.data_ptr(), n, T, 7, lat_dev.data_ptr(), 1, 0.7, 1.0 / 36500.0, stream))
    tplan = core.ThresholdPlan(ti, cols, PERC, T)
    mplan = core.MetricsPlan(doy_map, n_doy, DEFS, north, south, P)
    assert "lane_kernel" in tplan.describe()
    thr = torch.empty((n, P, n_doy), dtype=torch.float64, device=dev)
    out = torch.empty((4, P, D, Y, n), dtype=torch.int16, device=dev)
    tplan.run(xb.data_ptr(), n, thr.data_ptr(), stream)
    mplan.run(xm.data_ptr(), thr.data_ptr(), n, south_dev.data_ptr(), n, out.data_ptr(), stream)
    torch.cuda.synchronize(dev)

    # ---- properties over every cell (on the device)
    assert not bool(torch.isnan(thr).any())
    assert bool((thr[:, 1:, :] >= thr[:, :-1, :]).all()), "thresholds must not decrease with the percentile"
    hwf, hwn, hwd, hwa = out[0], out[1], out[2], out[3]
    assert int(hwf.min()) >= 0 and int(hwf.max()) <= 153
    assert bool((hwf >= hwd).all()) and bool((hwd >= hwa).all()) and bool((hwn <= hwf).all())
    for p in range(P):          # slab by slab: int32 temporaries of the whole array would not fit beside it
        f, k = hwf[p].to(torch.int32), hwn[p].to(torch.int32)
        want_a = torch.where(k > 0, torch.div(f, torch.clamp(k, min=1), rounding_mode="floor"), torch.zeros_like(f))
        assert bool((hwa[p].to(torch.int32) == want_a).all()), "HWA == HWF // HWN"
        assert bool((f[k == 0] == 0).all())
    assert int(hwf.to(torch.int64).sum()) > 0

    # ---- C-oracle equality on 512 cells strided over the band (256 southern + 256 northern latitude rows hit)
    idx = np.unique(np.linspace(0, n - 1, 512).astype(np.int64))
    it = torch.from_numpy(idx).to(dev)
    xs_b = xb.view(n, T)[it].cpu().numpy()
    xs_m = xm.view(n, T)[it].cpu().numpy()
    th_gpu = thr[it].cpu().numpy().transpose(0, 2, 1)                       # (cell, doy, percentile)
    met_gpu = np.transpose(out[..., it].cpu().numpy(), (1, 2, 4, 0, 3)).astype(np.int64)  # [P, D, n, 4, Y]
    hemi = (lat[idx] < 0).astype(np.uint8)
    assert 0 < hemi.sum() < idx.size
    win = cal.expand_window_table(ti, cols)
    th_cpu = c_oracle.thresholds(xs_b, win, PERC)
    assert np.array_equal(th_gpu, th_cpu)
    met_cpu = c_oracle.metrics(xs_m, th_cpu, doy_map, DEFS, north, south, hemi)
    assert np.array_equal(met_gpu, met_cpu)


def test_overlapping_and_unordered_season_ranges_match_the_oracle():
    """metric.py:304-341 takes any season_ranges (hdp/tests/test_heatwave_frequency.py:39 uses [[0,5],[0,10],[20,30],[42,50]]):
    the product entry points route such tables to the per-series path instead of refusing them."""
    rng = np.random.default_rng(5)
    dates = orc.noleap_date_range("2001-01-01", "2003-12-31")
    T, n = dates.size, 37
    t = np.arange(T)
    x = (15 + 8 * np.sin(2 * np.pi * t / 365.0)[None, :] + rng.normal(0, 2.5, size=(n, T))).astype(np.float32)
    ti, cols = cal.window_columns(dates, 7)
    q = [0.8, 0.9]
    thr = core.compute_percentiles(x, ti, cols, q)
    doy_map = cal.build_doy_map(dates)
    defs = [[3, 0, 0], [2, 1, 1], [1, 2, 3]]
    north = np.array([[0, 200], [100, 300], [250, 260], [40, 900]], dtype=np.int64)      # overlapping
    south = np.array([[700, 1000], [10, 50], [0, T], [500, 640]], dtype=np.int64)        # unordered, whole record
    is_south = (np.arange(n) % 3 == 0).astype(np.uint8)
    got = core.compute_heatwave_metrics(x, thr, doy_map, defs, north, south, is_south).astype(np.int64)
    want = c_oracle.metrics(x, thr, doy_map, defs, north, south, is_south)
    assert np.array_equal(got, want)
    planes = core.compute_heatwave_metric_planes(x, thr, doy_map, defs, north, south, is_south)
    assert np.array_equal(np.moveaxis(planes, 0, 3), want)
    # the single-series mirror of the njit function, with the reference test's own ranges
    from hdp_amd import metric as hmetric
    one = hmetric.compute_heatwave_metrics(x[0], thr[0, :, 0], doy_map, 3, 1, 1, [[0, 5], [0, 10], [20, 30], [42, 50]])
    ref = c_oracle.metrics(x[:1], thr[:1, :, :1], doy_map, [[3, 1, 1]], np.array([[0, 5], [0, 10], [20, 30], [42, 50]]),
                           np.array([[0, 5], [0, 10], [20, 30], [42, 50]]), np.zeros(1, np.uint8))
    assert np.array_equal(one, ref[0, 0, 0])


def test_time_major_device_inputs_match_series_major():
    """hdp_thresholds_f32_tm_dev / hdp_metrics_f32_tm_dev on [T][cells] inputs (CMIP order), with a pitch wider than the
    cell count and enough cells for several staging chunks, against the series-major entry points: identical bytes."""
    import torch
    lib = _lib.ensure_device()
    dev = torch.device("cuda", 0)
    ts = torch.cuda.Stream(dev)
    torch.cuda.set_stream(ts)
    stream = ts.cuda_stream
    dates, ti, cols, doy_map, north, south = _c3_tables(years=12)
    T, P, D, Y, n_doy = dates.size, PERC.size, len(DEFS), north.shape[0], ti.shape[0]
    n, pitch = 5000, 5120
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    season = 20 + 2 * torch.sin(2 * np.pi * (torch.arange(T, device=dev) + 90) / 365.0)
    xb = (season[None, :] + torch.rand((n, T), device=dev, generator=g) * 0.7).float().contiguous()
    xm = (xb + torch.arange(T, device=dev)[None, :] / 36500.0 + 0.05).float().contiguous()
    tm_b = torch.full((T, pitch), float("nan"), dtype=torch.float32, device=dev)
    tm_m = torch.full((T, pitch), float("nan"), dtype=torch.float32, device=dev)
    tm_b[:, :n] = xb.t()
    tm_m[:, :n] = xm.t()
    south_dev = (torch.arange(n, device=dev) % 2).to(torch.uint8)
    tplan = core.ThresholdPlan(ti, cols, PERC, T)
    mplan = core.MetricsPlan(doy_map, n_doy, DEFS, north, south, P)
    thr_a = torch.empty(n * P * n_doy, dtype=torch.float64, device=dev)
    thr_b = torch.empty_like(thr_a)
    out_a = torch.zeros(4 * P * D * Y * n, dtype=torch.int16, device=dev)
    out_b = torch.zeros_like(out_a)
    tplan.run(xb.data_ptr(), n, thr_a.data_ptr(), stream)
    mplan.run(xm.data_ptr(), thr_a.data_ptr(), n, south_dev.data_ptr(), n, out_a.data_ptr(), stream)
    tplan.run_time_major(tm_b.data_ptr(), pitch, n, thr_b.data_ptr(), stream)
    mplan.run_time_major(tm_m.data_ptr(), pitch, thr_b.data_ptr(), n, south_dev.data_ptr(), n, out_b.data_ptr(), stream)
    torch.cuda.synchronize(dev)
    assert bool(torch.equal(thr_a.view(torch.int64), thr_b.view(torch.int64)))
    assert bool(torch.equal(out_a, out_b)) and int(out_a.to(torch.int64).sum()) > 0


def test_comm_of_one_rank_allgathers_in_place():
    """hdp_comm_* / hdp_allgather_dev (RCCL linked behind the C ABI, no torch): a world of one on this box -- the id
    hand-off, ncclCommInitRank, both all-gather forms, and dist.allgather_cells riding on them."""
    lib = _lib.ensure_device()
    ident = hdist.comm_unique_id()
    assert len(ident) == hdist.COMM_ID_BYTES and any(ident)
    hdist.comm_init_rank(ident, 0, 1)
    try:
        assert hdist.comm_ready() and hdist.comm_world() == 1 and hdist.comm_rank() == 0
        assert hdist.current() == (0, 1) and hdist.current((3, 8)) == (3, 8)
        with pytest.raises(_lib.HdpError):
            hdist.comm_init_rank(ident, 0, 1)            # a communicator already exists
        a = np.arange(4096, dtype=np.int16)
        send, recv = lib.hdp_dev_alloc(a.nbytes), lib.hdp_dev_alloc(a.nbytes)
        for fn in (lib.hdp_allgather_dev, lib.hdp_allgather_direct_dev):
            got = np.zeros_like(a)
            _lib.check(lib.hdp_dev_memset(recv, 0, a.nbytes))
            _lib.check(lib.hdp_memcpy_h2d(send, a.ctypes.data_as(C.c_void_p), a.nbytes))
            _lib.check(fn(send, a.nbytes, recv, None))
            _lib.check(lib.hdp_memcpy_d2h(got.ctypes.data_as(C.c_void_p), recv, a.nbytes))
            assert np.array_equal(got, a)
        lib.hdp_dev_free(send)
        lib.hdp_dev_free(recv)
        local = np.arange(2 * 5 * 3, dtype=np.int64).reshape(2, 5, 3)
        assert np.array_equal(hdist.allgather_cells(local, 5, 1), local)
    finally:
        hdist.comm_destroy()
    assert not hdist.comm_ready()


def _regular_case(seed, n_doy, T, n, P, defs, long_runs):
    """A record on a REGULAR calendar (doy_map[t] = t mod n_doy), seasons given per year; hot spells across year ends."""
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 1, size=(n, T)).astype(np.float32)
    if long_runs:
        for s in range(n):
            for _ in range(6):   # hot spells of 5..90 days, some across a year end and across the record's end
                a = int(rng.integers(0, T))
                if rng.random() < 0.5:
                    a = int(rng.integers(1, max(2, T // n_doy + 1))) * n_doy - int(rng.integers(1, 30))
                x[s, max(a, 0): a + int(rng.integers(5, 90))] = 9.0
        x[0, T - 70:] = 9.0                # open at the end of the record
        x[1 % n, : min(T, 400)] = 9.0      # a run longer than a year's first words
    thr = np.sort(rng.normal(0.7, 0.4, size=(n, n_doy, P)), axis=2)
    doy_map = np.arange(T, dtype=np.int64) % n_doy
    years = (T + n_doy - 1) // n_doy
    north = np.array([[y * n_doy + 120, min(T, y * n_doy + 273)] for y in range(years) if y * n_doy + 120 < T])
    south = np.array([[y * n_doy + 304, min(T, (y + 1) * n_doy + 90)] for y in range(years) if y * n_doy + 304 < T])
    Y = min(len(north), len(south))
    is_south = (np.arange(n) % 3 == 1).astype(np.uint8)
    return x, thr, doy_map, defs, north[:Y], south[:Y], is_south


@pytest.mark.gpu
@pytest.mark.parametrize("n_doy,T,P,defs", [
    (365, 365 * 9, 10, [[3, 0, 0], [3, 1, 1], [4, 0, 0], [4, 1, 1], [5, 0, 0], [5, 1, 1]]),   # whole years
    (365, 365 * 9 + 1, 3, [[3, 0, 0], [2, 1, 1]]),          # one day of a tenth year
    (365, 365 * 8 + 330, 13, [[3, 1, 2], [6, 2, 0]]),        # two percentile groups (10 + 3); last span of the partial year
    (365, 365 * 8 + 64, 1, [[1, 0, 0], [0, 3, 1]]),          # exactly one word of the last year
    (365, 365 * 8 + 63, 2, [[25, 1, 1], [30, 0, 0]]),        # min_duration beyond the 19 repeated days of a short word
    (360, 360 * 7 + 200, 4, [[3, 0, 0], [40, 2, 1]]),        # 360-day calendar: 40-day last word
    (366, 366 * 6, 5, [[3, 1, 1], [22, 0, 0]]),
    (321, 321 * 7 + 5, 2, [[3, 0, 0], [2, 1, 1]]),           # one-day last word, 63 repeated days
    (384, 384 * 6 + 100, 3, [[3, 0, 0], [64, 1, 1]]),        # six full words per year, nothing repeated
    (365, 365 * 40 + 17, 10, [[3, 0, 0], [3, 1, 1], [5, 2, 2]]),  # 40 years: the steady-state loop with five years of loads in flight
    (365, 365 * 31, 2, [[4, 1, 1]]),
    (365, 500, 11, [[3, 0, 0], [2, 1, 1], [1, 0, 0]]),        # a year and a bit; 10 + 1 percentiles; an odd number of simple definitions
    (365, 365 * 26, 4, [[3, 0, 0], [4, 0, 0], [5, 0, 0], [6, 0, 0]]),  # every pair simple
])
def test_year_aligned_exceedance_words_match_the_oracle_and_the_day_aligned_path(n_doy, T, P, defs, monkeypatch):
    """exceed_years_kernel (regular calendars; six words per year, the last one running into the next year, thresholds in
    registers, words through the scalar cache) + the state machines reading that format: against the C oracle and against
    the day-aligned exceed_pairs_kernel path (HDP_METRICS_YEARS=0), 70 series (a ragged second wave), hot spells across
    year ends and the end of the record."""
    monkeypatch.setenv("HDP_METRICS_YEARS", "2")     # also for records shorter than 24 years
    case = _regular_case(4000 + n_doy + P, n_doy, T, 70, P, defs, long_runs=True)
    x, thr, doy_map, dfs, north, south, is_south = case
    plan = core.MetricsPlan(doy_map, n_doy, dfs, north, south, P)
    assert "exceed_years_kernel" in plan.describe()
    want = c_oracle.metrics(x, thr, doy_map, dfs, north, south, is_south)
    got = core.compute_heatwave_metrics(*case)
    assert np.array_equal(got.astype(np.int64), want)
    monkeypatch.setenv("HDP_METRICS_BATCH", "32")    # three batches: the double-buffered scratch in the new format
    assert np.array_equal(core.compute_heatwave_metrics(*case), got)
    monkeypatch.delenv("HDP_METRICS_BATCH")
    monkeypatch.setenv("HDP_METRICS_YEARS", "0")
    plan0 = core.MetricsPlan(doy_map, n_doy, dfs, north, south, P)
    assert "exceed_years_kernel" not in plan0.describe()
    assert np.array_equal(core.compute_heatwave_metrics(*case), got)


@pytest.mark.gpu
def test_irregular_calendars_keep_the_day_aligned_words():
    """A leap-year calendar (doy_map is not t mod n_doy) must not take the year-aligned path."""
    dates = utils.noleap_date_range("2000-01-01", "2003-12-31")
    doy_map = cal.build_doy_map(dates).copy()
    doy_map[500:] = (doy_map[500:] + 1) % 365     # a shifted calendar from day 500 on
    north, south, _ = cal.hemisphere_season_tables(dates)
    plan = core.MetricsPlan(doy_map, 365, [[3, 0, 0]], north, south, 2)
    assert "exceed_years_kernel" not in plan.describe()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_random_regular_calendars_definitions_and_record_lengths(seed, monkeypatch):
    """Randomised sweep of the round-2 metrics path: calendar length in (320, 384], record length (whole and partial years, short
    and beyond 24 years), 1..12 percentiles, 1..8 definitions with a random share of max_break = 0 (simple pairs, odd counts,
    two passes), both hemispheres inside every wave -- year-aligned path against the oracle and the day-aligned path."""
    rng = np.random.default_rng(9000 + seed)
    n_doy = int(rng.choice([365, 365, 365, 360, 366, int(rng.integers(321, 385))]))
    years = int(rng.choice([2, 5, 9, 26, 33]))
    T = years * n_doy + int(rng.choice([0, 0, 1, 63, 64, 200, n_doy - 1]))
    if T > 65535:
        T = 65535
    P = int(rng.integers(1, 13))
    D = int(rng.integers(1, 9))
    defs = [[int(rng.integers(0, 8)), 0 if rng.random() < 0.5 else int(rng.integers(1, 4)), int(rng.integers(0, 3))]
            for _ in range(D)]
    monkeypatch.setenv("HDP_METRICS_YEARS", "2")
    case = _regular_case(9100 + seed, n_doy, T, 67, P, defs, long_runs=bool(seed & 1))
    x, thr, doy_map, dfs, north, south, is_south = case
    if north.shape[0] == 0:
        pytest.skip("no complete season in this record")
    want = c_oracle.metrics(x, thr, doy_map, dfs, north, south, is_south)
    got = core.compute_heatwave_metrics(*case)
    assert np.array_equal(got.astype(np.int64), want), (n_doy, T, P, defs)
    monkeypatch.setenv("HDP_METRICS_YEARS", "0")
    monkeypatch.setenv("HDP_METRICS_SIMPLE", "0")
    assert np.array_equal(core.compute_heatwave_metrics(*case), got), (n_doy, T, P, defs)
