#!/usr/bin/env python
"""bench.py -- grid-cell-days/sec of compute_thresholds + compute_group_metrics on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c2|c5|tiny] [--layout cm|tm]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over the synthetic grid the config names: the thresholds kernel over
the baseline series and the metrics kernel over the measure series of every grid cell a rank owns.

Multi-GPU (BASELINE.json configs[3], the reference's only parallelism: threshold.py:161-169, metric.py:444-452):
ONE grid is sharded over the ranks by contiguous cell ranges (hdp_amd.dist.shard_bounds) -- strong scaling, no
data-path collective; `value` = cell-days of the whole grid / max-over-ranks step time.  The all-gather that
reassembles the int16 metrics on every rank runs after the timed region at the shard's real size and is reported
separately (`allgather`), through RCCL behind the library's own C ABI (hdp_allgather_dev) when the process group
is nccl.  `--weak` gives every rank the full grid instead (labelled "weak").

Inputs are resident in HBM when a kernel span starts.  At N = 1 the C3 grid does not fit (baseline + measure =
303 GB > 288 GB), so a step walks two resident bands of cells; every band gets ITS OWN series (its cells'
latitudes and cell offset), regenerated on the device between the event-timed kernel spans, and `ms_per_step`
is then the sum of the kernel spans (HIP events on the launch stream), with the wall clock of the loop reported
beside it (`wall_ms_per_step`).  With one band per rank (every N >= 2 shard of C3) nothing is regenerated and
`ms_per_step` is the wall clock between the two fences.

torch is plumbing here (process group, barrier, device buffers whose raw pointers go to the C ABI).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)
HBM_COPY_GBS = 6290.0       # measured float4 copy ceiling, same table

PERCENTILES = np.arange(0.9, 1.0, 0.01)                                   # 10 (README.md:48)
DEFINITIONS = [[3, 0, 0], [3, 1, 1], [4, 0, 0], [4, 1, 1], [5, 0, 0], [5, 1, 1]]  # 6 (README.md:51)
# configs[4] names only the counts (20 percentiles x 12 definitions); these values are this build's choice
PERCENTILES_C5 = np.linspace(0.80, 0.99, 20)
DEFINITIONS_C5 = [[d, b, b] for d in (3, 4, 5, 6) for b in (0, 1, 2)]
CONFIGS = {
    # name: (years, n_lat, n_lon, ensemble members, percentiles, definitions)
    "c3": (100, 720, 1440, 1, PERCENTILES, DEFINITIONS),  # BASELINE.json configs[2] (N = 1) and configs[3] (N = 8)
    "c2": (10, 180, 360, 1, PERCENTILES, DEFINITIONS),    # configs[1]: 3650 d x 180 x 360
    # configs[4]: 10 members x 36500 d x 192 x 288; members are concatenated along time for the thresholds
    # (threshold.py:114-119) and share their cell's thresholds in the metrics pass; members of a cell stay on one GPU
    "c5": (100, 192, 288, 10, PERCENTILES_C5, DEFINITIONS_C5),
    "tiny": (10, 16, 32, 1, PERCENTILES, DEFINITIONS),
    "tiny5": (10, 8, 16, 3, PERCENTILES_C5, DEFINITIONS_C5),
}
PARITY_CELLS = 1024   # cells of the parity sample, strided over every band of the rank's shard (>= 1000)
CPU_CELLS = 4096      # cells of the all-threads CPU-baseline sample (the parity cells first, then more of the same bands)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default=os.environ.get("HDP_BENCH_CONFIG", "c3"), choices=sorted(CONFIGS))
    ap.add_argument("--cells", type=int, default=0, help="override the number of grid cells of the whole grid (debug)")
    ap.add_argument("--weak", action="store_true", help="every rank processes the full grid (weak scaling)")
    ap.add_argument("--layout", default="cm", choices=["cm", "tm"],
                    help="cm: series-major [cell][T] inputs (the reference generator's layout, utils.py:82); "
                         "tm: time-major [T][cell] inputs (CMIP order), transposed chunk by chunk on a second stream")
    ap.add_argument("--qset", default="tail", choices=["tail", "median", "spread"],
                    help="requested quantiles: tail = the config's own (0.90 .. 0.99, what BASELINE.json names); median = ten "
                         "around 0.5 (the merge walks 750 of a window's 1500 samples); spread = 0.05 .. 0.95 (both ends)")
    ap.add_argument("--no-tm", action="store_true", help="skip the time-major variant reported in `layout_tm`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target time of the all-cores CPU baseline sample")
    ap.add_argument("--mem-fraction", type=float, default=0.88)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo only to rehearse the N>1 code path on a one-GPU box")
    ap.add_argument("--allgather-timeout", type=float, default=240.0,
                    help="seconds after which the post-run all-gather report is abandoned (N > 1)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal: every rank uses GPU 0 (needs --backend gloo and a small --cells)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: no launcher set the rank environment, so this process becomes the launcher --
        # BEFORE torch or the HIP library is touched here -- and only relays its ranks' output and worst exit code
        raise SystemExit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    from hdp_amd import _lib, calendar as cal, core, utils
    from hdp_amd import dist as hdist

    lib = _lib.ensure_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ts = torch.cuda.Stream(dev)
    torch.cuda.set_stream(ts)
    stream = ts.cuda_stream

    years, n_lat, n_lon, M, PERC, DEFS = CONFIGS[args.config]
    if args.qset == "median":
        PERC = np.linspace(0.455, 0.545, PERC.size)
    elif args.qset == "spread":
        PERC = np.linspace(0.05, 0.95, PERC.size)
    T = years * 365
    P, D = PERC.size, len(DEFS)
    n_grid = args.cells if args.cells > 0 else n_lat * n_lon
    if args.weak or world == 1:
        c_lo, c_hi = 0, n_grid
    else:
        c_lo, c_hi = hdist.shard_bounds(n_grid, world, rank)     # contiguous cell range of this rank
    cells_rank = c_hi - c_lo
    scaling = "weak" if (args.weak and world > 1) else "strong"
    grid_cells_total = n_grid * (world if scaling == "weak" else 1)

    # ---- host tables (the reference builds the same ones in Python: threshold.py:125, metric.py:410-416)
    dates = utils.noleap_date_range("2000-01-01", f"{2000 + years - 1}-12-31")
    time_index, cols = cal.window_columns(np.concatenate([dates] * M), 7)   # [n_doy, M * years]
    doy_map = cal.build_doy_map(dates)
    north, south, season_years = cal.hemisphere_season_tables(dates)
    Y = north.shape[0]
    n_doy = time_index.shape[0]
    tplan = core.ThresholdPlan(time_index, cols, PERC, M * T)
    mplan = core.MetricsPlan(doy_map, n_doy, DEFS, north, south, P)
    Yp = mplan.year_pitch

    # ---- bands: the largest equal split of this rank's cells whose buffers fit in HBM ----------------
    per_cell = 2 * M * T * 4 + n_doy * P * 8 + M * (4 * P * D * Yp * 2 + 1) + 4
    if args.layout == "tm":
        per_cell += 0   # the time-major sources are transposed into the same [cell][T] buffers chunk by chunk
    free_b, total_b = torch.cuda.mem_get_info(dev)
    budget = int(free_b * args.mem_fraction)
    n_bands = 1
    while (max(cells_rank, 1) + n_bands - 1) // n_bands * per_cell > budget:
        n_bands += 1
    bc = (max(cells_rank, 1) + n_bands - 1) // n_bands            # cells per band (the last band may own fewer)
    band_cells = [max(0, min(bc, cells_rank - b * bc)) for b in range(n_bands)]

    def raw(nbytes):
        return torch.empty(int(max(nbytes, 16)), dtype=torch.uint8, device=dev)

    # baseline [cell][M * T] (members appended along time); measure [member][cell][T] = M * bc series
    xb, xm = raw(bc * M * T * 4), raw(M * bc * T * 4)
    thr = raw(bc * n_doy * P * 8)
    out = torch.empty(max(4 * P * D * M * bc * Yp, 8), dtype=torch.int16, device=dev)
    # latitude of every cell of the grid, row-major (lat, lon); this rank owns grid cells [c_lo, c_hi)
    lat_axis = np.linspace(-90.0, 90.0, n_lat)
    grid_ids = (c_lo + np.arange(max(cells_rank, 1))) % (n_lat * n_lon)
    lat_cells = lat_axis[grid_ids // n_lon].astype(np.float32)      # [cells_rank]
    lat_dev, south_dev = [], []
    for b in range(n_bands):
        lc = lat_cells[b * bc: b * bc + max(band_cells[b], 1)]
        lat_dev.append(torch.from_numpy(np.tile(lc, M)).to(dev))
        south_dev.append(torch.from_numpy(np.tile((lc < 0).astype(np.uint8), M)).to(dev))

    # synthetic inputs, generated on the device (reference generator formula + hashed noise of (seed, cell, t)):
    # baseline = control, measure = control + warming trend (hdp/utils.py:41: t / (365*100)).  cell_offset makes
    # the noise a function of the GRID cell, so every sharding of the grid sees the same data.
    seed = 0
    weak_off = rank * n_grid if scaling == "weak" else 0

    def generate(b):
        nb = band_cells[b]
        if nb == 0:
            return
        off = weak_off + c_lo + b * bc
        _lib.check(lib.hdp_generate_series_dev(xb.data_ptr(), nb, M * T, off, lat_dev[b].data_ptr(),
                                               seed, 0.7, 0.0, stream))
        _lib.check(lib.hdp_generate_series_dev(xm.data_ptr(), M * nb, T, off * M, lat_dev[b].data_ptr(),
                                               seed + 1, 0.7, 1.0 / 36500.0, stream))

    def run_band(b, ev=None):
        nb = band_cells[b]
        if ev:
            lib.hdp_event_record(ev[0], stream)
        if nb:
            tplan.run(xb.data_ptr(), nb, thr.data_ptr(), stream)
        if ev:
            lib.hdp_event_record(ev[1], stream)
        if nb:
            mplan.run(xm.data_ptr(), thr.data_ptr(), nb, south_dev[b].data_ptr(), M * nb, out.data_ptr(), stream)
        if ev:
            lib.hdp_event_record(ev[2], stream)

    generate(0)
    torch.cuda.synchronize(dev)
    mplan.reserve(M * bc)   # exceedance scratch allocated before anything is timed
    regen = n_bands > 1     # every band gets its own series; regeneration sits outside the event-timed spans
    ev = [[[lib.hdp_event_create() for _ in range(3)] for _ in range(n_bands)] for _ in range(args.steps)]

    def step(k):
        """k >= 0: timed step k (events recorded, read back after the timed region); k < 0: warm-up"""
        for b in range(n_bands):
            if regen:
                generate(b)
            run_band(b, ev[k][b] if k >= 0 else None)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step(-1)
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    fence()
    wall = time.perf_counter() - t0
    ms = ctypes.c_float()
    t_thr, t_met = [], []
    for k in range(args.steps):
        for b in range(n_bands):
            e0, e1, e2 = ev[k][b]
            _lib.check(lib.hdp_event_elapsed_ms(e0, e1, ctypes.byref(ms))); t_thr.append(float(ms.value))
            _lib.check(lib.hdp_event_elapsed_ms(e1, e2, ctypes.byref(ms))); t_met.append(float(ms.value))
    kernel_s = (sum(t_thr) + sum(t_met)) * 1e-3
    timed = kernel_s if regen else wall          # see the module docstring
    if world > 1:
        tt = torch.tensor([timed, wall], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        timed, wall = float(tt[0].item()), float(tt[1].item())
    ms_per_step = timed * 1e3 / max(1, args.steps)
    cell_days_step = 2.0 * grid_cells_total * M * T   # thresholds pass + metrics pass over the whole job's cells
    value = cell_days_step / (ms_per_step * 1e-3)

    # ---- roofline of the dominant kernel: algorithmic bytes per launch / measured launch time --------
    nb0 = band_cells[0]
    bytes_thr = nb0 * (4 * M * T + 8 * n_doy * P)
    bytes_met = nb0 * (M * 4 * T + 8 * n_doy * P + M * 2 * 4 * Y * P * D)  # int16 metrics; thresholds shared by members
    full = [i for i in range(len(t_thr)) if band_cells[i % n_bands] == nb0] or [0]
    ms_thr, ms_met = float(np.mean([t_thr[i] for i in full])), float(np.mean([t_met[i] for i in full]))
    kern = {
        "thresholds_kernel": {"ms_per_launch": ms_thr, "algorithmic_bytes": bytes_thr,
                              "GBps": bytes_thr / ms_thr / 1e6, "frac_hbm": bytes_thr / ms_thr / 1e6 / HBM_PEAK_GBS,
                              "cell_days_per_s": nb0 * M * T / (ms_thr * 1e-3), "kernel": tplan.describe()},
        "metrics_kernel": {"ms_per_launch": ms_met, "algorithmic_bytes": bytes_met,
                           "GBps": bytes_met / ms_met / 1e6, "frac_hbm": bytes_met / ms_met / 1e6 / HBM_PEAK_GBS,
                           "cell_days_per_s": nb0 * M * T / (ms_met * 1e-3), "kernel": mplan.describe()},
    }
    # The roofline object describes the longest single kernel LAUNCH: the thresholds pass is one launch per band, the
    # metrics pass a pipeline of two kernels over batches of series (their overlapped spans have no single "launch
    # duration" a profile could confirm).  Both passes are in `kernels`, their joint figure in `both_kernels_frac`.
    n_met_batches = max(1, -(-int(M * nb0) // max(1, int(mplan.batch_cells(M * nb0)))))
    dom = "thresholds_kernel" if ms_thr >= ms_met / n_met_batches else "metrics_kernel"
    # HBM traffic of the dominant kernel: PMC counters cannot be collected from inside this process, so the per-cell
    # figure measured with rocprofv3 for THIS build and workload (profiles/traffic_per_cell.json: FETCH_SIZE doubled as
    # the guide prescribes for gfx950, + WRITE_SIZE, separate --pmc passes) is scaled to the cells of one launch; null
    # when the file does not describe this workload and kernel.
    traffic = None
    try:
        tp = json.load(open(os.path.join(ROOT, "profiles", "traffic_per_cell.json")))
        if tp.get("workload") != args.config:
            tp = tp.get("workloads", {}).get(args.config, {})
        if tp and tp[dom].get("kernel", "") in kern[dom]["kernel"]:
            traffic = float(tp[dom]["bytes_per_cell"]) * nb0
    except Exception:
        traffic = None
    roofline = {"bound": "hbm", "kernel": kern[dom]["kernel"], "achieved": kern[dom]["GBps"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": kern[dom]["frac_hbm"], "traffic": traffic,
                "traffic_source": "rocprofv3 2 x FETCH_SIZE + WRITE_SIZE per cell (profiles/traffic_per_cell.json) x cells per launch",
                "both_kernels_frac": (bytes_thr + bytes_met) / (ms_thr + ms_met) / 1e6 / HBM_PEAK_GBS,
                "pass_fracs": {"thresholds": kern["thresholds_kernel"]["frac_hbm"], "metrics": kern["metrics_kernel"]["frac_hbm"]},
                "metrics_batches_per_band": n_met_batches,
                "measured_copy_ceiling": HBM_COPY_GBS,
                "note": "thresholds performance depends on the requested quantiles: the merge walks down to the deepest "
                        "requested rank from the nearer end of the window (q = 0.90 of 1500 samples: 151 steps per row; "
                        "q = 0.5 would walk 750)"}

    # ---- pre-step: fused Celsius -> heat index -> Celsius kernel, reported beside `value`
    pre_step = None
    if rank == 0:
        try:
            n_el = int(min(nb0 * T, thr.numel() // 4, out.numel() // 2, 1 << 30)) & ~3
            e0, e1 = lib.hdp_event_create(), lib.hdp_event_create()
            rh_buf = thr[: n_el * 4]      # reuse resident scratch as (arbitrary) humidity / output operands
            lib.hdp_heat_index_celsius_f32_dev(xm.data_ptr(), rh_buf.data_ptr(), n_el, out.data_ptr(), stream)
            lib.hdp_event_record(e0, stream)
            reps = 5
            for _ in range(reps):
                _lib.check(lib.hdp_heat_index_celsius_f32_dev(xm.data_ptr(), rh_buf.data_ptr(), n_el, out.data_ptr(), stream))
            lib.hdp_event_record(e1, stream)
            hms = ctypes.c_float()
            _lib.check(lib.hdp_event_elapsed_ms(e0, e1, ctypes.byref(hms)))
            gbps = 12.0 * n_el * reps / (hms.value * 1e-3) / 1e9      # 2 x 4 B read + 4 B written per element
            pre_step = {"kernel": "heat_index_kernel<celsius>", "elements": n_el, "ms": hms.value / reps,
                        "GBps": gbps, "frac_hbm": gbps / HBM_PEAK_GBS}
        except Exception as e:   # reported beside `value`; must not cost the bench line
            pre_step = {"error": f"{type(e).__name__}: {e}"[:300]}

    # ---- time-major inputs (CMIP order [T][cell]): the chunk-pipelined device path ----------------
    layout_tm = None
    if rank == 0 and (args.layout == "tm" or not args.no_tm):
        try:
            layout_tm = bench_time_major(lib, torch, dev, stream, tplan, mplan, xb, xm, thr, out, south_dev[0],
                                         nb0, M, T, n_doy, P, ms_thr, ms_met)
        except Exception as e:
            layout_tm = {"error": f"{type(e).__name__}: {e}"[:300]}

    # ---- parity sample + CPU baseline ---------------------------------------------------------------------------
    # PARITY_CELLS cells strided over EVERY band of this rank's shard (both hemispheres at N = 1): their series are
    # copied out of the device buffers the kernels read, the kernels' own outputs for them are copied out, and the C
    # restatement of the reference algorithm (oracle/) recomputes them on the host: bit-exact or the line says so.
    # The same sample, timed, is the all-cores CPU baseline; a smaller one gives the 1-thread figure.
    cpu = None
    parity = None
    if rank == 0 and not args.no_cpu_baseline:
        try:
            from oracle import c_oracle
            cores = c_oracle.max_threads()
            win = cal.expand_window_table(time_index, cols)
            per_band = max(1, PARITY_CELLS // n_bands)
            xs_b, xs_m, th_g, met_g, hemi = [], [], [], [], []
            for b in range(n_bands):
                nb = band_cells[b]
                if nb == 0:
                    continue
                if regen or b > 0:
                    generate(b)
                run_band(b)
                torch.cuda.synchronize(dev)
                idx = np.unique(np.linspace(0, nb - 1, min(per_band, nb)).astype(np.int64))
                it = torch.from_numpy(idx).to(dev)
                xs_b.append(xb[: nb * M * T * 4].view(torch.float32).view(nb, M * T)[it].cpu().numpy())
                xs_m.append(xm[: M * nb * T * 4].view(torch.float32).view(M, nb, T)[:, it].cpu().numpy())  # [M, n, T]
                # device layouts: thresholds [cell][P][n_doy]; metrics [4][P][D][Y][M * nb] (series-minor)
                th_g.append(thr[: nb * n_doy * P * 8].view(torch.float64).view(nb, P, n_doy)[it].cpu().numpy()
                            .transpose(0, 2, 1))
                og = out[: 4 * P * D * Y * M * nb].view(4, P, D, Y, M, nb)[..., it].cpu().numpy()   # [4,P,D,Y,M,n]
                met_g.append(np.transpose(og, (1, 2, 4, 5, 0, 3)))                                   # [P,D,M,n,4,Y]
                hemi.append((lat_cells[b * bc + idx] < 0).astype(np.uint8))
            xs_b = np.concatenate(xs_b)
            th_gpu = np.concatenate(th_g)
            ns = xs_b.shape[0]
            xs_m = np.concatenate(xs_m, axis=1).reshape(M * ns, T)          # member-major, like the device buffer
            met_gpu = np.concatenate(met_g, axis=3).reshape(P, D, M * ns, 4, Y).astype(np.int64)
            hemi = np.tile(np.concatenate(hemi), M)
            n_south = int(hemi[:ns].sum())
            th_cpu = c_oracle.thresholds(xs_b, win, PERC)
            met_cpu = c_oracle.metrics(xs_m, np.concatenate([th_cpu] * M), doy_map, DEFS, north, south, hemi)
            parity = {"cells": int(ns), "southern_cells": n_south, "northern_cells": int(ns - n_south),
                      "bands_sampled": int(sum(1 for c in band_cells if c)),
                      "thresholds_bit_exact": bool(np.array_equal(th_gpu, th_cpu, equal_nan=True)),
                      "metrics_bit_exact": bool(np.array_equal(met_gpu, met_cpu))}
            # The all-threads CPU baseline: CPU_CELLS cells of the band still resident (series copied out of the device
            # buffers the kernels read), both passes, timed; the same cells' GPU results are compared once more.
            nbl = band_cells[n_bands - 1] if band_cells[n_bands - 1] else nb0
            ncpu = int(min(CPU_CELLS if M == 1 else max(64, CPU_CELLS // (4 * M)), nbl))
            idc = torch.from_numpy(np.unique(np.linspace(0, nbl - 1, ncpu).astype(np.int64))).to(dev)
            ncpu = int(idc.numel())
            cb = xb[: nbl * M * T * 4].view(torch.float32).view(nbl, M * T)[idc].cpu().numpy()
            cm = xm[: M * nbl * T * 4].view(torch.float32).view(M, nbl, T)[:, idc].cpu().numpy().reshape(M * ncpu, T)
            lat_last = lat_cells[(n_bands - 1) * bc + idc.cpu().numpy()] if band_cells[n_bands - 1] else lat_cells[idc.cpu().numpy()]
            ch = np.tile((lat_last < 0).astype(np.uint8), M)
            tc = time.perf_counter()
            th_c = c_oracle.thresholds(cb, win, PERC)
            c_oracle.metrics(cm, np.concatenate([th_c] * M), doy_map, DEFS, north, south, ch)
            cpu_s = time.perf_counter() - tc
            # 1-thread figure on a few cells of the same sample
            n1 = int(max(1, min(ncpu, 16)))
            one = c_oracle.set_threads(1)
            t1 = time.perf_counter()
            th1 = c_oracle.thresholds(cb[:n1], win, PERC)
            c_oracle.metrics(cm.reshape(M, ncpu, T)[:, :n1].reshape(M * n1, T), np.concatenate([th1] * M), doy_map,
                             DEFS, north, south, np.tile(ch[:n1], M))
            one_s = time.perf_counter() - t1
            c_oracle.set_threads(cores)
            host = host_cpu()
            cpu = None if world > 1 else {   # reported at N = 1 only (torchrun pins OMP_NUM_THREADS=1)
                "value": 2.0 * ncpu * M * T / cpu_s, "unit": "cell-days/s",
                "cores": int(host["physical_cores"] or cores), "threads": cores, "kind": "port",
                "cpu_model": host["model"], "physical_cores": host["physical_cores"], "logical_cpus": host["logical_cpus"],
                "sockets": host["sockets"],
                "thread_binding": (f"OpenMP, {cores} threads (omp_get_max_threads), no explicit pinning (OMP_PROC_BIND="
                                   f"{os.environ.get('OMP_PROC_BIND', 'unset')}); {host['affinity_cpus']} CPUs in the affinity mask"),
                "sample": f"{ncpu} cells strided over one band of the same workload (T={T}, P={P}, D={D}), both "
                          f"passes, {cpu_s:.1f} s; oracle/hdp_oracle.c (reference algorithm restated in C, OpenMP over cells)",
                "one_thread": {"value": 2.0 * n1 * M * T / one_s, "unit": "cell-days/s", "cores": int(one),
                               "sample": f"first {n1} cells of that sample, {one_s:.1f} s"}}
        except Exception as e:   # the checker must not cost the bench line
            parity = {"error": f"{type(e).__name__}: {e}"[:300]}

    def emit(allgather):
        line = {
            "metric": "grid-cell-days/sec for compute_thresholds+compute_group_metrics",
            "value": value, "unit": "cell-days/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {f'{M} members x ' if M > 1 else ''}{T} d x {n_lat} x {n_lon} fp32, {P} percentiles x {D} definitions, "
                                   f"window radius 7, noleap" + (f" ({n_grid} cells)" if args.cells else "")
                                   + (f" [quantile set: {args.qset}]" if args.qset != "tail" else ""),
                       "grid_cells": int(n_grid), "cells_per_gpu": int(cells_rank), "members": M, "T": T, "percentiles": P,
                       "definitions": D, "seasons": int(Y), "resident_bands_per_step": n_bands, "cells_per_band": int(bc),
                       "band_data": ("every band has its own series (regenerated on the device between the event-timed kernel "
                                     "spans; ms_per_step = sum of the spans)" if regen else "whole shard resident"),
                       "layout": "series-major [cell][T] (the reference generator's layout); the time-major [T][cell] variant of both "
                                 "passes is timed on a slice of the same data in `layout_tm`",
                       "sharding": ("one grid, contiguous cell ranges per rank (hdp_amd.dist.shard_bounds), no data-path collective"
                                    if scaling == "strong" else "every rank processes the full grid")},
            "wall_ms_per_step": wall * 1e3 / max(1, args.steps),
            "roofline": roofline, "cpu_baseline": cpu, "kernels": kern, "pre_step": pre_step, "layout_tm": layout_tm,
            "allgather": allgather, "parity_sample": parity, "device": _lib.device_info(),
            # which runtime libraries this process mapped: the first thing a failed multi-GPU run's log must say
            "runtime": dict(_lib.runtime_report(), torch=torch.__version__,
                            collective_transport=(None if world == 1 else
                                                  ("library RCCL communicator (hdp_comm_*)" if hdist.comm_ready() else
                                                   f"torch.distributed {args.backend}"))),
        }
        print(json.dumps(line), flush=True)

    # ---- all-gather of the metrics (reassembly step of north_star), reported separately -----------------------
    # It runs after the timed region and must never cost the bench line: an exception is reported in the line, and a
    # collective that does not come back within --allgather-timeout seconds (a second RCCL communicator next to torch's
    # has only been rehearsed with one rank on the one-GPU box) is abandoned -- rank 0 prints the line without it and every
    # rank leaves through os._exit(3): non-zero, so a launcher never takes a hung collective for a success.
    allgather = None
    if world > 1:
        import threading
        done = threading.Event()

        def bail():
            if done.is_set():
                return
            if rank == 0:
                emit({"error": f"all-gather did not finish within {args.allgather_timeout:.0f} s; abandoned"})
            sys.stdout.flush()
            os._exit(3)     # an abandoned collective is a failure of the run: the launcher must see it

        timer = threading.Timer(args.allgather_timeout, bail)
        timer.daemon = True
        timer.start()
        try:
            allgather = bench_allgather(lib, torch, dist, hdist, dev, stream, args, out, world, rank, fence,
                                        4 * P * D * Yp * M * bc)
        except Exception as e:
            allgather = {"error": f"{type(e).__name__}: {e}"[:300]}
        done.set()
        timer.cancel()
    if rank == 0:
        emit(allgather)
    if world > 1:
        dist.barrier()
        if hdist.comm_world() > 1 or hdist.comm_ready():
            hdist.comm_destroy()
        dist.destroy_process_group()


def host_cpu():
    """{model, physical_cores, logical_cpus, sockets} from /proc/cpuinfo (the CPU baseline names its host)"""
    model, cores, sockets, logical = None, set(), set(), 0
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "processor":
                logical += 1
            elif k == "model name" and model is None:
                model = v
            elif k == "physical id":
                phys = v
                sockets.add(v)
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core))
                phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    return {"model": model, "physical_cores": len(cores) or None, "logical_cpus": logical or os.cpu_count(),
            "sockets": len(sockets) or None, "affinity_cpus": len(os.sched_getaffinity(0))}


def spawn_ranks(n, script=None, argv=None):
    """Start `n` fresh rank processes of this script (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as a
    launcher would) and wait for them.  Rank 0 inherits stdout, so its JSON line is this command's line; the return
    value is the worst exit code of the ranks (a rank killed by a signal counts as 128 + signal)."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "1" if r else str(os.cpu_count() or 1))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)]
                                      + list(sys.argv[1:] if argv is None else argv), env=env))
    # Poll every rank: the first one that exits non-zero (or is killed) takes the others down after a short grace period,
    # as torchrun does -- otherwise they sit in the rendezvous or a collective until torch's own timeout, minutes later.
    worst, failed_at = 0, None
    try:
        while True:
            alive = 0
            for pr in procs:
                rc = pr.poll()
                if rc is None:
                    alive += 1
                    continue
                rc = 128 - rc if rc < 0 else rc
                worst = max(worst, rc)
                if rc and failed_at is None:
                    failed_at = time.monotonic()
            if not alive:
                break
            if failed_at is not None:
                late = time.monotonic() - failed_at
                if late > 1.0:      # a moment for the others to fail on their own account (and say why) first
                    for pr in procs:
                        if pr.poll() is None:
                            (pr.kill if late > 11.0 else pr.terminate)()
            time.sleep(0.05)
    except KeyboardInterrupt:
        for pr in procs:
            pr.terminate()
        worst = max(worst, 130)
    return worst


def bench_time_major(lib, torch, dev, stream, tplan, mplan, xb, xm, thr, out, south, nb, M, T, n_doy, P, ms_thr, ms_met):
    """Both passes from time-major [T][cell] sources (CMIP order; run_cmip_workflow.py:31-32): hdp_*_tm_dev transposes
    chunks of cells on the plan's copy stream while the previous chunk computes.  Sources are built by transposing a slice
    of the resident series-major inputs (so results can be compared), sized to the memory left."""
    from hdp_amd import _lib
    if M != 1:
        return {"skipped": "time-major bench variant is wired for single-member configs"}
    free_b, _ = torch.cuda.mem_get_info(dev)
    # ONE time-major source at a time (the baseline series for the thresholds pass, then the measure series for the
    # metrics pass), beside the plans' series-major staging (two chunks each, <= 10 GiB per plan) and the second set of
    # results: as many cells as that leaves, so that the chunk pipeline runs more than its fill
    per_cell = M * T * 4 + 8 * n_doy * P + 2 * (out.numel() // nb)
    n = int(min(nb, max(1024, (free_b * 0.9 - (22 << 30)) // per_cell)))
    src = torch.empty((T, n), dtype=torch.float32, device=dev)
    thr2 = torch.empty(n * n_doy * P, dtype=torch.float64, device=dev)
    out_n = out.numel() // nb * n
    out2 = torch.empty(out_n, dtype=torch.int16, device=dev)
    tplan.run(xb.data_ptr(), n, thr.data_ptr(), stream)
    mplan.run(xm.data_ptr(), thr.data_ptr(), n, south.data_ptr(), n, out.data_ptr(), stream)
    ev = [lib.hdp_event_create() for _ in range(2)]
    ms = ctypes.c_float()
    res = {}

    def fill(x):   # src <- x[:n]^T in slabs (a whole-array .t() copy would want a temporary of its own)
        xs = x[: n * T * 4].view(torch.float32).view(n, T)
        step = 16384
        for c0 in range(0, n, step):
            src[:, c0:c0 + step].copy_(xs[c0:c0 + step].t())

    fill(xb)
    for it in range(2):   # first pass warms the plan's staging buffers
        lib.hdp_event_record(ev[0], stream)
        _lib.check(lib.hdp_thresholds_f32_tm_dev(tplan.handle, src.data_ptr(), n, n, thr2.data_ptr(), stream))
        lib.hdp_event_record(ev[1], stream)
        _lib.check(lib.hdp_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))); res["thresholds_ms"] = float(ms.value)
    torch.cuda.synchronize(dev)
    fill(xm)
    for it in range(2):
        lib.hdp_event_record(ev[0], stream)
        _lib.check(lib.hdp_metrics_f32_tm_dev(mplan.handle, src.data_ptr(), n, thr2.data_ptr(), n, south.data_ptr(), n,
                                              out2.data_ptr(), stream))
        lib.hdp_event_record(ev[1], stream)
        _lib.check(lib.hdp_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))); res["metrics_ms"] = float(ms.value)
    torch.cuda.synchronize(dev)
    same_thr = bool(torch.equal(thr2.view(torch.int64), thr[: n * n_doy * P * 8].view(torch.int64)))
    same_met = bool(torch.equal(out2.view(-1, n), out[: out_n].view(-1, n)))
    scale = n / nb
    res.update({"cells": n, "series_major_thresholds_ms_same_cells": ms_thr * scale, "series_major_metrics_ms_same_cells": ms_met * scale,
                "cell_days_per_s": 2.0 * n * T / ((res["thresholds_ms"] + res["metrics_ms"]) * 1e-3),
                "identical_to_series_major": same_thr and same_met,
                "note": "includes the transposition of every chunk (read + write of the whole input, on a second stream "
                        "beside the kernels)"})
    return res


def bench_allgather(lib, torch, dist, hdist, dev, stream, args, out, world, rank, fence, shard_elems):
    """All-gather of the int16 metrics at this run's real shard size (config 4 at N = 8: 6.2 GB per rank).  With the
    nccl backend the bytes go through the library's own RCCL communicator (hdp_comm_* / hdp_allgather_dev, the C ABI a
    torch-free caller uses); the gloo rehearsal moves host bytes through torch.distributed."""
    from hdp_amd import _lib
    share = int(min(out.numel(), shard_elems))
    free_b, _ = torch.cuda.mem_get_info(dev)
    share = int(min(share, free_b * 0.8 / 2 / world)) & ~3
    if args.backend != "nccl":
        share = min(share, 1 << 24)
    # every rank must hand the SAME count to the collective: shards differ when the grid does not divide by the world
    # and free memory differs per GPU, and unequal counts are undefined for ncclAllGather / ncclSend / ncclRecv
    agreed = torch.tensor([share], dtype=torch.int64, device=dev if args.backend == "nccl" else "cpu")
    dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
    share = int(agreed.item()) & ~3
    if share <= 0:
        return {"skipped": "a rank has no metrics to gather"}
    # one checksum per rank of the bytes it contributes, gathered through torch's group: every slice of the gathered
    # buffer is compared with its owner's checksum, not only this rank's own
    mine = out[:share].view(torch.int32).sum(dtype=torch.int64).reshape(1)      # no widened temporary
    sums = torch.empty(world, dtype=torch.int64, device=mine.device)
    if args.backend == "nccl":
        dist.all_gather_into_tensor(sums, mine)
    else:
        sums_c = torch.empty(world, dtype=torch.int64)
        dist.all_gather_into_tensor(sums_c, mine.cpu())
        sums = sums_c
        gathered = torch.empty(share * world, dtype=torch.int16)
        g8, o8 = gathered.view(torch.uint8), out[:share].cpu().view(torch.uint8)
        dist.all_gather_into_tensor(g8, o8)
        fence()
        t1 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            dist.all_gather_into_tensor(g8, o8)
        fence()
        dt = (time.perf_counter() - t1) / reps
        ok = bool(torch.equal(gathered.view(world, share).view(torch.int32).sum(dim=1, dtype=torch.int64), sums))
        return {"path": "gloo rehearsal (host bytes)", "bytes_per_rank": share * 2, "ms": dt * 1e3,
                "every_shard_intact": ok}
    # RCCL behind the C ABI: rank 0 makes the unique id, torch.distributed carries its 128 bytes to the other ranks
    ident = torch.zeros(hdist.COMM_ID_BYTES, dtype=torch.uint8)
    if rank == 0:
        ident = torch.from_numpy(np.frombuffer(hdist.comm_unique_id(), dtype=np.uint8).copy())
    idd = ident.to(dev)
    dist.broadcast(idd, 0)
    hdist.comm_init_rank(bytes(idd.cpu().numpy().tobytes()), rank, world)
    gathered = torch.empty(share * world, dtype=torch.int16, device=dev)
    res = {"path": "RCCL ncclAllGather behind hdp_allgather_dev", "bytes_per_rank": share * 2}
    for name, fn in (("allgather", lib.hdp_allgather_dev), ("sendrecv", lib.hdp_allgather_direct_dev)):
        _lib.check(fn(out.data_ptr(), share * 2, gathered.data_ptr(), stream))
        fence()
        t1 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            _lib.check(fn(out.data_ptr(), share * 2, gathered.data_ptr(), stream))
        fence()
        dt = (time.perf_counter() - t1) / reps
        ok = bool(torch.equal(gathered[rank * share:(rank + 1) * share], out[:share]))
        ok_all = bool(torch.equal(gathered.view(world, share).view(torch.int32).sum(dim=1, dtype=torch.int64), sums))
        res[name] = {"ms": dt * 1e3, "recv_GBps_per_gpu": share * 2 * (world - 1) / dt / 1e9, "own_shard_intact": ok,
                     "every_shard_intact": ok_all}
    res["ms"] = res["allgather"]["ms"]
    return res


if __name__ == "__main__":
    main()
