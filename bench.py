#!/usr/bin/env python
"""bench.py -- grid-cell-days/sec of compute_thresholds + compute_group_metrics on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c2|tiny]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic input already resident in
HBM: the thresholds kernel over the baseline series and the metrics kernel over the measure
series of every grid cell a rank owns.  Grid cells are independent, so ranks shard them with
no data-path collective (weak scaling: every rank processes the full per-GPU workload); the
RCCL all-gather that reassembles the metrics Dataset is exercised and reported separately
(`allgather`), never inside `value`.

torch is used only as plumbing here (process group, barrier, the all-gather, device buffers
whose raw pointers go to the C ABI).  The kernels are launched on torch's current stream and
timed there with HIP events (hdp_event_*), which is what `roofline.achieved` is computed from.
"""
import argparse
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
    "c3": (100, 720, 1440, 1, PERCENTILES, DEFINITIONS),  # BASELINE.json configs[2], the config the target is quoted on
    "c2": (10, 180, 360, 1, PERCENTILES, DEFINITIONS),    # configs[1]: 3650 d x 180 x 360
    # configs[4]: 10 members x 36500 d x 192 x 288; members are concatenated along time for the thresholds
    # (threshold.py:114-119) and share their cell's thresholds in the metrics pass
    "c5": (100, 192, 288, 10, PERCENTILES_C5, DEFINITIONS_C5),
    "tiny": (10, 16, 32, 1, PERCENTILES, DEFINITIONS),
    "tiny5": (10, 8, 16, 3, PERCENTILES_C5, DEFINITIONS_C5),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default=os.environ.get("HDP_BENCH_CONFIG", "c3"), choices=sorted(CONFIGS))
    ap.add_argument("--cells", type=int, default=0, help="override the number of grid cells per rank (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline sample time")
    ap.add_argument("--mem-fraction", type=float, default=0.88)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo only to rehearse the N>1 code path on a one-GPU box")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal: every rank uses GPU 0 (needs --backend gloo and a small --cells)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
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

    lib = _lib.ensure_device(local_rank)
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.current_stream(dev).cuda_stream

    years, n_lat, n_lon, M, PERCENTILES, DEFINITIONS = CONFIGS[args.config]
    T = years * 365
    cells_rank = args.cells if args.cells > 0 else n_lat * n_lon   # weak scaling: full grid per rank
    P, D = PERCENTILES.size, len(DEFINITIONS)

    # ---- host tables (the reference builds the same ones in Python: threshold.py:125, metric.py:410-416)
    dates = utils.noleap_date_range("2000-01-01", f"{2000 + years - 1}-12-31")
    time_index, cols = cal.window_columns(np.concatenate([dates] * M), 7)   # [n_doy, M * years]
    doy_map = cal.build_doy_map(dates)
    north, south, season_years = cal.hemisphere_season_tables(dates)
    Y = north.shape[0]
    n_doy = time_index.shape[0]
    tplan = core.ThresholdPlan(time_index, cols, PERCENTILES, M * T)
    mplan = core.MetricsPlan(doy_map, n_doy, DEFINITIONS, north, south, P)
    Yp = mplan.year_pitch

    # ---- bands: the largest equal split of this rank's cells whose buffers fit in HBM ----------------
    per_cell = 2 * M * T * 4 + n_doy * P * 8 + M * (4 * P * D * Yp * 2 + 1) + 4
    free_b, total_b = torch.cuda.mem_get_info(dev)
    budget = int(free_b * args.mem_fraction)
    n_bands = 1
    while (cells_rank + n_bands - 1) // n_bands * per_cell > budget:
        n_bands += 1
    bc = (cells_rank + n_bands - 1) // n_bands
    cells_rank_eff = bc * n_bands   # equal bands (== cells_rank for the named configs)

    def raw(nbytes):
        return torch.empty(int(nbytes), dtype=torch.uint8, device=dev)

    # baseline [cell][M * T] (members appended along time); measure [member][cell][T] = M * bc series
    xb, xm = raw(bc * M * T * 4), raw(M * bc * T * 4)
    thr = raw(bc * n_doy * P * 8)
    out = torch.empty(4 * P * D * M * bc * Yp, dtype=torch.int16, device=dev)
    # latitude of every cell of this rank's grid, row-major (lat, lon); band b owns cells [b*bc, (b+1)*bc)
    lat_axis = np.linspace(-90.0, 90.0, n_lat)
    cell_ids = np.arange(cells_rank_eff) % (n_lat * n_lon)
    lat_cells = lat_axis[cell_ids // n_lon].astype(np.float32)
    lat_dev = torch.from_numpy(np.tile(lat_cells[:bc], M)).to(dev)
    south_dev = [torch.from_numpy(np.tile((lat_cells[b * bc:(b + 1) * bc] < 0).astype(np.uint8), M)).to(dev)
                 for b in range(n_bands)]

    # synthetic inputs, generated on the device (reference generator formula + hashed noise):
    # baseline = control, measure = control + warming trend (hdp/utils.py:41: t / (365*100))
    seed = 0
    _lib.check(lib.hdp_generate_series_dev(xb.data_ptr(), bc, M * T, rank * cells_rank_eff, lat_dev.data_ptr(),
                                           seed, 0.7, 0.0, stream))
    _lib.check(lib.hdp_generate_series_dev(xm.data_ptr(), M * bc, T, rank * cells_rank_eff * M, lat_dev.data_ptr(),
                                           seed + 1, 0.7, 1.0 / 36500.0, stream))
    torch.cuda.synchronize(dev)

    import ctypes
    n_ev_steps = args.steps
    mplan.reserve(M * bc)   # exceedance scratch allocated before anything is timed
    ev = [[[lib.hdp_event_create() for _ in range(3)] for _ in range(n_bands)] for _ in range(n_ev_steps)]
    t_thr, t_met = [], []

    def step(k):
        """k >= 0: timed step k (events recorded, read back after the timed region); k < 0: warm-up"""
        for b in range(n_bands):
            if k >= 0:
                lib.hdp_event_record(ev[k][b][0], stream)
            tplan.run(xb.data_ptr(), bc, thr.data_ptr(), stream)
            if k >= 0:
                lib.hdp_event_record(ev[k][b][1], stream)
            mplan.run(xm.data_ptr(), thr.data_ptr(), bc, south_dev[b].data_ptr(), M * bc, out.data_ptr(), stream)
            if k >= 0:
                lib.hdp_event_record(ev[k][b][2], stream)

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
    elapsed = time.perf_counter() - t0
    ms = ctypes.c_float()
    for k in range(args.steps):
        for b in range(n_bands):
            e0, e1, e2 = ev[k][b]
            _lib.check(lib.hdp_event_elapsed_ms(e0, e1, ctypes.byref(ms))); t_thr.append(float(ms.value))
            _lib.check(lib.hdp_event_elapsed_ms(e1, e2, ctypes.byref(ms))); t_met.append(float(ms.value))
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed * 1e3 / max(1, args.steps)
    cell_days_step = 2.0 * cells_rank_eff * M * T * world   # thresholds pass + metrics pass, all ranks
    value = cell_days_step / (ms_per_step * 1e-3)

    # ---- roofline of the dominant kernel: algorithmic bytes per launch / measured launch time --------
    bytes_thr = bc * (4 * M * T + 8 * n_doy * P)
    bytes_met = bc * (M * 4 * T + 8 * n_doy * P + M * 2 * 4 * Y * P * D)  # int16 metrics; thresholds shared by members
    ms_thr, ms_met = float(np.mean(t_thr)), float(np.mean(t_met))
    kern = {
        "thresholds_kernel": {"ms_per_launch": ms_thr, "algorithmic_bytes": bytes_thr,
                              "GBps": bytes_thr / ms_thr / 1e6, "frac_hbm": bytes_thr / ms_thr / 1e6 / HBM_PEAK_GBS,
                              "cell_days_per_s": bc * M * T / (ms_thr * 1e-3)},
        "metrics_kernel": {"ms_per_launch": ms_met, "algorithmic_bytes": bytes_met,
                           "GBps": bytes_met / ms_met / 1e6, "frac_hbm": bytes_met / ms_met / 1e6 / HBM_PEAK_GBS,
                           "cell_days_per_s": bc * M * T / (ms_met * 1e-3)},
    }
    dom = "thresholds_kernel" if ms_thr >= ms_met else "metrics_kernel"
    # HBM traffic of the dominant kernel: PMC counters cannot be collected from inside this process,
    # so the per-cell figure measured with rocprofv3 (profiles/traffic_per_cell.json, same workload
    # shape) is scaled to the cells of one launch; null when no profile matches this workload.
    traffic = None
    try:
        tp = json.load(open(os.path.join(ROOT, "profiles", "traffic_per_cell.json")))
        if tp.get("workload") == args.config:
            traffic = float(tp[dom]["bytes_per_cell"]) * bc
    except Exception:
        traffic = None
    dom_name = {"thresholds_kernel": tplan.describe() + ", one launch per band",
                "metrics_kernel": "exceed_kernel + metrics_kernel_cells16 (batches of series, two overlapping launches per batch, timed together)"}[dom]
    roofline = {"bound": "hbm", "kernel": dom_name, "achieved": kern[dom]["GBps"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": kern[dom]["frac_hbm"], "traffic": traffic,
                "traffic_source": "rocprofv3 FETCH_SIZE/WRITE_SIZE per cell (profiles/) x cells per launch",
                "both_kernels_frac": (bytes_thr + bytes_met) / (ms_thr + ms_met) / 1e6 / HBM_PEAK_GBS,
                "measured_copy_ceiling": HBM_COPY_GBS}

    # ---- pre-step: fused Celsius -> heat index -> Celsius kernel, reported beside `value`
    pre_step = None
    if rank == 0:
        try:
            n_el = int(min(bc * T, thr.numel() // 4, out.numel() // 2, 1 << 30)) & ~3
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

    # ---- post-step: latitude-weighted spatial mean of the resident int16 metrics, one row per
    # (metric, percentile, definition, season) over this band's series; reported beside `value`
    post_step = None
    if rank == 0:
        try:
            n_rows, n_ser = 4 * P * D * int(Yp), M * bc
            w_dev = torch.from_numpy(np.cos(np.deg2rad(np.tile(lat_cells[:bc], M).astype(np.float64)))).to(dev)
            mean_dev = torch.empty(n_rows, dtype=torch.float64, device=dev)
            e0, e1 = lib.hdp_event_create(), lib.hdp_event_create()
            _lib.check(lib.hdp_weighted_mean_i16_dev(out.data_ptr(), n_rows, n_ser, w_dev.data_ptr(), mean_dev.data_ptr(), stream))
            lib.hdp_event_record(e0, stream)
            reps = 3
            for _ in range(reps):
                _lib.check(lib.hdp_weighted_mean_i16_dev(out.data_ptr(), n_rows, n_ser, w_dev.data_ptr(), mean_dev.data_ptr(), stream))
            lib.hdp_event_record(e1, stream)
            wms = ctypes.c_float()
            _lib.check(lib.hdp_event_elapsed_ms(e0, e1, ctypes.byref(wms)))
            gbps = (2.0 * n_rows * n_ser + 8.0 * n_ser + 8.0 * n_rows) * reps / (wms.value * 1e-3) / 1e9
            post_step = {"kernel": "weighted_rows_mean_i16x8_kernel", "rows": n_rows, "series": n_ser,
                         "ms": wms.value / reps, "GBps": gbps, "frac_hbm": gbps / HBM_PEAK_GBS}
            del w_dev, mean_dev
        except Exception as e:
            post_step = {"error": f"{type(e).__name__}: {e}"[:300]}

    # ---- RCCL all-gather of the metrics (reassembly step of north_star), reported separately ------------
    allgather = None
    if world > 1:
        try:
            total_c4 = 4 * P * D * (n_lat * n_lon) * Yp              # the C3/C4 grid's int16 metrics, elements
            share = min(out.numel(), total_c4 // 8)                   # the per-GPU shard of config 4
            free_b, _ = torch.cuda.mem_get_info(dev)
            share = int(min(share, free_b * 0.8 / 2 / world))
            if args.backend != "nccl":   # rehearsal path: gloo moves host bytes
                share = min(share, 1 << 24)
            share &= ~3                                               # whole 8-byte words
            gdev = dev if args.backend == "nccl" else torch.device("cpu")
            gathered = torch.empty(share * world, dtype=torch.int16, device=gdev)
            # RCCL has no int16 type: the bytes travel as int64 words (also keeps the element count of a 6 GB shard
            # well inside 32 bits)
            g8, o8 = gathered.view(torch.int64), out[:share].to(gdev).view(torch.int64)
            dist.all_gather_into_tensor(g8, o8)
            fence()
            t1 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                dist.all_gather_into_tensor(g8, o8)
            fence()
            dt = (time.perf_counter() - t1) / reps
            allgather = {"bytes_per_rank": share * 2, "ms": dt * 1e3,
                         "recv_GBps_per_gpu": share * 2 * (world - 1) / dt / 1e9}
            del gathered
        except Exception as e:   # the reassembly step is reported beside `value`; it must not cost the bench line
            allgather = {"error": f"{type(e).__name__}: {e}"[:300]}

    # ---- CPU baseline: the C restatement of the reference algorithm on a bounded sample, rank 0 only ----
    cpu = None
    parity = None
    if rank == 0 and not args.no_cpu_baseline:
        try:
            from oracle import c_oracle
            cores = c_oracle.max_threads()
            win = cal.expand_window_table(time_index, cols)
            xb_h = lambda n: xb[: n * M * T * 4].cpu().numpy().view(np.float32).reshape(n, M * T)  # noqa: E731
            # the M members of the first n cells, member-major like the device buffer
            xm_h = lambda n: (xm.view(torch.float32).view(M, bc, T)[:, :n].cpu().numpy()  # noqa: E731
                              .reshape(M * n, T))
            thr_m = lambda th: np.concatenate([th] * M)                                     # noqa: E731
            hemi_m = lambda n: np.tile((lat_cells[:n] < 0).astype(np.uint8), M)             # noqa: E731
            n0 = min(bc, cores)
            tc = time.perf_counter()
            th0 = c_oracle.thresholds(xb_h(n0), win, PERCENTILES)
            c_oracle.metrics(xm_h(n0), thr_m(th0), doy_map, DEFINITIONS, north, south, hemi_m(n0))
            per_round = time.perf_counter() - tc
            rounds = int(max(1, min(64, args.cpu_seconds / max(per_round, 1e-3))))
            ns = min(bc, n0 * rounds)
            tc = time.perf_counter()
            th_cpu = c_oracle.thresholds(xb_h(ns), win, PERCENTILES)
            met_cpu = c_oracle.metrics(xm_h(ns), thr_m(th_cpu), doy_map, DEFINITIONS, north, south, hemi_m(ns))
            cpu_s = time.perf_counter() - tc
            cpu = None if world > 1 else {   # reported at N = 1 only (torchrun pins OMP_NUM_THREADS=1)
                "value": 2.0 * ns * M * T / cpu_s, "unit": "cell-days/s", "cores": cores, "kind": "port",
                   "sample": f"first {ns} cells of band 0 of the same workload (T={T}, P={P}, D={D}), both passes, "
                          f"{cpu_s:.1f} s; oracle/hdp_oracle.c (reference algorithm restated in C, OpenMP over cells)"}
            # the same sample doubles as a parity spot-check of what the timed kernels produced (band 0 flags)
            tplan.run(xb.data_ptr(), bc, thr.data_ptr(), stream)
            mplan.run(xm.data_ptr(), thr.data_ptr(), bc, south_dev[0].data_ptr(), M * bc, out.data_ptr(), stream)
            torch.cuda.synchronize(dev)
            # device layout [cell][P][n_doy] -> the reference's (cell, doy, percentile)
            th_gpu = thr[: ns * n_doy * P * 8].cpu().numpy().view(np.float64).reshape(ns, P, n_doy).transpose(0, 2, 1)
            # device layout [4][P][D][Y][series] -> the reference's (percentile, definition, series, metric, year)
            out_gpu = out.view(4, P * D, Y, M, bc)[..., :ns].cpu().numpy()
            met_gpu = np.transpose(out_gpu.reshape(4, P, D, Y, M * ns), (1, 2, 4, 0, 3)).astype(np.int64)
            parity = {"cells": ns, "thresholds_bit_exact": bool(np.array_equal(th_gpu, th_cpu)),
                      "metrics_bit_exact": bool(np.array_equal(met_gpu, met_cpu))}
        except Exception as e:   # the checker must not cost the bench line
            parity = {"error": f"{type(e).__name__}: {e}"[:300]}

    if rank == 0:
        line = {
            "metric": "grid-cell-days/sec for compute_thresholds+compute_group_metrics",
            "value": value, "unit": "cell-days/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {f'{M} members x ' if M > 1 else ''}{T} d x {n_lat} x {n_lon} fp32, {P} percentiles x {D} definitions, "
                                   f"window radius 7, noleap, per GPU",
                       "cells_per_gpu": int(cells_rank_eff), "members": M, "T": T, "percentiles": P, "definitions": D,
                       "seasons": int(Y), "resident_bands_per_step": n_bands, "cells_per_band": int(bc),
                       "sharding": "independent grid cells per rank, no data-path collective"},
            "roofline": roofline, "cpu_baseline": cpu, "kernels": kern, "pre_step": pre_step, "post_step": post_step, "allgather": allgather,
            "parity_sample": parity, "device": _lib.device_info(),
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
