#!/usr/bin/env python
"""Generate the golden fixtures in this directory from the reference itself.

Runs ONLY in the build container (needs /root/reference); the GPU box and the
test-suite consume the committed .npz files, never this script's imports.

How the reference is executed here:
  * ``hdp/metric.py`` imports unmodified once ``numba`` (identity decorators),
    ``xarray``, ``cftime``, ``dask``, ``tqdm`` are present as empty stub
    modules in ``sys.modules`` -- its njit functions then run as the plain
    Python/NumPy they are written in.
  * ``hdp/threshold.py`` does not parse on Python 3.10 (nested-quote f-string at
    line 173), so the two functions that precede that line --
    ``datetimes_to_windows`` (12-49) and the body of ``compute_percentiles``
    (59-78) -- are compiled from their source line ranges, read at run time.
    Under the stubs ``np.quantile`` is NumPy's, not Numba's: the stored
    thresholds pin the gather/window logic exactly and the quantile arithmetic
    to ~1 ulp (the oracle follows Numba's formula; tolerance 1e-6 relative as
    north_star states, observed ~1e-15).

No reference source text is written to disk; the fixtures hold inputs and
outputs only.
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))


def install_stubs():
    ident = lambda *a, **k: (a[0] if (len(a) == 1 and callable(a[0]) and not k) else (lambda f: f))
    nb = types.ModuleType("numba")
    nb.njit = ident
    nb.jit = ident
    nb.vectorize = ident
    nb.guvectorize = ident
    class _T:  # usable as a NumPy dtype (metric.py: dtype=nb.int64) and callable as a signature (measure.py:61)
        def __init__(self, dt):
            self.dtype = np.dtype(dt)

        def __call__(self, *args):
            return ("signature", self.dtype, args)

    nb.int64 = _T(np.int64)
    nb.float64 = _T(np.float64)
    nb.float32 = _T(np.float32)
    nb.boolean = _T(np.bool_)
    sys.modules["numba"] = nb
    xr = types.ModuleType("xarray")
    xr.DataArray = type("DataArray", (), {})
    xr.Dataset = type("Dataset", (), {})
    sys.modules["xarray"] = xr
    for name in ("cftime", "dask", "dask.array", "tqdm", "tqdm.auto"):
        m = types.ModuleType(name)
        sys.modules[name] = m
    sys.modules["tqdm.auto"].tqdm = lambda x, **k: x
    sys.modules["dask"].array = sys.modules["dask.array"]


def load_reference():
    install_stubs()
    sys.path.insert(0, REF)
    import hdp.metric as ref_metric  # noqa: E402
    src = open(os.path.join(REF, "hdp", "threshold.py")).read().split("\n")
    ns = {"np": np}
    exec(compile("\n".join(src[11:49]), "threshold.py[12:49]", "exec"), ns)
    body = "def compute_percentiles(temperatures, window_samples, percentiles, output):\n" + \
        "\n".join(src[73:78]) + "\n"
    exec(compile(body, "threshold.py[74:78]", "exec"), ns)
    return ref_metric, ns["datetimes_to_windows"], ns["compute_percentiles"]


def main():
    from oracle import hdp_oracle as orc  # only for the date class + generators
    ref_metric, ref_windows, ref_percentiles = load_reference()
    rng = np.random.default_rng(20261004)

    # ---- (1) index_heatwaves + season metrics on random series ---------------
    hots, defs, ids_out, offs = [], [], [], [0]
    seas, seas_off, hwf, hwn, hwd, hwa = [], [0], [], [], [], []
    for case in range(400):
        n = int(rng.integers(1, 160))
        p_hot = rng.choice([0.1, 0.3, 0.5, 0.8])
        hot = rng.random(n) < p_hot
        d = [int(rng.integers(0, 6)), int(rng.integers(0, 4)), int(rng.integers(0, 4))]
        ids = ref_metric.index_heatwaves(hot, d[0], d[1], d[2])
        assert ids.size == n
        # season ranges: sorted disjoint for even cases, arbitrary overlapping for odd
        k = int(rng.integers(1, 5))
        if case % 2 == 0:
            cuts = np.sort(rng.choice(np.arange(n + 1), size=min(2 * k, n + 1), replace=False))
            cuts = cuts[: (cuts.size // 2) * 2]
            ranges = cuts.reshape(-1, 2)
        else:
            a = rng.integers(0, n, size=k)
            b = np.minimum(n, a + rng.integers(1, n + 1, size=k))
            ranges = np.stack([a, b], axis=1)
        ranges = ranges[ranges[:, 1] > ranges[:, 0]]
        if ranges.shape[0] == 0:
            ranges = np.array([[0, n]])
        ranges = ranges.astype(np.int64)
        hots.append(hot.astype(np.uint8))
        defs.append(d)
        ids_out.append(ids.astype(np.int64))
        offs.append(offs[-1] + n)
        seas.append(ranges)
        seas_off.append(seas_off[-1] + ranges.shape[0])
        hwf.append(ref_metric.heatwave_frequency(ids, ranges))
        hwn.append(ref_metric.heatwave_number(ids, ranges))
        hwd.append(ref_metric.heatwave_duration(ids, ranges))
        hwa.append(ref_metric.heatwave_average(ids, ranges))
    np.savez_compressed(
        os.path.join(HERE, "metric_random.npz"),
        hot=np.concatenate(hots), offsets=np.array(offs), definitions=np.array(defs),
        ids=np.concatenate(ids_out), ranges=np.concatenate(seas), range_offsets=np.array(seas_off),
        hwf=np.concatenate(hwf), hwn=np.concatenate(hwn), hwd=np.concatenate(hwd),
        hwa=np.concatenate(hwa))

    # ---- (2) window tables ---------------------------------------------------
    tables = {}
    tables["r7_3yr"] = ref_windows(orc.noleap_date_range("2001-01-01", "2003-12-31"), 7)
    tables["r7_partial_final_year"] = ref_windows(orc.noleap_date_range("2001-01-01", "2003-06-30"), 7)
    tables["r7_midyear_start"] = ref_windows(orc.noleap_date_range("2001-03-15", "2004-03-14"), 7)
    tables["r1_6days"] = ref_windows(orc.noleap_date_range("2001-01-01", "2001-01-06"), 1)
    tables["r2_2yr"] = ref_windows(orc.noleap_date_range("1999-01-01", "2000-12-31"), 2)
    tables["r0_2yr"] = ref_windows(orc.noleap_date_range("1999-01-01", "2000-12-31"), 0)
    np.savez_compressed(os.path.join(HERE, "window_tables.npz"), **tables)

    # ---- (3) C1 workflow: generator defaults, grid (2,3) ------------------------
    out = {}
    percentiles = np.array([0.9, 0.95])
    definitions = [[3, 0, 0]]
    for tag, noise in (("plain", False), ("noise", True)):
        base, lon, lat, bdates = orc.generate_control(add_noise=noise)
        meas, _, _, mdates = orc.generate_warming(add_noise=noise)
        base32 = base.astype(np.float32)
        meas32 = meas.astype(np.float32)
        win = ref_windows(bdates, 7)
        thr = np.zeros(base.shape[:2] + (win.shape[0], percentiles.size))
        for i in range(base.shape[0]):
            for j in range(base.shape[1]):
                ref_percentiles(base32[i, j], win, percentiles, thr[i, j])
        doy_map = ref_metric.build_doy_map(mdates)
        north = ref_metric.get_range_indices(mdates, (5, 1), (10, 1))
        south = ref_metric.get_range_indices(mdates, (11, 1), (4, 1))
        met = np.zeros((percentiles.size, len(definitions)) + base.shape[:2] + (4, north.shape[0]), dtype=np.int64)
        for p in range(percentiles.size):
            for d, hd in enumerate(definitions):
                for i in range(base.shape[0]):
                    for j in range(base.shape[1]):
                        seasons = south if lat[j] < 0 else north
                        met[p, d, i, j] = ref_metric.compute_heatwave_metrics(
                            meas32[i, j], thr[i, j, :, p], doy_map, hd[0], hd[1], hd[2], seasons)
        out[f"{tag}_thresholds"] = thr
        out[f"{tag}_metrics"] = met
        if tag == "plain":
            out["window_table_rows"] = win[[0, 1, 6, 7, 180, 357, 358, 360, 364]]
            out["window_table_row_ids"] = np.array([0, 1, 6, 7, 180, 357, 358, 360, 364])
            out["window_table_sum_per_row"] = win.sum(axis=1)
            out["doy_map"] = doy_map
            out["north"] = north
            out["south"] = south
            out["lat"] = lat
            out["lon"] = lon
    out["percentiles"] = percentiles
    out["definitions"] = np.array(definitions)
    np.savez_compressed(os.path.join(HERE, "c1_workflow.npz"), **out)

    # ---- (4) small random workflow: 3 years, ragged calendars, 6 definitions -----
    out = {}
    cases = [("full3", "2001-01-01", "2005-12-31"), ("ragged", "2001-01-01", "2004-08-19")]
    percentiles = np.arange(0.9, 1, 0.01)
    definitions = [[3, 0, 0], [3, 1, 1], [4, 2, 0], [4, 1, 3], [5, 0, 1], [5, 1, 4]]
    for tag, s, e in cases:
        dates = orc.noleap_date_range(s, e)
        T = dates.size
        n_cells = 5
        t = np.arange(T)
        base = (15 + 8 * np.sin(2 * np.pi * (t[None, :] - 110) / 365)
                + rng.normal(0, 2.5, size=(n_cells, T))).astype(np.float32)
        meas = (base + rng.normal(0.8, 1.5, size=(n_cells, T))).astype(np.float32)
        win = ref_windows(dates, 7)
        thr = np.zeros((n_cells, win.shape[0], percentiles.size))
        for c in range(n_cells):
            ref_percentiles(base[c], win, percentiles, thr[c])
        doy_map = ref_metric.build_doy_map(dates)
        north = ref_metric.get_range_indices(dates, (5, 1), (10, 1))
        south = ref_metric.get_range_indices(dates, (11, 1), (4, 1))
        # trimming of incomplete years exactly as compute_hemisphere_ranges does is
        # host logic covered separately; here keep rows without -1 in either table
        keep = ~((north == -1).any(axis=1) | (south == -1).any(axis=1))
        north, south = north[keep], south[keep]
        is_south = np.array([0, 1, 0, 1, 1], dtype=np.uint8)
        met = np.zeros((percentiles.size, len(definitions), n_cells, 4, north.shape[0]), dtype=np.int64)
        for p in range(percentiles.size):
            for d, hd in enumerate(definitions):
                for c in range(n_cells):
                    seasons = south if is_south[c] else north
                    met[p, d, c] = ref_metric.compute_heatwave_metrics(
                        meas[c], thr[c, :, p], doy_map, hd[0], hd[1], hd[2], seasons)
        out.update({f"{tag}_baseline": base, f"{tag}_measure": meas, f"{tag}_window": win,
                    f"{tag}_thresholds": thr, f"{tag}_doy_map": doy_map, f"{tag}_north": north,
                    f"{tag}_south": south, f"{tag}_is_south": is_south, f"{tag}_metrics": met,
                    f"{tag}_range": np.array([s, e])})
    out["percentiles"] = percentiles
    out["definitions"] = np.array(definitions)
    np.savez_compressed(os.path.join(HERE, "small_workflow.npz"), **out)

    # ---- (5) season-range tables from the reference's get_range_indices ------------
    out = {}
    for tag, s, e in (("50yr", "2000-01-01", "2049-12-31"), ("midyear", "2001-07-10", "2006-02-03"),
                      ("short", "2001-01-01", "2001-12-31")):
        dates = orc.noleap_date_range(s, e)
        out[f"{tag}_north"] = ref_metric.get_range_indices(dates, (5, 1), (10, 1))
        out[f"{tag}_south"] = ref_metric.get_range_indices(dates, (11, 1), (4, 1))
        out[f"{tag}_doy_map_head"] = ref_metric.build_doy_map(dates)[:400]
        out[f"{tag}_range"] = np.array([s, e])
    np.savez_compressed(os.path.join(HERE, "season_tables.npz"), **out)

    # ---- (6) heat index: the reference ufunc body run as a scalar function ------
    # Under the stubs the arithmetic is NumPy scalar float32 (weak Python-float promotion), not
    # Numba's float64-with-float32-arguments typing: the fixture pins the formula and the branch
    # structure to ~1e-5 relative, not the last bit.
    import importlib
    ref_measure = importlib.import_module("hdp.measure")
    temps = np.linspace(40, 120, 81).astype(np.float32)
    rhs = np.linspace(0, 100, 81).astype(np.float32)
    tg, rg = np.meshgrid(temps, rhs, indexing="ij")
    tt = np.concatenate([tg.ravel(), rng.uniform(60, 115, 400)]).astype(np.float32)
    rr = np.concatenate([rg.ravel(), rng.uniform(0, 100, 400)]).astype(np.float32)
    with np.errstate(all="ignore"):
        hi = np.array([ref_measure.heat_index(np.float32(a), np.float32(b)) for a, b in zip(tt, rr)],
                      dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "heat_index.npz"), temp_f=tt, rel_humid=rr, reference_stub_run=hi)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
