#!/usr/bin/env python
"""Statistics behind DESIGN.md 5.1 ("sliding selection"): how far the split of a requested rank moves from one
day-of-year row to the next on the bench's kind of data (reference generator: 20 + 2 sin(2 pi (beta + t) / 365) + 0.7 u,
100 years, window radius 7), and on flat data (no seasonal cycle).

For row d and rank R (0-based from the top of the window's 1500 samples) let v be the R-th largest sample.  Going to row
d + 1 one column leaves and one enters; delta = (samples of the entering column above v) - (samples of the leaving column
above v) is the number of single-element moves a carried split needs.  Printed: standard deviation and extremes of delta over
all rows, and the mean over groups of 64 rows of max |delta| (lanes of a wave move in lock step).  CPU only; numpy."""
import numpy as np


def stats(x, label, years=100, radius=7):
    cols = x.reshape(years, 365).T                    # [doy][year]
    srt = -np.sort(-cols, axis=1)                     # columns sorted descending
    W = 2 * radius + 1
    out = {}
    for R in (15, 150, 750):
        deltas = []
        for d in range(radius, 365 - radius - 1):
            win = np.sort(cols[d - radius:d + radius + 1].ravel())[::-1]
            v = win[R]
            leave, enter = srt[d - radius], srt[d + radius + 1]
            deltas.append(int((enter > v).sum()) - int((leave > v).sum()))
        dl = np.array(deltas)
        grp = [np.abs(dl[i:i + 64]).max() for i in range(0, len(dl) - 63, 64)]
        out[R] = (dl.std(), dl.min(), dl.max(), float(np.mean(grp)))
        print(f"{label:28s} rank {R:4d}: std(delta) {dl.std():5.2f}  min {dl.min():4d}  max {dl.max():4d}  "
              f"mean over 64-row groups of max|delta| {np.mean(grp):5.1f}")
    return out


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    T = 36500
    t = np.arange(T)
    noise = 0.7 * rng.random(T)
    stats(20 + 2 * np.sin(2 * np.pi * (270 + t) / 365.0) + noise, "generator (seasonal + 0.7 u)")
    stats(20 + noise, "flat (0.7 u only)")
    stats(20 + 8 * np.sin(2 * np.pi * t / 365.0) + rng.normal(0, 3.0, T), "8 sin + N(0, 3) (mid-latitude)")
