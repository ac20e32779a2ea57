"""Synthetic inputs and the small calendar type used by tests, bench and smoke.

Mirrors the reference's generators (hdp/utils.py:39-92: sinusoidal seasonal cycle,
hemisphere phase, latitude gradient, optional seeded noise, optional warming trend) as
plain numpy, plus a noleap date carrying the attributes the hot path reads from cftime
objects (``dayofyr``, ``year``, ``month``, ``day``, ``calendar``).
"""
from __future__ import annotations

import datetime as _dt
import time as _time

import numpy as np


_MONTH_LEN = (31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31)
_MONTH_START = tuple(int(v) for v in np.concatenate([[0], np.cumsum(_MONTH_LEN)[:-1]]))


class NoLeapDate:
    """Minimal stand-in for ``cftime.DatetimeNoLeap`` (attributes only)."""
    __slots__ = ("year", "month", "day")
    calendar = "noleap"

    def __init__(self, year, month, day):
        self.year, self.month, self.day = int(year), int(month), int(day)

    @property
    def dayofyr(self):
        return _MONTH_START[self.month - 1] + self.day

    def __str__(self):
        return f"{self.year:04d}-{self.month:02d}-{self.day:02d} 00:00:00"

    __repr__ = __str__

    def __eq__(self, other):
        return (self.year, self.month, self.day) == (other.year, other.month, other.day)

    def __hash__(self):
        return hash((self.year, self.month, self.day))


def _parse(s):
    parts = [int(p) for p in str(s).split("-")]
    return parts + [1] * (3 - len(parts))


def noleap_date_range(start, end):
    """Daily noleap dates from start to end inclusive ("YYYY[-MM[-DD]]")."""
    y0, m0, d0 = _parse(start)
    y1, m1, d1 = _parse(end)
    first = y0 * 365 + _MONTH_START[m0 - 1] + d0 - 1
    last = y1 * 365 + _MONTH_START[m1 - 1] + d1 - 1
    ordinal = np.arange(first, last + 1)
    year, doy0 = ordinal // 365, ordinal % 365
    month = np.searchsorted(np.asarray(_MONTH_START), doy0, side="right")
    day = doy0 - np.asarray(_MONTH_START)[month - 1] + 1
    out = np.empty(ordinal.size, dtype=object)
    for i in range(ordinal.size):
        out[i] = NoLeapDate(year[i], month[i], day[i])
    return out


def generate_control_array(start_date="1700-01-01", end_date="1749-12-31", grid_shape=(2, 3),
                           add_noise=False, seed=0):
    """hdp/utils.py:53-92 -> (float64 [lon, lat, time], lon, lat, dates)."""
    dates = noleap_date_range(start_date, end_date)
    t = np.arange(dates.size, dtype=float)
    n_lon, n_lat = grid_shape
    vals = np.empty((n_lon, n_lat, t.size))
    vals[:, n_lat // 2:, :] = 20 + 2 * np.sin(2 * np.pi * ((270 + t) / 365))
    vals[:, : n_lat // 2, :] = 20 + 2 * np.sin(2 * np.pi * ((90 + t) / 365))
    if add_noise:
        np.random.seed(seed)
        vals += np.random.random(vals.shape) * (np.std(vals) / 2)
    lat = np.linspace(-90, 90, n_lat, dtype=float)
    lon = np.linspace(-180, 180, n_lon, dtype=float)
    vals = vals - 10 * (np.abs(lat) / 90)[None, :, None]
    return vals, lon, lat, dates


def generate_warming_array(start_date="2000-01-01", end_date="2049-12-31", grid_shape=(2, 3),
                           warming_period=100, add_noise=False):
    """hdp/utils.py:39-42."""
    vals, lon, lat, dates = generate_control_array(start_date, end_date, grid_shape, add_noise)
    return vals + (np.arange(dates.size) / (365 * warming_period))[None, None, :], lon, lat, dates


def get_time_stamp():
    return _dt.datetime.fromtimestamp(_time.time()).strftime("%Y-%m-%d %H:%M")


def get_version():
    from . import __version__
    return __version__


def add_history(obj, msg):
    """hdp/utils.py:14-20 (provenance string in attrs["history"])."""
    if "history" not in obj.attrs:
        obj.attrs["history"] = f"({get_time_stamp()}) History metadata initialized by HDP v{get_version()}.\n"
    obj.attrs["history"] += f"({get_time_stamp()}) {msg}\n"
    return obj
