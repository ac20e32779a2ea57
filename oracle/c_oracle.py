"""ctypes wrapper of oracle/libhdp_oracle.so (TEST INFRASTRUCTURE ONLY; see hdp_oracle.c)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libhdp_oracle.so")
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE, "libhdp_oracle.so"], check=True, capture_output=True)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            build()
        _lib = C.CDLL(_PATH)
        _lib.oracle_max_threads.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def max_threads():
    return int(load().oracle_max_threads())


def set_threads(n):
    """OpenMP threads of the following calls; returns the count in effect."""
    lib = load()
    lib.oracle_set_threads.restype = C.c_int
    return int(lib.oracle_set_threads(C.c_int(int(n))))


def thresholds(x, win, q):
    lib = load()
    x = np.ascontiguousarray(x, dtype=np.float32)
    win = np.ascontiguousarray(win, dtype=np.int64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    out = np.empty((x.shape[0], win.shape[0], q.size))
    lib.oracle_thresholds(_p(x), C.c_int64(x.shape[0]), C.c_int64(x.shape[1]), _p(win),
                          C.c_int64(win.shape[0]), C.c_int64(win.shape[1]), _p(q), C.c_int64(q.size), _p(out))
    return out


def metrics(x, thr, doy_map, defs, north, south, is_south):
    lib = load()
    x = np.ascontiguousarray(x, dtype=np.float32)
    thr = np.ascontiguousarray(thr, dtype=np.float64)
    dm = np.ascontiguousarray(doy_map, dtype=np.int64)
    defs = np.ascontiguousarray(np.asarray(defs, dtype=np.int64).reshape(-1, 3))
    north = np.ascontiguousarray(north, dtype=np.int64)
    south = np.ascontiguousarray(south, dtype=np.int64)
    hemi = np.ascontiguousarray(is_south, dtype=np.uint8)
    n, T = x.shape
    n_thr, n_doy, P = thr.shape
    D, Y = defs.shape[0], north.shape[0]
    out = np.zeros((P, D, n, 4, Y), dtype=np.int64)
    i64 = C.c_int64
    lib.oracle_metrics(_p(x), i64(n), i64(T), _p(thr), i64(n_thr), i64(n_doy), i64(P), _p(dm), _p(defs),
                       i64(D), _p(north), _p(south), _p(hemi), i64(Y), _p(out))
    return out
