"""ctypes binding of libhdp_hip.so (include/hdp_hip.h).

There is no CPU fallback: if the shared library is missing, or no HIP device can
be initialised, the first compute call raises.  ``load()`` alone (symbol check)
works without a GPU.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# HDP_LIB_PATH: an instrumented build of the same library (make EXTRA=-DHDP_DEBUG_ABLATIONS LIB=...), development only
LIB_PATH = os.environ.get("HDP_LIB_PATH") or os.path.join(_HERE, "libhdp_hip.so")

HDP_OK = 0
ERROR_NAMES = {-1: "HDP_EINVAL", -2: "HDP_ENODEV", -3: "HDP_EHIP", -4: "HDP_ENOMEM",
               -5: "HDP_EUNSUP", -6: "HDP_EQUANT"}


class HdpError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"{ERROR_NAMES.get(code, code)}: {message}")
        self.code = code
        self.message = message


i64, i32, f64, f32, vp, sz = C.c_int64, C.c_int32, C.c_double, C.c_float, C.c_void_p, C.c_size_t
P = C.POINTER

# name -> (restype, argtypes); mirrors include/hdp_hip.h one to one
SIGNATURES = {
    "hdp_init": (C.c_int, [C.c_int]),
    "hdp_shutdown": (C.c_int, []),
    "hdp_device_count": (C.c_int, []),
    "hdp_last_error": (C.c_char_p, []),
    "hdp_device_info": (C.c_char_p, []),
    "hdp_dev_alloc": (vp, [sz]),
    "hdp_dev_free": (C.c_int, [vp]),
    "hdp_memcpy_h2d": (C.c_int, [vp, vp, sz]),
    "hdp_memcpy_d2h": (C.c_int, [vp, vp, sz]),
    "hdp_dev_memset": (C.c_int, [vp, C.c_int, sz]),
    "hdp_sync": (C.c_int, [vp]),
    "hdp_event_create": (vp, []),
    "hdp_event_record": (C.c_int, [vp, vp]),
    "hdp_event_elapsed_ms": (C.c_int, [vp, vp, P(f32)]),
    "hdp_event_destroy": (C.c_int, [vp]),
    "hdp_threshold_plan_create": (C.c_int, [vp, i64, i64, vp, i64, vp, i64, i64, P(vp)]),
    "hdp_threshold_plan_destroy": (C.c_int, [vp]),
    "hdp_threshold_plan_reserve": (C.c_int, [vp, i64, C.c_int]),
    "hdp_threshold_plan_describe": (C.c_char_p, [vp]),
    "hdp_thresholds_f32_dev": (C.c_int, [vp, vp, i64, vp, vp]),
    "hdp_thresholds_f32": (C.c_int, [vp, i64, i64, i64, i64, vp, i64, i64, vp, i64, vp, i64, vp]),
    "hdp_percentiles_table_f32": (C.c_int, [vp, i64, i64, i64, i64, vp, i64, i64, vp, i64, vp]),
    "hdp_metrics_plan_create": (C.c_int, [vp, i64, i64, vp, i64, vp, vp, i64, i64, P(vp)]),
    "hdp_metrics_plan_destroy": (C.c_int, [vp]),
    "hdp_metrics_year_pitch": (i64, [vp]),
    "hdp_metrics_plan_reserve": (C.c_int, [vp, i64]),
    "hdp_metrics_plan_batch_cells": (i64, [vp, i64]),
    "hdp_metrics_f32_dev": (C.c_int, [vp, vp, vp, i64, vp, i64, vp, vp]),
    "hdp_metrics_f32": (C.c_int, [vp, i64, i64, i64, i64, vp, i64, i64, i64, vp, vp, i64, vp, vp, vp, i64, vp]),
    "hdp_metrics_f32_planes_i64": (C.c_int, [vp, i64, i64, i64, i64, vp, i64, i64, i64, vp, vp, i64, vp, vp, vp, i64, vp]),
    "hdp_metrics_f32_layout_i16": (C.c_int, [vp, i64, i64, i64, i64, vp, i64, i64, i64, vp, vp, i64, vp, vp, vp, i64, vp]),
    "hdp_metrics_f32_planes_i64_sharded": (C.c_int, [vp, i64, i64, i64, i64, i64, vp, i64, i64, vp, vp, i64, vp, vp, vp,
                                                     i64, i64, vp, P(i64)]),
    "hdp_metrics_planes_i64_regroup": (C.c_int, [vp, i64, i64, i64, i64, i64, i64, vp]),
    "hdp_index_heatwaves": (C.c_int, [vp, i64, i64, i64, i64, i64, vp]),
    "hdp_season_metrics": (C.c_int, [vp, i64, i64, vp, i64, vp, vp]),
    "hdp_indicate_hot_days": (C.c_int, [vp, i64, i64, vp, i64, vp, vp]),
    "hdp_heat_index_f32": (C.c_int, [vp, vp, i64, vp]),
    "hdp_heat_index_f32_dev": (C.c_int, [vp, vp, i64, vp, vp]),
    "hdp_heat_index_celsius_f32_dev": (C.c_int, [vp, vp, i64, vp, vp]),
    "hdp_weighted_mean_i16_dev": (C.c_int, [vp, i64, i64, vp, vp, vp]),
    "hdp_weighted_mean_f64": (C.c_int, [vp, i64, i64, vp, vp]),
    "hdp_generate_series_dev": (C.c_int, [vp, i64, i64, i64, vp, C.c_uint64, f32, f32, vp]),
    "hdp_metrics_plan_describe": (C.c_char_p, [vp]),
    "hdp_thresholds_f32_tm_dev": (C.c_int, [vp, vp, i64, i64, vp, vp]),
    "hdp_metrics_f32_tm_dev": (C.c_int, [vp, vp, i64, vp, i64, vp, i64, vp, vp]),
    "hdp_comm_unique_id": (C.c_int, [vp]),
    "hdp_comm_init_rank": (C.c_int, [vp, C.c_int, C.c_int]),
    "hdp_comm_destroy": (C.c_int, []),
    "hdp_rccl_version": (C.c_int, [P(C.c_int)]),
    "hdp_comm_rank": (C.c_int, []),
    "hdp_comm_world": (C.c_int, []),
    "hdp_allgather_dev": (C.c_int, [vp, sz, vp, vp]),
    "hdp_allgather_direct_dev": (C.c_int, [vp, sz, vp, vp]),
}

_lib = None
_lock = threading.Lock()
_initialised_device = None


def _share_hip_runtime():
    """One HIP runtime per process, whatever the import order.

    PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 under the same sonames as the system ROCm.  If
    libhdp_hip.so is loaded first it pulls in the system copies, and a later ``import torch`` then mixes them with its
    bundled ones and reports "No HIP GPUs are available"; loaded after torch, libhdp_hip.so simply binds to the copies
    torch already mapped, and both sides share one runtime.  So when torch is installed but not imported yet, map
    ITS copies first -- the state the torch-first order produces -- without importing torch.
    ``HDP_HIP_RUNTIME=system`` skips this (the process must then never import torch after hdp_amd);
    returns which runtime the library will bind to, for device_info() and error messages."""
    import sys
    if os.environ.get("HDP_HIP_RUNTIME", "").lower() == "system":
        return "system"
    if "torch" in sys.modules:
        return "torch (already imported)"
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return "system"
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    mapped = []
    # NOT librccl: libhdp_hip.so NEEDs librccl.so.1 and the wheel bundles one under the same soname, so whichever copy is
    # mapped first serves both the library's communicator and a later torch.distributed nccl backend -- but mapping the
    # wheel's copy from here, ahead of the rest of torch's libraries, ends the process in "double free or corruption"
    # (measured, round 4).  A process that wants torch's RCCL imports torch first (bench.py does); runtime_report() says
    # which copy a run had.
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
                mapped.append(name)
            except OSError:
                return "system"
    return f"torch's bundled copies ({', '.join(mapped)})" if mapped else "system"


hip_runtime = None   # set by load(): which libamdhip64 the library is bound to


def load():
    """dlopen the library and bind every declared symbol (no GPU needed)."""
    global _lib, hip_runtime
    with _lock:
        if _lib is not None:
            return _lib
        hip_runtime = _share_hip_runtime()
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C hdp_amd/csrc` (hipcc --offload-arch=gfx950). hdp_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def check(rc):
    if rc != HDP_OK:
        msg = load().hdp_last_error().decode("utf-8", "replace")
        if rc == -6:
            raise ValueError(msg)  # numba raises ValueError('Quantiles must be in the range [0, 1]')
        if rc == -1 and msg.startswith("zero-size array"):
            raise ValueError(msg)  # what np.max([]) raises inside the reference (metric.py:136)
        raise HdpError(rc, msg)


def ensure_device(device=None):
    """Initialise the HIP device once per process (LOCAL_RANK selects it under torchrun)."""
    global _initialised_device
    lib = load()
    if device is None:
        if _initialised_device is not None:
            return lib
        device = int(os.environ.get("HDP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if _initialised_device != device:
        check(lib.hdp_init(int(device)))
        _initialised_device = device
    return lib


def device_info():
    return ensure_device().hdp_device_info().decode()


def mapped_libraries(names=("librccl", "libamdhip64", "libhsa-runtime64", "libhdp_hip")):
    """Resolved paths of the runtime libraries this process has mapped (/proc/self/maps): which librccl / libamdhip64 /
    libhsa-runtime64 a failing multi-GPU run was using is the first question its log must answer."""
    found = {n: [] for n in names}
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(None, 1)[-1] if "/" in line else ""
                base = os.path.basename(path)
                for n in names:
                    if base.startswith(n + ".so") and path not in found[n]:
                        found[n].append(path)
    except OSError:
        pass
    return found


def runtime_report():
    """{hip_runtime, mapped: {lib: [paths]}, rccl_version}: goes into bench.py's JSON line."""
    lib = load()
    ver = C.c_int(0)
    rc = lib.hdp_rccl_version(C.byref(ver))
    return {"hip_runtime": hip_runtime, "mapped": mapped_libraries(), "rccl_version": int(ver.value) if rc == HDP_OK else None}
