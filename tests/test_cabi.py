"""The C-ABI library loads without a GPU and exports every symbol include/hdp_hip.h declares;
compute entry points fail loudly (no CPU fallback) when no HIP device is usable.  CPU only."""
import os
import re

import numpy as np
import pytest

from hdp_amd import _lib, core

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "hdp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hdp_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 25
    lib = _lib.load()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in hdp_hip.h but not exported by libhdp_hip.so"
    assert sorted(_lib.SIGNATURES) == names, "ctypes signatures out of sync with the header"


def test_no_cpu_fallback_without_device():
    lib = _lib.load()
    if lib.hdp_device_count() > 0:
        pytest.skip("a HIP device is visible; the no-device behaviour cannot be observed here")
    with pytest.raises(_lib.HdpError, match="HDP_ENODEV"):
        core.index_heatwaves(np.zeros(4, dtype=bool), 1, 1, 1)
    x = np.zeros((1, 730), dtype=np.float32)
    with pytest.raises(_lib.HdpError, match="HDP_ENODEV"):
        core.compute_percentiles(x, np.arange(730).reshape(2, 365).T.copy(), np.zeros((365, 1), np.int32), [0.5])


def test_product_package_does_not_import_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "hdp_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src, f"{f} mentions the oracle: the product path must not depend on it"


def test_devbuf_is_empty_after_a_failed_allocation(tmp_path):
    """DevBuf::alloc must not record a size for a failed hipMalloc (the scratch guards are `bytes >= need`)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = tmp_path / "devbuf_invariant"
    subprocess.run([hipcc, "-std=c++17", "-O1", "-I", os.path.join(ROOT, "hdp_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "devbuf_invariant.cpp"), "-o", str(exe)], check=True,
                   capture_output=True, timeout=300)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
