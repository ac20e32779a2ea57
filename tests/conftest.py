import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionstart(session):
    """GPU runs that also use torch (device buffers for the full-size tests): torch must initialise HIP BEFORE
    libhdp_hip.so does -- it ships its own libamdhip64, and once the system one is loaded and has opened the device,
    torch's copy reports "No HIP GPUs are available".  Loaded first, its runtime is the one both sides share (this is
    also the order bench.py uses).  Without a device (the CPU suite) nothing is imported here."""
    if os.path.exists("/dev/kfd"):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass
