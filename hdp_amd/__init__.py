"""hdp_amd: MI355X (gfx950) implementation of the HDP hot path.

Drop-in for ``hdp.threshold.compute_thresholds`` and
``hdp.metric.compute_group_metrics`` (and their single-variable forms); see
``hdp_amd.threshold`` / ``hdp_amd.metric`` for the xarray-level signatures and
``hdp_amd.core`` for the array-level API.  The compute runs in hand-written HIP
kernels behind the C ABI of ``include/hdp_hip.h``; importing this package does not
need a GPU, calling a compute function does.
"""
__version__ = "0.1.0"

from . import _lib, calendar, core  # noqa: F401
