"""peritext_amd — MI355X-native batch merge engine for Peritext's hot path (apply op logs + materialise spans).

Layout: csrc/ (HIP kernels + C ABI -> lib/libperitext_hip.so), abi.py (ctypes mirror of include/peritext_hip.h),
wire.py (Change JSON <-> SoA op log), canon.py (canonical output + digest), engine.py (device driver),
host.py (mirror of the reference's Micromerge surface for this path), node/ (N-API addon + JS/TS host).
"""
from . import abi, canon, wire  # noqa: F401

__all__ = ["abi", "canon", "wire"]
