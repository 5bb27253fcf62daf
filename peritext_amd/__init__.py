"""peritext_amd — MI355X-native batch merge engine for Peritext's hot path (apply op logs + materialise spans).

Layout: csrc/ (HIP kernels + C ABI -> lib/libperitext_hip.so: merge_core.h batch merge, replay_core.h Patch[] streams,
gen_core.h / change_core.h on-device change(), cursor_core.h cursors), abi.py (ctypes mirror of include/peritext_hip.h), wire.py (Change JSON <-> SoA op log, span /
patch / change decoders), canon.py (canonical output + digest), engine.py (device driver over the C ABI), shard.py (document
sharding + digest all-gather), workloads.py (the BASELINE workload table), node/ (N-API addon + JS/TS host with the reference's
Change / InputOperation / Patch / FormatSpanWithText surface).
"""
from . import abi, canon, wire  # noqa: F401

__all__ = ["abi", "canon", "wire"]
