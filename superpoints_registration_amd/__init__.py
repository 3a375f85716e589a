"""MI355X-native hot path of Superpoints_Registration (RegTR with direct
superpoint matching): KPConv pyramid preprocessing, KPConv encoder, superpoint
self/cross attention, matching and the weighted-SVD pose solve, computed by
hand-written HIP kernels (libspr_hip.so, C ABI in include/spr.h).

Importing the package does not load the library; the first operator call
does, and raises if it is missing (no fallback path).
"""
from .config import Config, get_config, load_config  # noqa: F401

__all__ = ["Config", "get_config", "load_config"]
