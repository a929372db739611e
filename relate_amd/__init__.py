"""relate_amd -- MI355X-native all-pairs IBD HMM path of aalbrechtsen/relate.

The product is relate_amd/librelate_hip.so (C ABI: include/relate_hip.h; sources: relate_amd/csrc/).
This package only holds the ctypes plumbing used by tests, bench.py and the multi-GPU launcher.
"""
from .api import Relate, RelateError, findmax_bfgs, load_library, pinned_empty, rand_stream, shard  # noqa: F401
