"""expann_amd -- MI355X-native (gfx950) distance + top-k hot path of expANN.

Package contents: csrc/ (hand-written HIP kernels + the C ABI of include/expann_hip.h, built
into libexpann_hip.so) and the host-side mirror of the reference's engine interface.
"""
from . import _lib  # noqa: F401
from .engine import (GpuBruteForceEngine, ShardedBruteForceEngine, merge_topk_device,  # noqa: F401
                     merge_topk_strided_device)
from .pyrunner import AntitopoEngine  # noqa: F401

__all__ = ["GpuBruteForceEngine", "ShardedBruteForceEngine", "merge_topk_device", "merge_topk_strided_device", "AntitopoEngine"]
