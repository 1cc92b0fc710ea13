"""clrs_amd -- MI355X-native hot path of ClusteredLowRankSolver.jl (Schur assembly + block-Cholesky
solve of the interior-point normal equations), behind a C ABI (include/clrs_hip.h).

Import through the repo-root shim:  `import clrs_amd`.
"""
from . import sdp  # noqa: F401
from .sdp import Block, ClusteredLowRankSDP, HiLo, LowRankMat, flatten  # noqa: F401

__all__ = ["sdp", "Block", "ClusteredLowRankSDP", "HiLo", "LowRankMat", "flatten"]
