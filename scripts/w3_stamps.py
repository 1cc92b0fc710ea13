#!/usr/bin/env python3
"""Diagnostic: shader cycles per phase of k_cluster_assemble_w3 (wave 0), from an s_memtime-stamped build.
   python scripts/w3_stamps.py build      (CPU box)
   python scripts/w3_stamps.py [copies]   (GPU box; CLRS_W3_WGS_PER_CU=1|2 selects the waves per SIMD)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import clrs_amd
from clrs_amd import _lib

OUT = os.path.join(_lib.CSRC, "_diag", "libclrs_hip_w3stamps.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    print(_lib.build(extra_flags=["-DCLRS_W3_STAMPS"], out=OUT))
    sys.exit(0)
import torch
torch.cuda.set_device(0)
L = _lib.load(OUT)
from clrs_amd.problems import cohnelkies
from clrs_amd.sdp import replicate_clusters
from clrs_amd.solver import SchurContext
from tests.util import chol_blocks_np, spd_iterates

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
big = replicate_clusters(clrs_amd.flatten(cohnelkies(8, 15)), copies)
X, Y = spd_iterates(big, seed=1)
Xc = chol_blocks_np(big, X)
ctx = SchurContext(big)
tX, tY = torch.from_numpy(Xc).to("cuda:0"), torch.from_numpy(Y).to("cuda:0")
for _ in range(5):
    ctx.assemble_dev(tX.data_ptr(), tY.data_ptr())
torch.cuda.synchronize()
st = (C.c_uint64 * 16)()
L.clrs_debug_w3_stamps.restype = C.c_int
L.clrs_debug_w3_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
assert L.clrs_debug_w3_stamps(ctx.h, st) == 0
st = np.array(st, dtype=np.float64)
nb = st[15]
names = ["loop overhead / descriptors", "issue next loads + end-of-cluster prefetch", "LDS stage of L + T_Y MFMAs", "substitution (L^-1)", "G_Y MFMAs",
         "A_Y store", "Z MFMAs + scale", "G_X MFMAs + S accumulate", "dense + S store (cluster ends only)"]
print(f"wave 0: {int(nb)} blocks; shader cycles per block, by phase:")
for i, n in enumerate(names):
    print(f"  {n:45s} {st[i] / nb:8.0f}")
print(f"  {'total':45s} {st[:9].sum() / nb:8.0f}")
ctx.close()
