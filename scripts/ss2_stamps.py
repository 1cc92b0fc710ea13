#!/usr/bin/env python3
"""Diagnostic: shader cycles per phase of k_solve_small2 on cohnelkies(8,15) (s_memtime-stamped build: scripts/w3_stamps.py build)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.set_device(0)
import clrs_amd
from clrs_amd import _lib
L = _lib.load(os.path.join(_lib.CSRC, "_diag", "libclrs_hip_w3stamps.so"))
from clrs_amd.solver import SchurContext, compute_T_decomposition, solve_system
from tests.util import chol_blocks_np, flat, spd_iterates
name = sys.argv[1] if len(sys.argv) > 1 else "ce_8_3"
if name == "synthetic_ce":      # the shapes of cohnelkies(8,15) (2 clusters, P = 32, N = 31) with random, well conditioned data
    from tests.util import random_simple_sdp
    import clrs_amd as _c
    f = _c.flatten(random_simple_sdp(0, J=2, n_free=31, max_P=32, max_n=16, definite=True))
else:
    f = flat(name)
X, Y = spd_iterates(f, seed=2)
ctx = SchurContext(f)
Xc = ctx.cholesky_blocks(X)
compute_T_decomposition(ctx, Xc, Y)
for _ in range(5):
    solve_system(ctx, np.ones(f.x_len), np.ones(f.n_free))
st = (C.c_uint64 * 16)()
L.clrs_debug_ss2_stamps.restype = C.c_int
L.clrs_debug_ss2_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
assert L.clrs_debug_ss2_stamps(ctx.h, st) == 0
st = np.array(st, dtype=np.float64)
names = ["loads -> LDS", "forward solves (one wave per cluster)", "u = LinvB^T t", "Q forward + backward solve", "LinvB dy partial sums", "backward solves + store"]
for i, n in enumerate(names):
    print(f"  {n:40s} {st[i + 1] - st[i]:8.0f} cycles")
print(f"  {'total':40s} {st[6] - st[0]:8.0f} cycles")
print(f"  inside the staging: entry -> all loads issued {st[8] - st[0]:6.0f}, -> all loads landed {st[9] - st[8]:6.0f}, -> stored + barrier {st[1] - st[9]:6.0f}")
L.clrs_debug_cf_stamps.restype = C.c_int
L.clrs_debug_cf_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
st = (C.c_uint64 * 16)()
assert L.clrs_debug_cf_stamps(ctx.h, st) == 0
st = np.array(st, dtype=np.float64)
print("k_cluster_factor (workgroup 0):")
for i, n in enumerate(["descriptor + loads of S_j, B_j -> LDS", "Cholesky of S_j", "store L_j", "L_j^-1 B_j", "store LinvB_j", "partial Q_j"]):
    print(f"  {n:40s} {st[i + 1] - st[i]:8.0f} cycles")
print(f"  {'total':40s} {st[6] - st[0]:8.0f} cycles")
ctx.close()
