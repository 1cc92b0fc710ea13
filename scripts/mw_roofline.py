"""The multi-word Schur assembly on a many-cluster instance (the `roofline_mw` workload of bench.py), alone, for counter passes:
    rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -- python3 scripts/mw_roofline.py [limbs] [copies] [reps]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import clrs_amd
from clrs_amd.mw import MwSchurContext
from clrs_amd.problems import cohnelkies
from clrs_amd.sdp import replicate_clusters

K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
exact = {"auto": None, "on": True, "off": False}[sys.argv[4]] if len(sys.argv) > 4 else None      # exact slice products on the matrix cores (k_mws_pair)
flat = clrs_amd.flatten(cohnelkies(8, 15))
big = replicate_clusters(flat, copies)
ctx = MwSchurContext(big, limbs=K, exact_products=exact)
rng = np.random.default_rng(1)
X, Y = np.zeros((K, big.xy_len)), np.zeros((K, big.xy_len))
for b in range(big.n_blocks):
    n = int(big.block_n[b]); o = int(big.block_off[b])
    for M in (X, Y):
        G = rng.standard_normal((n, n))
        M[0, o:o + n * n] = (np.eye(n) + G @ G.T / n).reshape(-1)
dev = "cuda:0"
tX, tY = torch.tensor(X, device=dev), torch.tensor(Y, device=dev)
tXc = torch.empty_like(tX)
ctx.cholesky_blocks_dev(tX.data_ptr(), tXc.data_ptr())
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s_ = torch.cuda.ExternalStream(ctx.stream())
ctx.assemble_dev(tXc.data_ptr(), tY.data_ptr())
with torch.cuda.stream(s_):
    ev0.record()
    for _ in range(reps):
        ctx.assemble_dev(tXc.data_ptr(), tY.data_ptr())
    ev1.record()
ev1.synchronize()
t = 1e-3 * ev0.elapsed_time(ev1) / reps
m = ctx.counters()["assemble_muladds"]
print("exact products %s: limbs %d, %d clusters: assembly %.1f us, %.3g multiply-adds -> %.2f TFLOP/s at K(K+1) flops each" % (sys.argv[4] if len(sys.argv) > 4 else "auto", K, big.n_clusters, 1e6 * t, m, m * K * (K + 1) / t / 1e12))
if exact:
    import ctypes as C
    st = (C.c_uint64 * 16)()
    ctx.L.clrs_mw_debug_exact_stamps(ctx.h, None)
    ctx.assemble_dev(tXc.data_ptr(), tY.data_ptr())
    ctx.L.clrs_mw_debug_exact_stamps(ctx.h, st)
    v = [int(x) for x in st]
    print("k_mws_pair phase stamps of wave 0, workgroup 0 (us from start, 100 MHz clock):", ["%.1f" % ((x - v[0]) / 100.0) for x in v if x])
ctx.close()
