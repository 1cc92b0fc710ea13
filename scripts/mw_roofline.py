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
flat = clrs_amd.flatten(cohnelkies(8, 15))
big = replicate_clusters(flat, copies)
ctx = MwSchurContext(big, limbs=K)
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
print("limbs %d, %d clusters: assembly %.1f us, %.3g multiply-adds -> %.2f TFLOP/s at K(K+1) flops each" % (K, big.n_clusters, 1e6 * t, m, m * K * (K + 1) / t / 1e12))
ctx.close()
