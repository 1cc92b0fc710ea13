"""Dense ("high rank") branch of the fp64 assembly at scale: sdpa_scaled(nb, bs, m, blocks_per_constraint=nb).  usage: dense_roofline.py [nb bs m]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import clrs_amd
from clrs_amd.problems import sdpa_scaled, sdpa_to_sdp
from clrs_amd.solver import SchurContext
from oracle.oracle import Oracle
nb, bs, m = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 32, 256)
t0 = time.time(); f = clrs_amd.flatten(sdpa_to_sdp(sdpa_scaled(nb=nb, bs=bs, m=m, blocks_per_constraint=nb))); print("generated in %.1fs" % (time.time() - t0), f.dense_A.size * 8 / 1e6, "MB of constraint data")
rng = np.random.default_rng(1)
X, Y = np.zeros(f.xy_len), np.zeros(f.xy_len)
for b in range(f.n_blocks):
    n = int(f.block_n[b]); o = int(f.block_off[b])
    for M in (X, Y):
        G = rng.standard_normal((n, n)); M[o:o + n * n] = (np.eye(n) + G @ G.T / n).reshape(-1)
Xc = np.concatenate([np.linalg.cholesky(X[f.block_off[b]:f.block_off[b + 1]].reshape(bs, bs)).reshape(-1, order="F") for b in range(f.n_blocks)])
ctx = SchurContext(f)
print("plan", ctx.plan_info(), "fused clusters", ctx.fused_clusters())
dev = "cuda:0"
tXc, tY = torch.from_numpy(Xc).to(dev), torch.from_numpy(Y).to(dev)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for _ in range(3): ctx.assemble_dev(tXc.data_ptr(), tY.data_ptr())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ctx.assemble_dev(tXc.data_ptr(), tY.data_ptr())
e1.record(); e1.synchronize()
t = 1e-3 * e0.elapsed_time(e1) / 10
cnt = ctx.counters()
print("assembly %.3f ms, %.2f GFLOP algorithmic -> %.2f TFLOP/s = %.1f %% of 78.6" % (1e3 * t, cnt["assemble_flops"] / 1e9, cnt["assemble_flops"] / t / 1e12, 100 * cnt["assemble_flops"] / t / 78.6e12))
S, _ = ctx.compute_S_integrated(Xc, Y)
t0 = time.time(); S_ref, _ = Oracle(f, quad=False).schur_assemble(Xc, Y); print("oracle %.1fs" % (time.time() - t0), "S rel err", np.max(np.abs(S - S_ref)) / np.max(np.abs(S_ref)))
ctx.set_kernel_timing(-1)
for _ in range(5): ctx.assemble_dev(tXc.data_ptr(), tY.data_ptr())
for k, (kind, sec, n_) in ctx.kernel_times().items(): print("  %-28s %8.1f us per assembly (%d launches)" % (k, 1e6 * sec / 5, n_ // 5))
