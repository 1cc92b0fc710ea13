"""k_cluster_assemble_w3 alone on the roofline instance of bench.py (16384 clusters of the cohnelkies(8,15) shapes): average time of launches back to back
between two HIP events, algorithmic GB/s, and (unless --no-check) the parity of the first and last cluster against the fp64 oracle.
    [CLRS_HIP_LIB=csrc/_diag/<variant>.so] python scripts/w3_time.py [--copies 8192] [--launches 200] [--no-check]"""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import clrs_amd
from clrs_amd.problems import cohnelkies
from clrs_amd.sdp import replicate_clusters
from clrs_amd.solver import SchurContext

ap = argparse.ArgumentParser()
ap.add_argument("--copies", type=int, default=8192)
ap.add_argument("--launches", type=int, default=200)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--no-check", action="store_true")
args = ap.parse_args()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
from bench_fp64 import seeded_iterates
from clrs_amd.sharded import _DevArray
f = clrs_amd.flatten(cohnelkies(8, 15))
big = replicate_clusters(f, args.copies)
dev = "cuda:0"
ctx = SchurContext(big, device=0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
bX, bY = seeded_iterates(big, seed=3)
bXc = np.concatenate([np.linalg.cholesky(bX[big.block_off[b]:big.block_off[b + 1]].reshape(int(big.block_n[b]), -1, order="F")).reshape(-1, order="F")
                      for b in range(big.n_blocks)])
tX, tY = torch.from_numpy(bXc).to(dev), torch.from_numpy(bY).to(dev)
for _ in range(50):
    ctx.assemble_dev(tX.data_ptr(), tY.data_ptr())
torch.cuda.synchronize()
alg = ctx.counters()["assemble_bytes"]
best = []
for r in range(args.rounds):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.launches):
        ctx.assemble_dev(tX.data_ptr(), tY.data_ptr())
    e1.record()
    e1.synchronize()
    best.append(1e3 * e0.elapsed_time(e1) / args.launches)
us = float(np.median(best))
print("k_cluster_assemble_w3: %d clusters, %.2f us per launch (rounds %s), %.0f GB/s algorithmic = %.3f of 8 TB/s" %
      (big.n_clusters, us, " ".join("%.2f" % b for b in best), alg / us / 1e3, alg / us / 1e3 / 8000.0), flush=True)
if not args.no_check:
    from oracle.oracle import Oracle
    S = torch.as_tensor(_DevArray(ctx.S_buffer(), big.S_len), device=dev).cpu().numpy()
    ob = Oracle(f, quad=False)
    nxy = f.xy_len
    worst = 0.0
    for k in (0, args.copies // 2, args.copies - 1):
        Sk, _ = ob.schur_assemble(bXc[k * nxy:(k + 1) * nxy], bY[k * nxy:(k + 1) * nxy])
        worst = max(worst, float(np.max(np.abs(S[k * f.S_len:(k + 1) * f.S_len] - Sk)) / np.max(np.abs(Sk))))
    print("parity against the fp64 oracle (three clusters): max relative error %.2e" % worst)
