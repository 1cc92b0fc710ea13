"""The fp64 Schur assembly on many-cluster instances with the block shapes of the named problems: which kernels run, their time per assembly, the
algorithmic bytes and flops of SURVEY.md section 8d per assembly against the HBM and fp64-MFMA roofs, and the parity of three clusters against the fp64 oracle.
    python scripts/fp64_assembly_shapes.py [name copies] ...      (default: the shapes VERDICT r4 #2 names)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import clrs_amd
from tests.util import flat
from clrs_amd.sdp import replicate_clusters
from clrs_amd.solver import SchurContext
from clrs_amd.sharded import _DevArray
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
from bench_fp64 import seeded_iterates

def run(name, copies, check=True):
    f = flat(name)
    big = replicate_clusters(f, copies)
    dev = "cuda:0"
    ctx = SchurContext(big, device=0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    bX, bY = seeded_iterates(big, seed=3)
    bXc = np.concatenate([np.linalg.cholesky(bX[big.block_off[b]:big.block_off[b + 1]].reshape(int(big.block_n[b]), -1, order="F")).reshape(-1, order="F")
                          for b in range(big.n_blocks)])
    tX, tY = torch.from_numpy(bXc).to(dev), torch.from_numpy(bY).to(dev)
    for _ in range(20):
        ctx.assemble_dev(tX.data_ptr(), tY.data_ptr())
    torch.cuda.synchronize()
    c = ctx.counters()
    ctx.set_kernel_timing(-1)
    for _ in range(10):
        ctx.assemble_dev(tX.data_ptr(), tY.data_ptr())
    torch.cuda.synchronize()
    kt = {k: (1e6 * v[1] / 10, v[2] / 10) for k, v in ctx.kernel_times().items() if v[2] > 0}
    ctx.set_kernel_timing(-2)
    best = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ctx.assemble_dev(tX.data_ptr(), tY.data_ptr())
        e1.record(); e1.synchronize()
        best.append(1e3 * e0.elapsed_time(e1) / 50)
    us = float(np.median(best))
    shapes = sorted(set((int(n), int(P)) for n, P in zip(f.block_n, f.cluster_P[f.block_cluster])))
    print("%s x %d: %d clusters (%d by k_cluster_assemble_w4, %d by _w5), %d blocks (n, P of its cluster: %s)" % (name, copies, big.n_clusters, ctx.wave4_clusters(), ctx.wave5_clusters(), big.n_blocks, shapes))
    print("   assembly %.1f us; algorithmic %.1f MB -> %.0f GB/s = %.3f of 8 TB/s; %.2f GFLOP -> %.1f TFLOP/s = %.3f of 78.6" %
          (us, c["assemble_bytes"] / 1e6, c["assemble_bytes"] / us / 1e3, c["assemble_bytes"] / us / 1e3 / 8000, c["assemble_flops"] / 1e9,
           c["assemble_flops"] / us / 1e6, c["assemble_flops"] / us / 1e6 / 78.6))
    print("   kernels per assembly: " + ", ".join("%s %.1f us x%.0f" % (k, v[0], v[1]) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])), flush=True)
    if check:
        from oracle.oracle import Oracle
        S = torch.as_tensor(_DevArray(ctx.S_buffer(), big.S_len), device=dev).cpu().numpy()
        ob = Oracle(f, quad=False)
        nxy = f.xy_len
        worst = 0.0
        for k in (0, copies // 2, copies - 1):
            Sk, _ = ob.schur_assemble(bXc[k * nxy:(k + 1) * nxy], bY[k * nxy:(k + 1) * nxy])
            worst = max(worst, float(np.max(np.abs(S[k * f.S_len:(k + 1) * f.S_len] - Sk)) / np.max(np.abs(Sk))))
        print("   parity against the fp64 oracle (three copies): max relative error %.2e" % worst, flush=True)

args = sys.argv[1:]
cases = [(args[i], int(args[i + 1])) for i in range(0, len(args), 2)] if args else [("ce_8_15", 8192), ("polyopt40", 2048), ("ns_8_15_2", 512), ("delsarte_3_10", 8192), ("threepoint_4", 256)]
for name, copies in cases:
    run(name, copies)
