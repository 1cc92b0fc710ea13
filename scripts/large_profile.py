"""Staged (large-block) regime: per-kernel times of one Schur assembly + factorisation on the roofline instances R."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import clrs_amd
import bench_fp64 as bench
from clrs_amd import problems as P
from clrs_amd.solver import SchurContext
which = sys.argv[1] if len(sys.argv) > 1 else "polyopt512"
if which.startswith("polyopt"):
    flat = clrs_amd.flatten(P.polyopt_scaled(int(which[7:])))
elif which == "sdpa64":
    flat = clrs_amd.flatten(P.sdpa_to_sdp(P.sdpa_scaled(nb=64, bs=32, m=256, seed=64)))
torch.cuda.set_device(0)
ctx = SchurContext(flat)
print(which, "P", list(flat.cluster_P), "n", list(flat.block_n[:4]), "fused clusters", ctx.fused_clusters(), "plan", ctx.plan_info())
X, Y = bench.seeded_iterates(flat, seed=1)
dev = "cuda:0"
tX, tY = torch.from_numpy(X).to(dev), torch.from_numpy(Y).to(dev)
tXc = torch.empty_like(tX)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
def step():
    ctx.cholesky_blocks_dev(tX.data_ptr(), tXc.data_ptr())
    ctx.assemble_dev(tXc.data_ptr(), tY.data_ptr())
    ctx.factor_dev()
for _ in range(2): step()
torch.cuda.synchronize()
print("status", ctx.sync_status())
prof = bench.kernel_profile(ctx, step, 3)
cnt = ctx.counters()
tot = 0
for k, v in sorted(prof.items(), key=lambda kv: -kv[1][2]):
    print(f"{k:24s} avg {1e6*v[0]:9.1f} us  x{v[1]:6.1f}/step  = {1e6*v[2]:10.1f} us/step"); tot += v[2]
print("sum %.1f us;  assembly algorithmic %.2f GFLOP, %.1f MB; factor %.2f GFLOP" % (1e6*tot, cnt["assemble_flops"]/1e9, cnt["assemble_bytes"]/1e6, cnt["factor_flops"]/1e9))
step()                       # the launch graphs are captured again after the per-kernel timing above
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print("wall per (cholX + assemble + factor): %.2f ms -> %.2f TFLOP/s overall" % (1e3*dt, (cnt["assemble_flops"]+cnt["factor_flops"])/dt/1e12))
for name, fn in (("cholX", lambda: ctx.cholesky_blocks_dev(tX.data_ptr(), tXc.data_ptr())), ("assemble", lambda: ctx.assemble_dev(tXc.data_ptr(), tY.data_ptr())),
                 ("assemble + factor", lambda: (ctx.assemble_dev(tXc.data_ptr(), tY.data_ptr()), ctx.factor_dev()))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize()
    print("  phase %-18s %.2f ms" % (name, 1e3 * (time.perf_counter() - t0) / 3))
if "--graph" in sys.argv:       # the same step with every library call replayed as one hipGraph (on a stream of its own: the null stream cannot be captured)
    gs = torch.cuda.Stream()
    ctx.set_stream(gs.cuda_stream)
    ctx.set_graph_mode(True)
    for _ in range(2): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize()
    print("  hipGraph replay: %.2f ms per (cholX + assemble + factor)" % (1e3 * (time.perf_counter() - t0) / 5))

