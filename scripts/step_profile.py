"""Per-kernel HIP-event times of one hot-path step (eager), and host-side issue time per step."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import clrs_amd
from clrs_amd.sharded import HipLocal, ShardedSchur
torch.cuda.set_device(0)
torch.cuda.set_stream(torch.cuda.Stream())
name = sys.argv[1] if len(sys.argv) > 1 else "ce"
if name == "ce":
    flat = bench.build_problem(1)
else:
    from tests.util import flat as tf
    flat = tf(name)
if len(sys.argv) > 2:      # e.g. "wave2_assemble=2": library configuration keys set before the context is created
    from clrs_amd import _lib
    for kv in sys.argv[2:]:
        key, val = kv.split("=")
        _lib.check(_lib.load().clrs_config_set(key.encode(), int(val)))
sh = ShardedSchur(flat, 0, 1, lambda s: HipLocal(s, 0, graph=False))
f, ctx, dev = sh.shard, sh.local.ctx, "cuda:0"
X, Y = bench.seeded_iterates(flat, seed=1)
tX, tY = torch.from_numpy(X).to(dev), torch.from_numpy(Y).to(dev)
tXc = torch.empty_like(tX)
rng = np.random.default_rng(2)
trx, tryy = torch.from_numpy(rng.standard_normal(flat.x_len)).to(dev), torch.from_numpy(rng.standard_normal(flat.n_free)).to(dev)
tdx, tdy = torch.empty_like(trx), torch.empty_like(tryy)
def step():
    sh.local.cholesky_blocks(tX, tXc); sh.decompose(tXc, tY); sh.solve(trx, tryy, tdx, tdy); sh.solve(trx, tryy, tdx, tdy)
for _ in range(20): step()
torch.cuda.synchronize()
prof = bench.kernel_profile(ctx, step, 50)
tot = 0
for k, v in sorted(prof.items(), key=lambda kv: -kv[1][2]):
    print(f"{k:24s} avg {1e6*v[0]:7.2f} us  x{v[1]:4.1f}/step  = {1e6*v[2]:7.2f} us/step"); tot += v[2]
print("sum of kernel time per step: %.1f us" % (1e6 * tot))
torch.cuda.synchronize()
K = 300
t0 = time.perf_counter()
for _ in range(K): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host issue time per step: %.1f us; wall per step incl. drain: %.1f us" % (1e6 * (t1 - t0) / K, 1e6 * (t2 - t0) / K))
