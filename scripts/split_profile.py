"""Kernel-level view of the cluster-sharded step on ONE GPU (1-rank RCCL group, split-phase calls): run under rocprofv3 --kernel-trace --stats."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

import bench
import clrs_amd  # noqa: F401
from clrs_amd.sharded import HipLocal, ShardedSchur

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("NCCL_DEBUG", "WARN")
torch.cuda.set_device(0)
torch.cuda.set_stream(torch.cuda.Stream())
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
flat = bench.build_problem(1)
sh = ShardedSchur(flat, 0, 1, lambda s: HipLocal(s, 0), parts=[[0, 1]], force_split=True)
dev = "cuda:0"
X, Y = bench.seeded_iterates(flat, seed=1)
tX, tY = torch.from_numpy(X).to(dev), torch.from_numpy(Y).to(dev)
tXc = torch.empty_like(tX)
rng = np.random.default_rng(2)
trx, tryy = torch.from_numpy(rng.standard_normal(flat.x_len)).to(dev), torch.from_numpy(rng.standard_normal(flat.n_free)).to(dev)
tdx, tdy = torch.empty_like(trx), torch.empty_like(tryy)


def step():
    sh.local.cholesky_blocks(tX, tXc)
    sh.decompose(tXc, tY)
    sh.solve(trx, tryy, tdx, tdy)
    sh.solve(trx, tryy, tdx, tdy)


for _ in range(30):
    step()
torch.cuda.synchronize()
K = 300
t0 = time.perf_counter()
for _ in range(K):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("split step: host issue %.1f us, wall %.1f us" % (1e6 * (t1 - t0) / K, 1e6 * (t2 - t0) / K), file=sys.stderr)
sh.close()
dist.destroy_process_group()
