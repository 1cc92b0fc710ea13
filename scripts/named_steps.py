#!/usr/bin/env python3
"""Hot-path step (chol X + assembly + factorisation + 2 solves) of the named shapes: device resident on the GPU against the fp64 CPU
port (oracle, best of 1 / 8 threads) on the same host.  python3 scripts/named_steps.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

torch.cuda.set_device(0)
import clrs_amd  # noqa: F401
from clrs_amd.solver import SchurContext
from oracle.oracle import Oracle
from tests.util import chol_blocks_np, flat, spd_iterates

for name in ("ce_8_15", "polyopt40", "delsarte_3_10", "threepoint_4", "ns_8_15_2", "sdpa_small"):
    f = flat(name)
    X, Y = spd_iterates(f, seed=1)
    Xc = chol_blocks_np(f, X)
    rng = np.random.default_rng(0)
    rx, ry = rng.standard_normal(f.x_len), rng.standard_normal(f.n_free)
    cpu = None
    for thr in (1, 8):
        o = Oracle(f, quad=False)
        o.set_num_threads(thr)

        def cstep():
            chol_blocks_np(f, X)
            o.schur_assemble(Xc, Y)
            o.schur_factor()
            o.schur_solve(rx, ry)
            o.schur_solve(rx, ry)
        for _ in range(3):
            cstep()
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < 1.0:
            cstep()
            n += 1
        dt = (time.perf_counter() - t0) / n
        cpu = dt if cpu is None else min(cpu, dt)
    ctx = SchurContext(f)
    dev = "cuda:0"
    tX, tY = torch.from_numpy(X).to(dev), torch.from_numpy(Y).to(dev)
    tXc = torch.empty_like(tX)
    trx, tryy = torch.from_numpy(rx).to(dev), torch.from_numpy(ry).to(dev)
    tdx, tdy = torch.empty_like(trx), torch.empty_like(tryy)

    def gstep():
        ctx.cholesky_blocks_dev(tX.data_ptr(), tXc.data_ptr())
        ctx.assemble_dev(tXc.data_ptr(), tY.data_ptr())
        ctx.factor_dev()
        for _ in range(2):
            ctx.solve_dev(trx.data_ptr(), tryy.data_ptr() if f.n_free else 0, tdx.data_ptr(), tdy.data_ptr() if f.n_free else 0)
    for _ in range(200):
        gstep()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(1000):
        gstep()
    torch.cuda.synchronize()
    g = (time.perf_counter() - t0) / 1000
    print(f"{name:16s} clusters {f.n_clusters:3d} blocks {f.n_blocks:3d} N {f.n_free:3d}: GPU {1e6 * g:7.1f} us per step, CPU port {1e6 * cpu:7.1f} us  ({cpu / g:.1f}x)", flush=True)
    ctx.close()
