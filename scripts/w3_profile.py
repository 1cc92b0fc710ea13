#!/usr/bin/env python3
"""Assembly-only loop on the roofline instance (cohnelkies(8,15) block shapes, many clusters) for rocprofv3 passes.

    python3 scripts/w3_profile.py [copies] [reps] [wave3 0|1]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    copies = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    wave3 = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
    import torch
    torch.cuda.set_device(0)
    import clrs_amd
    from clrs_amd.problems import cohnelkies
    from clrs_amd.sdp import replicate_clusters
    from clrs_amd.solver import SchurContext
    f = clrs_amd.flatten(cohnelkies(8, 15))
    big = replicate_clusters(f, copies)
    rng = np.random.default_rng(3)
    # cheap SPD iterates: the same pair of blocks everywhere would sit in cache, so draw them all
    X, Y = np.zeros(big.xy_len), np.zeros(big.xy_len)
    Xc = np.zeros(big.xy_len)
    for b in range(big.n_blocks):
        n = int(big.block_n[b])
        sl = slice(int(big.block_off[b]), int(big.block_off[b + 1]))
        G = rng.standard_normal((n, n))
        Xc[sl] = np.linalg.cholesky(np.eye(n) + G @ G.T / n).reshape(-1, order="F")
        G = rng.standard_normal((n, n))
        Y[sl] = (np.eye(n) + G @ G.T / n).reshape(-1, order="F")
    ctx = SchurContext(big, wave3=wave3)
    tX, tY = torch.from_numpy(Xc).to("cuda:0"), torch.from_numpy(Y).to("cuda:0")
    torch.cuda.synchronize()
    for _ in range(3):
        ctx.assemble_dev(tX.data_ptr(), tY.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.assemble_dev(tX.data_ptr(), tY.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    cnt = ctx.counters()
    print(f"clusters {big.n_clusters} blocks {big.n_blocks}: {1e6 * dt:.1f} us per assembly (host clock, back to back), "
          f"{cnt['assemble_bytes'] / dt / 1e9:.0f} GB/s algorithmic, {cnt['assemble_flops'] / dt / 1e12:.2f} TFLOP/s")
    ctx.close()
    # the streaming rate of this device at the same footprint and read : write mix (PMC traffic of the 16384-cluster launch, scaled)
    import ctypes as C
    from clrs_amd._lib import load, check
    scale = big.n_clusters / 16384.0
    rd, wr = int(218711490 * scale), int(140524523 * scale)
    us = C.c_double(0.0)
    check(load().clrs_test_stream(0, rd, wr, reps, C.byref(us)))
    print(f"    traffic {1e-6 * (rd + wr):.0f} MB per launch -> kernel {(rd + wr) / dt / 1e9:.0f} GB/s; streaming probe of that footprint "
          f"{us.value:.1f} us = {(rd + wr) / us.value / 1e3:.0f} GB/s; kernel / probe = {us.value * 1e-6 / dt:.2f}")


if __name__ == "__main__":
    main()
