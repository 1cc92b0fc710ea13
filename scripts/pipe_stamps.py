"""step stamps of the pipelined factorisations (csrc/clrs_mw_pipe.hip.h) during a solve of cohnelkies(8,15)
(diagnostic build: `CLRS_MW_STAMPS=1 python -c "from clrs_amd import _lib; _lib.build()"` here, then run with
CLRS_HIP_LIB=clusteredlowranksolver.jl_amd/csrc/_diag/libclrs_hip_mwstamps.so; the product library carries no stamps)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.util import flat
from clrs_amd.mw import MwSchurContext, solvesdp_mw
f = flat("ce_8_15")
ctx = MwSchurContext(f, limbs=5)
solvesdp_mw(f, ctx=ctx, maxiterations=5)
assert ctx.L.clrs_mw_debug_pipe_stamps(ctx.h, None) == 0
solvesdp_mw(f, ctx=ctx, maxiterations=20)
st = (C.c_uint64 * (16 * 40 + 8 * 32 * 4 * 4))()      # (behind the step stamps: the stamps inside the steps, scripts/pipe_substamps.py)
assert ctx.L.clrs_mw_debug_pipe_stamps(ctx.h, st) == 0
v = np.array(list(st), dtype=np.int64)[:16 * 40].reshape(16, 40)
for base, name in ((0, "k_mw_factor_pipe, cluster 0"),):      # (rows 8.. hold k_mw_potrf_q_pipe's stamps when the pipeline of Q is on: mw_pipeline = 2; else scripts/chain_stamps.py's)
    t0 = min(int(v[base + r, 39]) for r in range(8) if v[base + r, 39])
    print(name, "(us from the first workgroup's start; role 0-3 stages, 4-7 W)")
    for r in range(8):
        row = v[base + r]
        steps = [(int(x) - t0) / 100.0 for x in row[:32] if x]
        print("  role %d: start %.1f end %.1f; steps at" % (r, (int(row[39]) - t0) / 100.0, (int(row[38]) - t0) / 100.0), " ".join("%.1f" % s for s in steps))
        if any(row[32:38]):
            print("          loader begins to complete columns 28..31 at", " ".join("%.1f" % ((int(x) - t0) / 100.0) for x in row[32:36]), "; has columns 30, 31 at",
                  " ".join("%.1f" % ((int(x) - t0) / 100.0) for x in row[36:38]))
ctx.close()
