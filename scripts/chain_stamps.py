"""Launch boundaries of the factorisation's chain by device clock, WITHOUT a profiler (diagnostic build: CLRS_MW_STAMPS=1 python -c "import __graft_entry__ as g; g.build()",
run with CLRS_HIP_LIB=clusteredlowranksolver.jl_amd/csrc/_diag/libclrs_hip_mwstamps.so): start of k_mw_linvb, start / end of k_mw_qgram, start of k_mw_potrf_q (its
first workgroup and its first ride workgroup) in the last iterations of a solve of cohnelkies(8,15) -- is the gap in front of k_mw_potrf_q that the kernel traces
show (8-14 us) there when nothing traces?"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.util import flat
from clrs_amd.mw import MwSchurContext, solvesdp_mw
f = flat("ce_8_15")
ctx = MwSchurContext(f, limbs=5)
solvesdp_mw(f, ctx=ctx, maxiterations=5)
assert ctx.L.clrs_mw_debug_pipe_stamps(ctx.h, None) == 0
rows = []
for n in (10, 20, 30, 40, 50):
    solvesdp_mw(f, ctx=ctx, maxiterations=n)
    st = (C.c_uint64 * (16 * 40))()
    assert ctx.L.clrs_mw_debug_pipe_stamps(ctx.h, st) == 0
    v = np.array(list(st), dtype=np.int64).reshape(16, 40)[8]
    t0 = int(v[0])
    rows.append([(int(v[i]) - t0) / 100.0 for i in range(6)])
print("us from the start of k_mw_linvb: [linvb start, qgram start, qgram last workgroup start, potrf_q start, potrf_q ride workgroup start, qgram last workgroup end]")
for r in rows:
    print("  ", " ".join("%7.2f" % x for x in r), "  -> gap qgram end -> potrf_q start %.2f us" % (r[3] - r[5]))
ctx.close()
