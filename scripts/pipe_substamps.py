"""stamps INSIDE the steps of the pipelined factorisation of cluster 0 (csrc/clrs_mw_pipe.hip.h, MWP_SUB) during a solve of cohnelkies(8,15): per role and
step, for the first lane of entry wave 0 / the last entry wave / the first loader wave / the loader wave whose turn the step is, the time from the top of the
step to: pivot read and scaled | arithmetic (or hand-off) done | behind the barrier.  Diagnostic build (CLRS_MW_STAMPS=1), run with CLRS_HIP_LIB=..._mwstamps.so"""
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
NW = 16 * 40 + 8 * 32 * 4 * 4
st = (C.c_uint64 * NW)()
assert ctx.L.clrs_mw_debug_pipe_stamps(ctx.h, st) == 0
v = np.array(list(st), dtype=np.int64)
top = v[:640].reshape(16, 40)
sub = v[640:].reshape(8, 32, 4, 4)
t0 = min(int(top[r, 39]) for r in range(8) if top[r, 39])
who = ["entry wave 0", "last entry wave", "first loader wave", "loader on turn"]
for r in range(8):
    print("role %d (%s): start %.1f end %.1f" % (r, "stage" if r < 4 else "W", (int(top[r, 39]) - t0) / 100.0, (int(top[r, 38]) - t0) / 100.0))
    for k in range(32):
        if not sub[r, k, 0, 0] and not sub[r, k, 2, 0]:
            continue
        base = min(int(x) for x in sub[r, k, :, 0] if x)
        cells = []
        for w in range(4):
            a = sub[r, k, w]
            if not a[0]:
                cells.append("%-34s" % ""); continue
            cells.append("%-34s" % ("%s +%.2f: %.2f %.2f %.2f" % (who[w][:12], (int(a[0]) - base) / 100.0, (int(a[1]) - int(a[0])) / 100.0, (int(a[2]) - int(a[0])) / 100.0, (int(a[3]) - int(a[0])) / 100.0)))
        print("   step %2d at %6.2f | %s" % (k, (base - t0) / 100.0, " | ".join(cells)))
ctx.close()
