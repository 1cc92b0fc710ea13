import sys, os
sys.path.insert(0, "/root/repo")
from tests.util import flat
from clrs_amd.mw import MwSchurContext, solvesdp_mw
for name in ("ns_8_15_2", "sdpa_x64"):
    f = flat(name)
    for K in (6, 8, 10):
        row = []
        for pipe in (False, None, True):
            ctx = MwSchurContext(f, limbs=K, pipeline=pipe)
            best = None
            for _ in range(2):
                r = solvesdp_mw(f, ctx=ctx)
                t = 1e3 * r.time_total / r.iterations
                best = t if best is None else min(best, t)
            row.append(best)
            ctx.close()
        print(f"{name} K={K}: off {row[0]:.3f}  default {row[1]:.3f}  forced {row[2]:.3f} ms/iteration ({r.status}, {r.iterations})", flush=True)
