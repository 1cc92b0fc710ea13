"""ms per interior-point iteration of the instances that take the blocked factorisation, with the diagonal blocks through the pipeline (default) and
through the four-workgroup kernel (pipeline=False):  gpurun -- python scripts/blocked_pipe_check.py [names...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import flat
from clrs_amd.mw import MwSchurContext, solvesdp_mw

names = sys.argv[1:] or ["ns_8_15_2", "ns_8_15_3", "threepoint_3_8_8", "sdpa_x64"]
for name in names:
    f = flat(name)
    row = []
    for pipe in (False, True):
        ctx = MwSchurContext(f, limbs=5, pipeline=pipe)
        best = None
        for _ in range(3):
            r = solvesdp_mw(f, ctx=ctx)
            t = 1e3 * r.time_total / r.iterations
            best = t if best is None else min(best, t)
        row.append((best, r.status, r.iterations, r.primal_objective))
        ctx.close()
    print(f"{name:20s} four workgroups {row[0][0]:.3f} ms/iteration, pipeline {row[1][0]:.3f}  ({row[1][1]}, {row[1][2]} iterations, objective {row[1][3]:.12g}; same objective: {row[0][3] == row[1][3]})", flush=True)
