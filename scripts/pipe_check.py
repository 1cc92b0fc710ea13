"""Pipelined factorisations (csrc/clrs_mw_pipe.hip.h) on and off: whole solves of the named problems, ms per iteration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clrs_amd
from clrs_amd.mw import MwSchurContext, solvesdp_mw
from clrs_amd.problems import cohnelkies
from tests.util import flat

for label, f in (("cohnelkies(8,15)", clrs_amd.flatten(cohnelkies(8, 15))), ("polyopt40", flat("polyopt40")), ("delsarte_3_10", flat("delsarte_3_10")), ("threepoint_4", flat("threepoint_4"))):
    kw = dict(omega_p=1e3, omega_d=1e3) if label == "threepoint_4" else {}
    for K in (5, 10) if label.startswith("cohn") else (5,):
        for pipe in (False, True):
            ctx = MwSchurContext(f, limbs=K, pipeline=pipe)
            solvesdp_mw(f, ctx=ctx, maxiterations=3, **kw)
            best = None
            for _ in range(3):
                r = solvesdp_mw(f, ctx=ctx, **kw)
                if best is None or r.time_total < best.time_total:
                    best = r
            ctx.close()
            print("%-18s K=%2d pipeline=%d: %s, %d iterations, objective %.14g, %.3f ms per iteration" % (label, K, pipe, best.status, best.iterations, best.primal_objective,
                  1e3 * best.time_total / best.iterations), flush=True)
