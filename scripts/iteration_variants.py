"""cohnelkies(8,15) at 5 limbs: ms per interior-point iteration without the refinement step, with it (default), with the predictor refined too,
without / with the pipelined factorisations (S_j only is the default; True = S_j and Q)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clrs_amd
from clrs_amd.mw import MwSchurContext, solvesdp_mw
from clrs_amd.problems import cohnelkies
f = clrs_amd.flatten(cohnelkies(8, 15))
for kw in (dict(refine=0), dict(), dict(refine_predictor=True), dict(pipeline=False), dict(pipeline=True)):
    ctx = MwSchurContext(f, limbs=5, **kw)
    solvesdp_mw(f, ctx=ctx, maxiterations=3)
    best = min((solvesdp_mw(f, ctx=ctx) for _ in range(3)), key=lambda r: r.time_total)
    print(kw, best.status, best.iterations, "%.3f ms per iteration" % (1e3 * best.time_total / best.iterations), "errors %.2e %.2e" % (best.dual_error, best.primal_error), flush=True)
    ctx.close()
