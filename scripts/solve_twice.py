"""time_total of repeated whole solves of one instance (fresh context each), to separate one-off costs from the iteration rate"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import flat
from clrs_amd.mw import solvesdp_mw, MwSchurContext
name = sys.argv[1] if len(sys.argv) > 1 else "polyopt40"
f = flat(name)
for i in range(4):
    t0 = time.time()
    r = solvesdp_mw(f, limbs=5, maxiterations=(2 if i == 0 else 500))
    print(name, "solve", i, "iterations", r.iterations, "time_total %.4f" % r.time_total, "wall incl. context %.4f" % (time.time() - t0), r.status, flush=True)
ctx = MwSchurContext(f, limbs=5)
for i in range(3):
    r = solvesdp_mw(f, limbs=5, ctx=ctx)
    print(name, "same context", i, "iterations", r.iterations, "time_total %.4f  -> %.3f ms per iteration" % (r.time_total, 1e3 * r.time_total / r.iterations), flush=True)
