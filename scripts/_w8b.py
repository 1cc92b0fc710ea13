import sys, os, time
sys.path.insert(0, "/root/repo")
import clrs_amd
from clrs_amd.mw import solvesdp_mw, MwSchurContext
from clrs_amd.problems import cohnelkies_multi
thr = dict(dual_error_threshold=1e-30, primal_error_threshold=1e-30, duality_gap_threshold=1e-15)
n = int(sys.argv[1])
full = clrs_amd.flatten(cohnelkies_multi(8, 15, [1.0 + 0.125 * k for k in range(n)]))
ctx = MwSchurContext(full, limbs=5)
for rep in range(2):
    r = solvesdp_mw(full, ctx=ctx, maxiterations=20, **thr)
print(full.n_clusters, "clusters", r.status, r.iterations, "%.3f ms/it" % (1e3 * r.time_total / r.iterations), flush=True)
ctx.close()
