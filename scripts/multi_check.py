import sys, time
sys.path.insert(0, "/root/repo")
import clrs_amd
from clrs_amd.problems import cohnelkies_multi
from clrs_amd.mw import solvesdp_mw
for world in (2, 4, 8):
    f = clrs_amd.flatten(cohnelkies_multi(8, 15, [1.0 + 0.125 * k for k in range(2 * world - 1)]))
    t = time.time(); r = solvesdp_mw(f, limbs=5); dt = time.time() - t
    print(world, f.n_clusters, r.status, r.error_code, r.iterations, r.primal_objective, "%.3f s" % dt, flush=True)
