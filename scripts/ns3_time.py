"""Nsphere_packing(8,15,[1/2,1/2,1/2]) (N = 3: 11 clusters, P = 192, N_free = 193) unsharded at 6 limbs (the reference's prec = 300): whole solve"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clrs_amd
from clrs_amd.problems import nsphere_packing
from clrs_amd.mw import solvesdp_mw
t0 = time.time()
f = clrs_amd.flatten(nsphere_packing(8, 15, [0.5, 0.5, 0.5]))
print("clusters", f.n_clusters, "P", list(f.cluster_P), "N", f.n_free, "generated in %.1f s" % (time.time() - t0), flush=True)
solvesdp_mw(f, limbs=6, maxiterations=2)
for _ in range(2):
    r = solvesdp_mw(f, limbs=6)
    print(r.status, r.error_code, r.iterations, "%.12g" % r.primal_objective, "%.3f s = %.2f ms per iteration" % (r.time_total, 1e3 * r.time_total / r.iterations), flush=True)
