"""The weak-scaled instances bench.py solves on N ranks (cohnelkies_multi with 2N - 1 scalings), unsharded on one GPU: status, iterations, objective, time --
what the N-rank job must reproduce:  gpurun -- python scripts/multi_instances_check.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clrs_amd
from clrs_amd.mw import solvesdp_mw
from clrs_amd.problems import cohnelkies_multi
from clrs_amd.sharded import partition_clusters

thr = dict(dual_error_threshold=1e-30, primal_error_threshold=1e-30, duality_gap_threshold=1e-15)
for world in (1, 2, 4, 8):
    t0 = time.time()
    full = clrs_amd.flatten(cohnelkies_multi(8, 15, [1.0 + 0.0625 * k for k in range(2 * world - 1)]))
    tg = time.time() - t0
    parts = partition_clusters(full, world)
    r = solvesdp_mw(full, limbs=5, **thr)
    r = solvesdp_mw(full, limbs=5, **thr)
    print(f"world {world}: {full.n_clusters} clusters, N = {full.n_free}, generated in {tg:.1f} s, clusters per rank {[len(p) for p in parts]}: {r.status} code {r.error_code}, "
          f"{r.iterations} iterations, objective {r.primal_objective:.12g}, {1e3 * r.time_total / r.iterations:.3f} ms per iteration unsharded", flush=True)
