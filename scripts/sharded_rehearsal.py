"""The instances of bench.py --gpus N (N = 2, 4, 8) through the SHARDED code path on one GPU: N contexts of one process, one thread each, exchanging through the
in-process group (clrs_mw_local_group_*) -- everything of the N-rank job but RCCL itself: partition, records, rank-order reductions, device-side termination.
    gpurun -- python scripts/sharded_rehearsal.py [N ...] [--clusters-per-rank C]      (C = 2: bench.py's primary regime; 32: its filled one)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from clrs_amd.mw import solvesdp_mw
from clrs_amd.sharded import partition_clusters
from bench import weak_scaling_instance
from tests.test_mw_parity import _solve_sharded_in_threads

argv = sys.argv[1:]
C = 2
if "--clusters-per-rank" in argv:
    i = argv.index("--clusters-per-rank")
    C = int(argv[i + 1])
    del argv[i:i + 2]
thr = dict(dual_error_threshold=1e-30, primal_error_threshold=1e-30, duality_gap_threshold=1e-15)
for world in [int(a) for a in argv] or [2, 4, 8]:
    full = weak_scaling_instance(world, C)
    parts = partition_clusters(full, world)
    assert all(len(p) == C for p in parts), [len(p) for p in parts]
    solvesdp_mw(full, limbs=5, maxiterations=2, **thr)
    ref = solvesdp_mw(full, limbs=5, **thr)
    t0 = time.time()
    res = _solve_sharded_in_threads(full, world, K=5, **thr)
    dt = time.time() - t0
    r0 = res[0][0]
    same = all(np.array_equal(r.y, r0.y) and np.array_equal(r.history, r0.history) for r, _ in res[1:])
    print(f"{world} ranks x {C} clusters: {r0.status} code {r0.error_code}, {r0.iterations} iterations (unsharded {ref.iterations}, {ref.status}, "
          f"{1e3 * ref.time_total / ref.iterations:.3f} ms per iteration on the one GPU), objective {r0.primal_objective:.12g} "
          f"(unsharded {ref.primal_objective:.12g}), y and table rows bit-identical on all ranks: {same}, {1e3 * r0.time_total / r0.iterations:.3f} ms per iteration "
          f"with {world} contexts sharing the one GPU", flush=True)
