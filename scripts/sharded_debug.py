"""Sharded Nsphere_packing(8,15,[1/2,1/2,1/2]) on one GPU (in-process group): outcome per limb count / refinement setting."""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import clrs_amd
from clrs_amd.mw import LocalGroup, MwSchurContext, shard_problem, solvesdp_mw
from clrs_amd.problems import nsphere_packing

full = clrs_amd.flatten(nsphere_packing(8, 15, [0.5, 0.5, 0.5]))
for K, world, refine in [(5, 2, 0), (5, 2, 1), (5, 2, 2), (6, 2, 1), (5, 3, 1)]:
    group = LocalGroup(world)
    out, err = [None] * world, [None] * world

    def run(rank):
        try:
            shard, info = shard_problem(full, rank, world)
            ctx = MwSchurContext(shard, limbs=K, refine=refine)
            ctx.comm_init_local(group, rank)
            out[rank] = solvesdp_mw(shard, ctx=ctx, shard_info=info)
            ctx.close()
        except Exception as e:
            err[rank] = e
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    group.close()
    if any(err):
        print(K, world, refine, "errors", err, flush=True)
        continue
    r = out[0]
    print("K=%d world=%d refine=%d: %s code %d, %d iterations, objective %.14g gap %.3g errors %.3g %.3g" % (
        K, world, refine, r.status, r.error_code, r.iterations, r.primal_objective, r.duality_gap, r.dual_error, r.primal_error), flush=True)
    h = r.history
    print("   last rows (iter mu dobj pobj gap P p d ad ap beta):", flush=True)
    for row in h[-3:]:
        print("   ", " ".join("%.3e" % v for v in row), flush=True)
