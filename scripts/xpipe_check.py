"""The Cholesky of the X blocks through the pipelines (k_mw_potrf_x_pipe) against the one-workgroup kernel: factors and inverse factors bit for bit
(stand-alone entry point), and whole solves with the knob off / on: iterations, objective, ms per iteration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.util import flat, spd_iterates, mw_from_double
from clrs_amd import _lib
from clrs_amd.mw import MwSchurContext, solvesdp_mw
L = _lib.load()
names = sys.argv[1:] or ["ce_8_15", "polyopt40", "ns_8_15_2", "ns_8_15_3", "threepoint_4", "sdpa_x64", "delsarte_3_10"]
for name in names:
    kw = dict(omega_p=1e3, omega_d=1e3) if name.startswith("threepoint") else {}
    f = flat(name)
    res = {}
    for v in (0, 1):
        _lib.check(L.clrs_config_set(b"mw_pipeline_x", v))
        ctx = MwSchurContext(f, limbs=5)
        X, Y = spd_iterates(f, seed=3)
        Xc = ctx.cholesky_blocks(mw_from_double(X, 5))
        solvesdp_mw(f, ctx=ctx, limbs=5, **kw)
        t = []
        for _ in range(6):
            r = solvesdp_mw(f, ctx=ctx, limbs=5, **kw)
            t.append(1e3 * r.time_total / r.iterations)
        res[v] = (np.array(Xc), r, min(t))
        ctx.close()
    _lib.check(L.clrs_config_set(b"mw_pipeline_x", 1))
    same = np.array_equal(res[0][0], res[1][0])
    r0, r1 = res[0][1], res[1][1]
    print("%-14s chol(X) bit-identical: %s | off: %d it %s obj %.15g %.4f ms/it | on: %d it %s obj %.15g %.4f ms/it | y identical: %s" %
          (name, same, r0.iterations, r0.status, r0.primal_objective, res[0][2], r1.iterations, r1.status, r1.primal_objective, res[1][2], np.array_equal(np.asarray(r0.y), np.asarray(r1.y))), flush=True)
