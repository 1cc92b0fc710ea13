"""whole solves of the named instances with the library CLRS_HIP_LIB names (or the product): ms per iteration, min / median of the rounds.
usage: [CLRS_HIP_LIB=...] lib_time.py [rounds] names..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.util import flat
from clrs_amd.mw import MwSchurContext, solvesdp_mw
rounds = int(sys.argv[1])
for name in sys.argv[2:]:
    kw = dict(omega_p=1e3, omega_d=1e3) if name.startswith("threepoint") else {}
    f = flat(name)
    ctx = MwSchurContext(f, limbs=5)
    solvesdp_mw(f, ctx=ctx, limbs=5, **kw)
    t = []
    for r in range(rounds):
        x = solvesdp_mw(f, ctx=ctx, limbs=5, **kw)
        t.append(1e3 * x.time_total / x.iterations)
    t = np.sort(t)
    print("%-14s %s: %d iterations %s obj %.15g  ms/iteration min %.4f median %.4f" % (name, os.environ.get("CLRS_HIP_LIB", "product").split("/")[-1], x.iterations, x.status, x.primal_objective, t[0], t[len(t) // 2]), flush=True)
    ctx.close()
