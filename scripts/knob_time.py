import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from tests.util import flat
from clrs_amd import _lib
from clrs_amd.mw import MwSchurContext, solvesdp_mw
knob, val = sys.argv[1].encode(), int(sys.argv[2])
_lib.check(_lib.load().clrs_config_set(knob, val))
for name in sys.argv[3:]:
    kw = dict(omega_p=1e3, omega_d=1e3) if name.startswith("threepoint") else {}
    f = flat(name); ctx = MwSchurContext(f, limbs=5)
    solvesdp_mw(f, ctx=ctx, limbs=5, **kw)
    t = []
    for _ in range(5):
        r = solvesdp_mw(f, ctx=ctx, limbs=5, **kw); t.append(1e3 * r.time_total / r.iterations)
    print("%s = %d  %-16s %d it %s obj %.15g  ms/it min %.4f" % (knob.decode(), val, name, r.iterations, r.status, r.primal_objective, min(t)), flush=True)
    ctx.close()
