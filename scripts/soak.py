"""Soak test of the device-resident loop: the same instance solved over and over on one context -- every solve must end with the same status, iteration
count and bit-identical y (the pipelined factorisations hand pivot columns over through tagged granules: a stale or torn hand-off would show here).
usage: soak.py name solves [limbs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.util import flat
from clrs_amd.mw import MwSchurContext, solvesdp_mw
name, solves = sys.argv[1], int(sys.argv[2])
K = int(sys.argv[3]) if len(sys.argv) > 3 else 5
kw = dict(omega_p=1e3, omega_d=1e3) if name.startswith("threepoint") else {}
f = flat(name)
ctx = MwSchurContext(f, limbs=K)
ref = solvesdp_mw(f, ctx=ctx, limbs=K, **kw)
bad = 0
t0 = time.time()
for i in range(solves):
    r = solvesdp_mw(f, ctx=ctx, limbs=K, **kw)
    same = r.status == ref.status and r.iterations == ref.iterations and np.array_equal(np.asarray(r.y), np.asarray(ref.y)) and r.error_code == ref.error_code
    if not same:
        bad += 1
        print("solve %d differs: %s %d iterations code %d (reference %s %d)" % (i, r.status, r.iterations, r.error_code, ref.status, ref.iterations), flush=True)
    if (i + 1) % 500 == 0:
        print("%d solves, %d differing, %.0f s" % (i + 1, bad, time.time() - t0), flush=True)
print("%s at %d limbs: %d solves of %d iterations (%s), %d differing from the first one" % (name, K, solves, ref.iterations, ref.status, bad))
ctx.close()
sys.exit(1 if bad else 0)
