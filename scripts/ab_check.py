"""A / B on one box: whole solves of one instance alternating between two settings of a clrs_config_set knob that contexts read when they (or their
iteration state) are created.  usage: ab_check.py <knob> <value A> <value B> [instance] [limbs] [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.util import flat
from clrs_amd import _lib
from clrs_amd.mw import MwSchurContext, solvesdp_mw

knob, va, vb = sys.argv[1].encode(), int(sys.argv[2]), int(sys.argv[3])
name = sys.argv[4] if len(sys.argv) > 4 else "ce_8_15"
K = int(sys.argv[5]) if len(sys.argv) > 5 else 5
rounds = int(sys.argv[6]) if len(sys.argv) > 6 else 12
L = _lib.load()
f = flat(name)
ctx = {}
for v in (va, vb):
    _lib.check(L.clrs_config_set(knob, v))
    ctx[v] = MwSchurContext(f, limbs=K)
    solvesdp_mw(f, ctx=ctx[v], limbs=K)              # (creates the iteration state under this setting)
times = {va: [], vb: []}
res = {}
for r in range(rounds):
    for v in (va, vb):
        x = solvesdp_mw(f, ctx=ctx[v], limbs=K)
        times[v].append(1e3 * x.time_total / x.iterations)
        res[v] = x
for v in (va, vb):
    t = np.sort(times[v])
    print("%s = %d: %d iterations %s obj %.15g  ms/iteration min %.4f median %.4f max %.4f" % (knob.decode(), v, res[v].iterations, res[v].status, res[v].primal_objective, t[0], t[len(t) // 2], t[-1]))
