"""Whole interior-point solves of cohnelkies(8,15) at 5 limbs, for `rocprofv3 --kernel-trace`: the trace of the last solve is what
scripts/iter_timeline.py turns into a per-iteration timeline (kernel, stream, start offset, duration, gap)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import flat
from clrs_amd.mw import MwSchurContext, solvesdp_mw

name = sys.argv[1] if len(sys.argv) > 1 else "ce_8_15"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
f = flat(name)
ctx = MwSchurContext(f, limbs=5)
for _ in range(reps):
    t = time.time()
    r = solvesdp_mw(f, ctx=ctx)
    dt = time.time() - t
    print(name, r.status, r.iterations, "%.4f s, %.1f us / iteration (host loop %.4f s)" % (dt, 1e6 * r.time_total / r.iterations, r.time_total))
ctx.close()
