import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clrs_amd
from clrs_amd.problems import polyopt_random
from clrs_amd.solver import solvesdp_device, SchurContext
d = int(sys.argv[1]) if len(sys.argv) > 1 else 20
f = clrs_amd.flatten(polyopt_random(d, seed=0)[0])
ctx = SchurContext(f)
solvesdp_device(f, ctx=ctx)
t = time.time(); n = 0
for _ in range(5):
    r = solvesdp_device(f, ctx=ctx); n += r.iterations
dt = time.time() - t
print("polyopt_random(%d): n %s P %s" % (d, list(f.block_n), list(f.cluster_P)), "iterations", r.iterations, "status", r.status, "%.0f it/s  (%.1f us per iteration)" % (n / dt, 1e6 * dt / n))
