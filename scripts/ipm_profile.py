import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, clrs_amd
from tests.util import flat
from clrs_amd.solver import solvesdp_device, SchurContext
name = sys.argv[1] if len(sys.argv) > 1 else "polyopt40"
f = flat(name)
for kv in sys.argv[2:]:      # library configuration keys, e.g. ipm_wmfma=1
    from clrs_amd import _lib
    key, val = kv.split("=")
    _lib.check(_lib.load().clrs_config_set(key.encode(), int(val)))
ctx = SchurContext(f)
solvesdp_device(f, ctx=ctx)
t = time.time(); n = 0
for _ in range(5):
    r = solvesdp_device(f, ctx=ctx); n += r.iterations
dt = time.time() - t
print(name, "iterations", r.iterations, "status", r.status, "%.0f it/s  (%.1f us per iteration)" % (n / dt, 1e6 * dt / n))
