"""Rate of the hot-path step through the HOST-pointer entry points (every call copies its arguments over PCIe and synchronises)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench, clrs_amd
from clrs_amd.solver import SchurContext
flat = bench.build_problem(1)
X, Y = bench.seeded_iterates(flat, seed=1)
rng = np.random.default_rng(2)
rx, ry = rng.standard_normal(flat.x_len), rng.standard_normal(flat.n_free)
ctx = SchurContext(flat)
def step():
    Xc = ctx.cholesky_blocks(X)
    ctx.compute_S_integrated(Xc, Y, want_S=False, want_AY=True)
    ctx.factor()
    ctx.solve(rx, ry); ctx.solve(rx, ry)
for _ in range(20): step()
t = time.perf_counter()
K = 300
for _ in range(K): step()
dt = time.perf_counter() - t
print("host-pointer API: %.1f us per step -> %.0f steps/s" % (1e6 * dt / K, K / dt))
