"""Where the time of whole solves back to back goes besides the iterations (bench.py's timed region = whole solves of cohnelkies(8,15)): wall time per solve against
the library's loop time, cProfile of ten solves."""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import flat
from clrs_amd.mw import MwSchurContext, solvesdp_mw
f = flat("ce_8_15")
thr = dict(dual_error_threshold=1e-30, primal_error_threshold=1e-30, duality_gap_threshold=1e-15)
ctx = MwSchurContext(f, limbs=5)
for _ in range(3):
    r = solvesdp_mw(f, ctx=ctx, **thr)
t0 = time.perf_counter()
loop = 0.0
n = 20
for _ in range(n):
    r = solvesdp_mw(f, ctx=ctx, **thr)
    loop += r.time_total
wall = time.perf_counter() - t0
print(f"{n} solves of {r.iterations} iterations: wall {1e3 * wall / n:.3f} ms per solve, of which the device loop {1e3 * loop / n:.3f} ms -> {1e3 * (wall - loop) / n:.3f} ms around it "
      f"= {100 * (wall - loop) / wall:.1f} %; {1e3 * wall / n / r.iterations:.4f} ms per iteration by the wall clock, {1e3 * loop / n / r.iterations:.4f} by the loop's")
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    solvesdp_mw(f, ctx=ctx, **thr)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14)
print("\n".join(s.getvalue().splitlines()[:40]))
ctx.close()
