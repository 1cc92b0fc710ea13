"""Device-resident IPM vs the CPU oracle loop as the problem grows (univariate polynomial optimisation, degree 2d)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, clrs_amd
from clrs_amd.problems import polyopt_random, delsarte
from clrs_amd.solver import solvesdp_device, SchurContext
from oracle.oracle import Oracle
cases = [("polyopt d=%d" % d, lambda d=d: clrs_amd.flatten(polyopt_random(d, seed=0)[0])) for d in (10, 20, 30, 40, 47)]
cases += [("delsarte(3,%d)" % d, lambda d=d: clrs_amd.flatten(delsarte(3, d, 0.5))) for d in (10, 20, 30)]
for label, mk in cases:
    f = mk()
    try:
        ctx = SchurContext(f)
        solvesdp_device(f, ctx=ctx)
        t = time.perf_counter(); its = 0
        for _ in range(3):
            r = solvesdp_device(f, ctx=ctx); its += r.iterations
        gpu = its / (time.perf_counter() - t)
        ctx.close()
        st = f"{r.status} it {r.iterations} obj {r.primal_objective:.8g}"
    except Exception as e:
        gpu, st = float("nan"), repr(e)[:60]
    best = 0
    for thr in (1, 8):
        o = Oracle(f, quad=False); o.set_num_threads(thr)
        t = time.perf_counter(); n = 0
        while time.perf_counter() - t < 1.0:
            ro = o.solvesdp(omega_p=1e4, omega_d=1e4, duality_gap_threshold=1e-7, dual_error_threshold=1e-9, primal_error_threshold=1e-9); n += ro["iterations"]
        best = max(best, n / (time.perf_counter() - t))
    print(f"{label:16s} n={int(max(f.block_n)):3d} P={int(sum(f.cluster_P)):4d}  device {gpu:8.0f} it/s   cpu port {best:8.0f} it/s   ratio {gpu/best:5.2f}   [{st}; oracle obj {ro['p_obj']:.8g} code {ro['error_code']}]")
