import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, clrs_amd
from tests.util import flat
from clrs_amd.solver import solvesdp, solvesdp_device
for name in sys.argv[1:] or ["x2p1", "polyopt8", "polyopt40", "delsarte_8_3", "delsarte_3_10", "ns_8_3_2", "sdpa_example", "threepoint_4"]:
    f = flat(name)
    kw = dict(omega_p=1e3, omega_d=1e3) if name == "threepoint_4" else (dict(omega_p=1e2, omega_d=1e2) if name == "sdpa_example" else {})
    t = time.time(); rd = solvesdp_device(f, maxiterations=200, **kw); td = time.time() - t
    t = time.time(); rh = solvesdp(f, maxiterations=200, **kw); th = time.time() - t
    print(f"{name:14s} device: {rd.status:12s} code {rd.error_code} it {rd.iterations:3d} p {rd.primal_objective:.10g} d {rd.dual_objective:.10g} gap {rd.duality_gap:.1e} {rd.iterations/td:7.0f} it/s | "
          f"host loop: {rh.status:12s} code {rh.error_code} it {rh.iterations:3d} p {rh.primal_objective:.10g} {rh.iterations/th:6.0f} it/s")
    n = min(4, len(rd.history), len(rh.history))
    print("   mu   dev", rd.history[:n, 1], " host", rh.history[:n, 1])
    print("   a_d  dev", rd.history[:n, 8], " host", rh.history[:n, 8])
