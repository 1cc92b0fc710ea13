"""whole solves of instances with large blocks / clusters at several limb counts (the kernels of the large-problem paths at K != 5)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import flat
from clrs_amd.mw import solvesdp_mw
for name, expect, kw in (("ns_8_15_2", 0.25366950790104804, {}), ("threepoint_3_8_8", 12.5227962013944, dict(omega_p=1e3, omega_d=1e3)), ("sdpa_x64", -125.091980229314, {})):
    f = flat(name)
    for K in (4, 6, 8):
        thr = dict(dual_error_threshold=1e-25, primal_error_threshold=1e-25, duality_gap_threshold=1e-12) if K == 4 else {}
        try:
            r = solvesdp_mw(f, limbs=K, **kw, **thr)
        except Exception as e:
            print(name, 'limbs', K, 'refused:', str(e)[-90:], flush=True)
            continue
        print(name, "limbs", K, r.status, r.error_code, r.iterations, "%.15g" % r.primal_objective, "|diff| %.2e" % abs(r.primal_objective - expect), "%.3f s" % r.time_total, flush=True)
