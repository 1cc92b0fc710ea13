"""Mixed-precision refinement inside the interior-point solve (clrs_mw_options.factor_limbs): every instance solved with the factor stage and the solve
stage's products in all limbs and in the automatic (reduced, adaptive) form -- iterations, objectives, errors, time per iteration, and the per-iteration
first-pass accuracy (refine_bits).  usage: kf_check.py [limbs] [names...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.util import flat, load_flat
from clrs_amd.mw import solvesdp_mw

args = [a for a in sys.argv[1:]]
K = int(args.pop(0)) if args and args[0].isdigit() else 5
names = args or ["ce_8_15", "min_f_2", "delsarte_3_10", "delsarte_8_3", "polyopt40", "threepoint_4", "sdpa_example"]
KW = {"threepoint_4": dict(omega_p=1e3, omega_d=1e3)}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in names:
    f = load_flat(os.path.join(root, "tests", "golden", "min_f_2.npz"))[0] if name == "min_f_2" else flat(name)
    kw = KW.get(name, {})
    res = {}
    for label, fl in (("all", K), ("auto", None)):
        solvesdp_mw(f, limbs=K, maxiterations=2, factor_limbs=fl, **kw)
        res[label] = min((solvesdp_mw(f, limbs=K, factor_limbs=fl, **kw) for _ in range(3)), key=lambda r_: r_.time_total)
    a, b = res["all"], res["auto"]
    rb = b.timings["refine_bits"]
    low = sum(1 for i, v in enumerate(rb))
    print("%-14s K=%d  all: %3d it %s obj %.15g gap %.2e err %.1e/%.1e %.3f ms/it | auto: %3d it %s obj %.15g gap %.2e err %.1e/%.1e %.3f ms/it  (x%.3f)  rel obj diff %.1e"
          % (name, K, a.iterations, a.status, a.primal_objective, a.duality_gap, a.dual_error, a.primal_error, 1e3 * a.time_total / max(a.iterations, 1),
             b.iterations, b.status, b.primal_objective, b.duality_gap, b.dual_error, b.primal_error, 1e3 * b.time_total / max(b.iterations, 1),
             a.time_total / b.time_total * b.iterations / max(a.iterations, 1), abs(a.primal_objective - b.primal_objective) / max(1.0, abs(a.primal_objective))), flush=True)
    print("    refine_bits all :", a.timings["refine_bits"])
    print("    refine_bits auto:", rb, flush=True)
    hd = np.max(np.abs(a.history[:min(len(a.history), len(b.history)), 1:] - b.history[:min(len(a.history), len(b.history)), 1:]) / (np.abs(a.history[:min(len(a.history), len(b.history)), 1:]) + 1e-300))
    print("    largest relative difference of the table rows (mu, objectives, errors, step lengths, beta_c): %.2e" % hd, flush=True)
