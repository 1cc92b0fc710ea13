"""Side-by-side iteration table of the multi-word device loop and the multi-precision oracle loop (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.util import flat
from clrs_amd.mw import solvesdp_mw
from oracle.oracle import Oracle

name = sys.argv[1] if len(sys.argv) > 1 else "polyopt8"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
nit = int(sys.argv[3]) if len(sys.argv) > 3 else 12
f = flat(name)
r = solvesdp_mw(f, limbs=K, maxiterations=nit)
ro = Oracle(f, mp_bits=256).solvesdp(maxiterations=nit)
np.set_printoptions(linewidth=250, precision=6)
cols = ["iter", "mu", "d_obj", "p_obj", "gap", "P_err", "p_err", "d_err", "alpha_d", "alpha_p", "beta"]
print(" " * 4 + " ".join("%12s" % c for c in cols))
for i in range(min(len(r.history), len(ro["hist"]))):
    print("gpu " + " ".join("%12.5e" % v for v in r.history[i]))
    print("ora " + " ".join("%12.5e" % v for v in ro["hist"][i]))
print("gpu", r.status, r.error_code, r.iterations, r.primal_objective, r.dual_objective)
print("ora", ro["error_code"], ro["iterations"], ro["p_obj"], ro["d_obj"])
