"""Relative errors (log2) of the multi-word HIP path on the trajectory fixture, per limb count and iteration (calibrates the test tolerances)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.util import flat, mw_relerr
from clrs_amd.mw import MwSchurContext
f = flat("ce_8_15")
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ce_8_15_traj.npz"))
for K in (3, 4, 5):
    ctx = MwSchurContext(f, limbs=K)
    for s, it in enumerate(g["iters"]):
        X, Y = np.ascontiguousarray(g["X"][s][:K]), np.ascontiguousarray(g["Y"][s][:K])
        try:
            Xc = ctx.cholesky_blocks(X)
        except Exception as e:
            print(K, it, "chol X failed"); continue
        S, _ = ctx.compute_S_integrated(Xc, Y)
        eS = mw_relerr(S, g["S"][s]); st = ctx.factor()
        if st:
            print("K=%d it=%d mu=%.1e  S 2^%.1f  factor status %d" % (K, it, g["mu"][s], np.log2(eS), st)); continue
        dx, dy = ctx.solve(np.ascontiguousarray(g["rhs_x"][s][:K]), np.ascontiguousarray(g["rhs_y"][s][:K]))
        print("K=%d it=%d mu=%.1e  S 2^%.1f  dx 2^%.1f  dy 2^%.1f" % (K, it, g["mu"][s], np.log2(eS), np.log2(mw_relerr(dx, g["dx"][s])), np.log2(mw_relerr(dy, g["dy"][s]))))
    ctx.close()
