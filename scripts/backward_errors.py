"""Measure the KKT backward errors of the multi-word solve stage over the parity-test instances (sets the tolerances of tests/test_mw_parity.py)."""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.util import flat, mw_with_tails, mw_relerr
from tests.test_mw_parity import _iterates, _sym_limbs, NAMES
from clrs_amd.mw import MwSchurContext
from oracle.oracle import Oracle

pad = lambda a: np.vstack([a, np.zeros((1, a.shape[1]))])
for name in NAMES:
    f = flat(name)
    for K in (2, 3, 4, 5, 6, 8, 10):
        if K > 5 and name not in ("ce_8_15", "threepoint_4", "sdpa_small", "ns_8_15_2"):
            continue
        X, Y = _iterates(f, K); X, Y = _sym_limbs(f, X), _sym_limbs(f, Y)
        o = Oracle(f, mp_bits=320 if K <= 5 else 640)
        ctx = MwSchurContext(f, limbs=K)
        Xc = ctx.cholesky_blocks(X)
        S, _ = ctx.compute_S_integrated(Xc, Y)
        S_ref, _ = o.schur_assemble_mw(pad(Xc), pad(Y))
        st = ctx.factor()
        if st:
            print(name, K, "factor status", st); ctx.close(); continue
        rng = np.random.default_rng(5)
        rx, ry = mw_with_tails(rng.standard_normal(f.x_len), K, 1), mw_with_tails(rng.standard_normal(max(f.n_free, 1)), K, 2)[:, :f.n_free]
        dx, dy = ctx.solve(rx, ry)
        ex, ey = o.kkt_backward_error_mw(S_ref, dx, dy, rx, ry)
        o.set_S_mw(S_ref); o.schur_factor()
        dxr, dyr = o.schur_solve_mw(pad(rx), pad(ry) if f.n_free else np.zeros((K + 1, 0)))
        exo, eyo = o.kkt_backward_error_mw(S_ref, dxr, dyr, rx, ry)
        fe = mw_relerr(dx, dxr)
        lg = lambda v: (math.log2(v) if v > 0 else -9999)
        print("%-14s K=%2d  backward x 2^%.0f y 2^%.0f (53K-x = %.0f / %.0f)  oracle's own 2^%.0f 2^%.0f  forward dx 2^%.0f" %
              (name, K, lg(ex), lg(ey), 53 * K + lg(ex), 53 * K + lg(ey), lg(exo), lg(eyo), lg(fe)), flush=True)
        ctx.close()
