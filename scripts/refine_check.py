"""Iterative refinement of the multi-word solve stage (k_mw_refine): KKT backward errors with and without it on the ill-conditioned parity
instances, and the whole solves whose outcome depends on it (Nsphere_packing(8,15,[1/2,1/2,1/2]) at 5 limbs; cohnelkies(8,15) timing)."""
import math, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import clrs_amd
from clrs_amd import _lib
from tests.util import flat, mw_with_tails, mw_relerr
from tests.test_mw_parity import _iterates, _sym_limbs
from clrs_amd.mw import MwSchurContext, solvesdp_mw
from oracle.oracle import Oracle

L = _lib.load()
pad = lambda a: np.vstack([a, np.zeros((1, a.shape[1]))])
lg = lambda v: (math.log2(v) if v > 0 else -9999)
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["ce_8_15", "ns_8_15_2", "polyopt40", "threepoint_4"]
for name in names:
    f = flat(name)
    for K in (4, 5, 6):
        X, Y = _iterates(f, K); X, Y = _sym_limbs(f, X), _sym_limbs(f, Y)
        o = Oracle(f, mp_bits=320 if K <= 5 else 640)
        for refine in (0, 1, 2):
            L.clrs_config_set(b"mw_refine", refine)
            ctx = MwSchurContext(f, limbs=K)
            L.clrs_config_set(b"mw_refine", 1)
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
            print("%-14s K=%2d refine=%d  backward x 2^%.0f y 2^%.0f (bits lost %.0f / %.0f)" % (name, K, refine, lg(ex), lg(ey), 53 * K + lg(ex), 53 * K + lg(ey)), flush=True)
            ctx.close()

from clrs_amd.problems import nsphere_packing, cohnelkies
for label, sdp, Ks in (("cohnelkies(8,15)", cohnelkies(8, 15), (5,)), ("ns3", nsphere_packing(8, 15, [0.5, 0.5, 0.5]), (5, 6))):
    f = clrs_amd.flatten(sdp)
    for K in Ks:
        for refine in (0, 1, 2):
            L.clrs_config_set(b"mw_refine", refine)
            solvesdp_mw(f, limbs=K, maxiterations=2)
            r = solvesdp_mw(f, limbs=K)
            L.clrs_config_set(b"mw_refine", 1)
            print("%-18s K=%d refine=%d: %s code %d, %d iterations, objective %.14g gap %.3g errors %.3g %.3g, %.3f s = %.3f ms per iteration" % (
                label, K, refine, r.status, r.error_code, r.iterations, r.primal_objective, r.duality_gap, r.dual_error, r.primal_error, r.time_total,
                1e3 * r.time_total / max(r.iterations, 1)), flush=True)
