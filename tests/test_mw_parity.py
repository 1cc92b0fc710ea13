"""Parity of the multi-word (extended precision) HIP path with the multi-precision CPU oracle.

The oracle computes at 320 bits (oracle/mpx.hpp, the stand-in for the reference's Arb midpoints); the HIP path at K limbs of
fp64 (about 53 K - K bits).  Tolerances are stated per test as 2^-(53 K - slack): `slack` covers the length of the dot
products and the conditioning of the seeded iterates (cond ~ 10), nothing else."""
import os

import numpy as np
import pytest

from tests.util import (chol_blocks_np, flat, mw_diff, mw_from_double, mw_relerr, mw_with_tails, spd_iterates)

pytestmark = pytest.mark.gpu

NAMES = ["polyopt8", "ce_8_3", "ce_8_15", "ns_8_3_2", "ns_8_15_2", "delsarte_3_10", "threepoint_4", "sdpa_small", "polyopt40"]


@pytest.fixture(scope="module")
def oracle_built():
    from oracle import oracle
    oracle.build()


def _iterates(f, K, seed=1):
    X, Y = spd_iterates(f, seed=seed)
    return mw_with_tails(X, K, seed=seed + 10), mw_with_tails(Y, K, seed=seed + 20)


def _sym_limbs(f, M):
    """make every block of a planar xy-layout array exactly symmetric, limb by limb"""
    M = M.copy()
    for b in range(f.n_blocks):
        n = int(f.block_n[b]); sl = slice(int(f.block_off[b]), int(f.block_off[b + 1]))
        for l in range(M.shape[0]):
            A = M[l, sl].reshape(n, n, order="F")
            A = np.tril(A) + np.tril(A, -1).T
            M[l, sl] = A.reshape(-1, order="F")
    return M


def tol(K, slack):
    # an operation carries about 53 K - K bits: the slacks below were set at K <= 5; one more bit per limb beyond that
    return 2.0 ** (-(53 * K - slack - max(0, K - 5)))


# Backward error of the solve stage: bits lost against 53 K (x rows, y rows), measured over every instance and limb count
# (scripts/refine_check.py, scripts/backward_errors.py; gpurun_out/r4_refine_b.log).  The solve stage multiplies with explicit inverse factors
# (DESIGN.md section 5.5), whose residual grows with cond(L_j), cond(L_Q): 24 / 57 bits on cohnelkies(8,15), 32 / 82 on Nsphere_packing(8,15) --
# and then takes ONE step of iterative refinement against the assembled S_j and B (k_mw_refine, k_mw_solve_bwd MODE 1 / 2), after which every
# instance is within 4 bits of 53 K, like the substitutions of the reference (src/solver.jl:1538, 1557, 1567-1572; the 320-bit oracle's own:
# 0-3 / 30-35 bits -- it does not refine its y rows).  `BACKWARD_SLACK_UNREFINED` documents what clrs_config_set("mw_refine", 0) gives.
BACKWARD_SLACK = {}
BACKWARD_SLACK_UNREFINED = {"ce_8_15": (32, 68), "ns_8_15_2": (38, 96)}


def assert_backward_stable(o, S_ref, dx, dy, rx, ry, K, name=None, slack=None):
    """(dx, dy) solve [S -B; B^T 0](dx; dy) = (rx; ry) (src/solver.jl:1527) with S the ORACLE's matrix: residuals formed by the
    oracle at 320 / 640 bits, relative to |S||dx| + |B||dy| + |rhs| -- free of the conditioning of S, unlike a forward comparison."""
    sx, sy = slack if slack is not None else BACKWARD_SLACK.get(name, (16, 16))
    bx, by = o.kkt_backward_error_mw(S_ref, dx, dy, rx, ry)
    assert tol(K, max(sx, sy)) < 1e-6
    assert bx <= tol(K, sx), ("backward error, x rows", name, K, np.log2(max(bx, 1e-300)))
    assert by <= tol(K, sy), ("backward error, y rows", name, K, np.log2(max(by, 1e-300)))


def assert_forward_close(a, b, bound, what):
    """forward comparison with the oracle's solution -- only where the bound still says something"""
    if bound < 1e-6:
        assert mw_relerr(a, b) <= bound, (what, mw_relerr(a, b))


@pytest.mark.parametrize("K,DL", [(2, 1), (2, 2), (3, 2), (4, 1), (4, 2), (5, 2), (6, 2), (8, 2), (10, 2)])
@pytest.mark.parametrize("name", NAMES)
def test_mw_assemble_factor_solve_match_oracle(name, K, DL, oracle_built):
    """K limbs per computed number; DL limbs of problem data (1: the fp64 roundings, 2: the (hi, lo) pairs of the FlatSDP)."""
    from clrs_amd.mw import MwSchurContext
    from oracle.oracle import Oracle
    if name in ("ns_8_15_2",) and K in (2, 3):
        pytest.skip("covered at K = 4, 5")
    if K > 5 and name not in ("ce_8_15", "threepoint_4", "sdpa_small", "ns_8_15_2"):
        pytest.skip("6, 8 and 10 limbs (checked against the 640-bit build of the oracle): the north-star shapes, a mixed and a dense instance, the blocked path")
    f = flat(name)
    X, Y = _iterates(f, K)
    X, Y = _sym_limbs(f, X), _sym_limbs(f, Y)
    o = Oracle(f, mp_bits=320 if K <= 5 else 640, use_lo=(DL == 2))
    ctx = MwSchurContext(f, limbs=K, data_limbs=DL)
    # Cholesky of the X blocks
    Xc = ctx.cholesky_blocks(X)
    st, Xc_ref = o.cholesky_blocks_mw(np.vstack([X, np.zeros((1, f.xy_len))]))
    assert st == 0
    assert mw_relerr(Xc, Xc_ref) <= tol(K, 14), ("chol X", mw_relerr(Xc, Xc_ref))
    # assembly from the SAME factors
    S, AY = ctx.compute_S_integrated(Xc, Y)
    S_ref, AY_ref = o.schur_assemble_mw(np.vstack([Xc, np.zeros((1, f.xy_len))]), np.vstack([Y, np.zeros((1, f.xy_len))]))
    assert mw_relerr(S, S_ref) <= tol(K, 22), ("S", mw_relerr(S, S_ref))
    if f.n_terms:
        assert mw_relerr(AY, AY_ref, scale=max(1.0, np.max(np.abs(AY_ref[0])))) <= tol(K, 16)
    # exact symmetry (symmetric!, src/tools.jl:43-57)
    for j in range(f.n_clusters):
        P = int(f.cluster_P[j]); o0 = int(f.S_off[j])
        for l in range(K):
            Sj = S[l, o0:o0 + P * P].reshape(P, P, order="F")
            assert np.array_equal(Sj, Sj.T)
    # factorisation of the SAME S (the well-conditioned ones; the 2d = 30 sphere-packing S is the subject of the next test)
    o.set_S_mw(np.vstack([S, np.zeros((1, f.S_len))]))
    st_ref = o.schur_factor()
    st_gpu = ctx.factor()
    if name in ("ce_8_15", "ns_8_15_2") and K == 2:
        assert st_gpu == st_ref or st_gpu > 0
        ctx.close()
        return
    assert st_ref == 0 and st_gpu == 0, (st_ref, st_gpu)
    L, LinvB, LQ = ctx.get_factor()
    L_ref, LinvB_ref, LQ_ref = o.get_factor_mw(K + 1)
    # conditioning of S enters the factor: cond(S) is up to ~1e20 on the sphere-packing instances
    amp = {"ce_8_15": 70, "ns_8_15_2": 83, "polyopt40": 40, "threepoint_4": 40}.get(name, 30)
    assert mw_relerr(L, L_ref) <= tol(K, 22 + amp), ("L", mw_relerr(L, L_ref))
    if f.n_free:
        assert mw_relerr(LinvB, LinvB_ref) <= tol(K, 22 + amp), ("LinvB", mw_relerr(LinvB, LinvB_ref))
        assert mw_relerr(LQ, LQ_ref) <= tol(K, 22 + 2 * amp), ("LQ", mw_relerr(LQ, LQ_ref))
    rng = np.random.default_rng(5)
    rx, ry = mw_with_tails(rng.standard_normal(f.x_len), K, 1), mw_with_tails(rng.standard_normal(max(f.n_free, 1)), K, 2)[:, :f.n_free]
    dx, dy = ctx.solve(rx, ry)
    assert_backward_stable(o, S_ref, dx, dy, rx, ry, K, name)
    dx_ref, dy_ref = o.schur_solve_mw(np.vstack([rx, np.zeros((1, f.x_len))]), np.vstack([ry, np.zeros((1, f.n_free))]) if f.n_free else np.zeros((K + 1, 0)))
    assert_forward_close(dx, dx_ref, tol(K, 22 + 3 * amp), "dx")
    if f.n_free:
        assert_forward_close(dy, dy_ref, tol(K, 22 + 3 * amp), "dy")
    ctx.close()


@pytest.mark.parametrize("name", ["ce_8_15", "ns_8_15_2"])
@pytest.mark.parametrize("K", [4, 5, 6])
def test_refinement_gives_the_solve_stage_the_backward_error_of_substitutions(name, K, oracle_built):
    """The reference's solves are substitutions (src/solver.jl:1538, 1557, 1567-1572): residuals at the working accuracy.  Products with explicit
    inverse factors alone lose cond(L) there (refine = 0: the y rows of these two instances lose more than 40 bits -- the assertion that shows
    this test can fail); with the refinement step (the default, correction in all K limbs, and refine = 2, correction in mw_kc(K) limbs) both row
    sets are within 12 bits of 53 K.  cohnelkies(8,15): LDS-resident one-workgroup path; Nsphere_packing(8,15,.,2): blocked factorisation, row-parallel solve."""
    from clrs_amd.mw import MwSchurContext
    from oracle.oracle import Oracle
    f = flat(name)
    X, Y = _iterates(f, K)
    X, Y = _sym_limbs(f, X), _sym_limbs(f, Y)
    o = Oracle(f, mp_bits=320 if K <= 5 else 640)
    pad = lambda a: np.vstack([a, np.zeros((1, a.shape[1]))])
    rng = np.random.default_rng(5)
    rx, ry = mw_with_tails(rng.standard_normal(f.x_len), K, 1), mw_with_tails(rng.standard_normal(f.n_free), K, 2)
    lost = {}
    for refine in (0, 1, 2):
        ctx = MwSchurContext(f, limbs=K, refine=refine)
        Xc = ctx.cholesky_blocks(X)
        ctx.compute_S_integrated(Xc, Y)
        S_ref, _ = o.schur_assemble_mw(pad(Xc), pad(Y))
        assert ctx.factor() == 0
        dx, dy = ctx.solve(rx, ry)
        bx, by = o.kkt_backward_error_mw(S_ref, dx, dy, rx, ry)
        lost[refine] = (53 * K + np.log2(bx), 53 * K + np.log2(by))
        ctx.close()
    assert lost[0][1] > 40, lost
    assert max(lost[1]) <= 12 and max(lost[2]) <= 12, lost


def test_stream_words_and_events_give_the_same_solve(oracle_built):
    """clrs_config_set("mw_stream_words", 0): the two streams of the iteration synchronise through events only (what counter-collection runs use, and what
    sharded contexts always do).  Same kernels, same order of operations: histories and iterates are bit-identical to the default (words)."""
    from clrs_amd import _lib
    from clrs_amd.mw import MwSchurContext, solvesdp_mw
    L = _lib.load()
    f = flat("ce_8_3")
    out = []
    for words in (1, 0):
        _lib.check(L.clrs_config_set(b"mw_stream_words", words))
        try:
            ctx = MwSchurContext(f, limbs=4)
        finally:
            L.clrs_config_set(b"mw_stream_words", 1)
        r = solvesdp_mw(f, ctx=ctx, limbs=4, duality_gap_threshold=1e-10, dual_error_threshold=1e-20, primal_error_threshold=1e-20)
        ctx.close()
        assert r.error_code == 0 and r.status == "Optimal"
        out.append(r)
    a, b = out
    assert a.iterations == b.iterations and np.array_equal(a.history, b.history)
    assert np.array_equal(a.x, b.x) and np.array_equal(a.y, b.y) and np.array_equal(a.X, b.X) and np.array_equal(a.Y, b.Y)


def test_verbose_table_rows_come_from_the_callback_of_the_one_call_loop(capsys, oracle_built):
    """clrs_mw_ipm_solve_cb: `verbose` prints the reference's table rows (src/solver.jl:566-582) from a callback per record while the loop stays the
    one-call loop (iterations enqueued one ahead, termination on the device): as many rows as iterations, the row of iteration k carrying the objectives
    of the iterate it started from, and bit for bit the solve of the silent call."""
    from clrs_amd.mw import MwSchurContext, solvesdp_mw
    f = flat("ce_8_3")
    kw = dict(limbs=4, duality_gap_threshold=1e-10, dual_error_threshold=1e-20, primal_error_threshold=1e-20)
    ctx = MwSchurContext(f, limbs=4)
    a = solvesdp_mw(f, ctx=ctx, **kw)
    capsys.readouterr()
    b = solvesdp_mw(f, ctx=ctx, verbose=True, **kw)
    out = [l for l in capsys.readouterr().out.splitlines() if l.strip() and l.split()[0].isdigit()]
    ctx.close()
    assert a.status == b.status == "Optimal" and np.array_equal(a.history, b.history) and np.array_equal(a.y, b.y)
    assert len(out) == b.iterations and [int(l.split()[0]) for l in out] == list(range(1, b.iterations + 1))
    for k, l in enumerate(out):
        cols = l.split()
        assert abs(float(cols[2]) - b.history[k, 1]) <= 1e-3 * abs(b.history[k, 1])                       # mu
        assert abs(float(cols[3]) - b.history[k, 2]) <= 1e-3 * abs(b.history[k, 2]) + 1e-300             # D-obj of the iterate the iteration started from


def test_matmul_prec_reduces_the_pairing_products_as_the_oracle_does(oracle_built):
    """The reference's `matmul_prec` keyword (src/solver.jl:125, 304, 312-313, 1125-1143): the products that form the pairing matrices at fewer bits than the
    rest.  clrs_mw_options.matmul_limbs = 4 at 5 limbs (~209 bits): S_j agrees with the oracle whose pairing products run at 212 bits as closely as two
    209-bit computations can, and is measurably further from the oracle at full precision; a whole solve with the reference's default thresholds gets at least as far as the
    oracle's with the same matmul_prec (which ends in a SolverFailure near gap 1e-14 where its full-precision solve ends Optimal)."""
    from clrs_amd.mw import MwSchurContext, solvesdp_mw
    from oracle.oracle import Oracle
    f = flat("ce_8_15")
    K = 5
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ce_8_15_traj.npz"))
    X, Y = z["X"][2][:K], z["Y"][2][:K]                       # iterate of iteration 28
    pad = lambda a: np.vstack([a, np.zeros((1, a.shape[1]))])
    ctx = MwSchurContext(f, limbs=K, matmul_limbs=4)
    Xc = ctx.cholesky_blocks(X)
    S, _ = ctx.compute_S_integrated(Xc, Y)
    ctx.close()
    S_red, _ = Oracle(f, mp_bits=320, matmul_bits=212).schur_assemble_mw(pad(Xc), pad(Y))
    S_full, _ = Oracle(f, mp_bits=320).schur_assemble_mw(pad(Xc), pad(Y))
    e_red, e_full, e_orc = mw_relerr(S, S_red), mw_relerr(S, S_full), mw_relerr(S_red[:K], S_full)
    assert e_orc > 2.0 ** -225, np.log2(e_orc)                     # the oracle's own reduction is visible at this iterate ...
    assert 2.0 ** -225 < e_full < 2.0 ** -150, np.log2(e_full)     # ... and so is the GPU's, at the same order
    assert e_red < 2.0 ** -150 and abs(np.log2(e_full) - np.log2(e_orc)) < 30, (np.log2(e_red), np.log2(e_full), np.log2(e_orc))
    ro = Oracle(f, mp_bits=256, matmul_bits=212).solvesdp()
    rg = solvesdp_mw(f, limbs=K, matmul_prec=212)
    # the oracle forms X^-1 explicitly and rounds X^-1 V to matmul_prec (the reference's method 3, :1117-1143: errors relative to |X^-1| |V|); the HIP path
    # rounds Z = chol(X)^-1 V and takes V^T X^-1 V = Z^T Z (errors relative to the Gram matrix itself), so at the same matmul_prec its S_j is at least as
    # good: the oracle ends with a failed factorisation at gap 4e-14, the HIP solve goes at least as far
    assert ro["error_code"] == 1 and ro["gap"] < 1e-9
    assert rg.duality_gap <= 10 * ro["gap"] and abs(rg.primal_objective - ro["p_obj"]) <= 1e-9, (rg.status, rg.error_code, rg.duality_gap, ro["gap"])


def test_correctoronly_follows_the_oracle(oracle_built):
    """The reference's `correctoronly` keyword (src/solver.jl:121, 370-374, 945): mu_p = mu in the predictor's residual and no "optimal" termination -- the loop
    ends on need_dual_feasible / need_primal_feasible (or an error, or maxiterations).  Same iteration count and iterate as the oracle with the same keyword;
    and without a feasibility target the loop runs to maxiterations (code 2) where the default loop would have stopped as optimal."""
    from clrs_amd.mw import solvesdp_mw
    from oracle.oracle import Oracle
    f = flat("ce_8_15")
    kw = dict(need_dual_feasible=True, dual_error_threshold=1e-20)
    r = solvesdp_mw(f, limbs=5, correctoronly=True, **kw)
    ro = Oracle(f, mp_bits=256).solvesdp(correctoronly=1, need_dual_feasible=1, dual_error_threshold=1e-20)
    rd = solvesdp_mw(f, limbs=5, **kw)
    # from the reference's starting point mu_p = mu leaves this problem no room: both stop after the same few iterations with "step length too small" (code 4),
    # where the default loop reaches dual feasibility in 27 iterations
    assert r.error_code == ro["error_code"] and r.iterations == ro["iterations"], (r.error_code, ro["error_code"], r.iterations, ro["iterations"])
    n = min(len(r.history), len(ro["hist"]))
    assert n >= 2 and np.allclose(r.history[:n, 1], ro["hist"][:n, 1], rtol=1e-6)                  # mu per iteration
    assert rd.error_code == 0 and rd.iterations > r.iterations and not np.allclose(r.history[:2, 1], rd.history[:2, 1], rtol=1e-3)
    g = flat("ce_8_3")
    kw3 = dict(duality_gap_threshold=1e-10, dual_error_threshold=1e-20, primal_error_threshold=1e-20)
    r2 = solvesdp_mw(g, limbs=4, correctoronly=True, maxiterations=40, **kw3)
    o2 = Oracle(g, mp_bits=212).solvesdp(correctoronly=1, maxiterations=40, **kw3)
    assert r2.error_code == o2["error_code"] and r2.error_code != 0 and abs(r2.iterations - o2["iterations"]) <= 1, (r2.error_code, o2["error_code"], r2.iterations, o2["iterations"])


def test_refined_predictor_option_and_comm_probe_without_communicator(oracle_built):
    """clrs_mw_options.refine_predictor = 1: both solves of an iteration take the refinement step (the default refines the corrector's only): same
    iteration count, objectives equal far inside the tolerances of the solve.  clrs_mw_comm_probe on a context without a communicator reports that."""
    from clrs_amd.mw import MwSchurContext, solvesdp_mw
    f = flat("ce_8_15")
    rs = []
    for rp in (False, True):
        ctx = MwSchurContext(f, limbs=5, refine_predictor=rp)
        if not rp:
            pr = ctx.comm_probe(2)
            assert pr["backend"] == "none" and pr["world"] == 1 and pr["q_us"] == 0.0
        rs.append(solvesdp_mw(f, ctx=ctx))
        ctx.close()
    a, b = rs
    assert a.status == b.status == "Optimal" and a.iterations == b.iterations == 56
    assert abs(a.primal_objective - b.primal_objective) <= 1e-20 and abs(a.duality_gap - b.duality_gap) <= 1e-6 * a.duality_gap


def test_stream_words_make_progress_when_streams_share_hardware_queues(oracle_built):
    """The two streams of the interior-point iteration synchronise through words that kernels store and await (clrs_mw_ipm_host.inc) instead of events
    where a launch exists to do it.  HIP may map several streams onto one hardware queue; the rule that keeps that safe -- a waiting launch is submitted
    after the launch that stores its word -- is exercised here with enough live contexts (two streams each) that streams must share queues: every solve
    must finish at its usual speed (a wait whose producer sat behind it in the same queue would take seconds per iteration: bench.py met exactly that
    before the rule)."""
    import time
    from clrs_amd.mw import MwSchurContext, solvesdp_mw
    f = flat("ce_8_3")
    ctxs = [MwSchurContext(f, limbs=4) for _ in range(10)]
    try:
        for c in ctxs:                       # every context creates its side stream (clrs_mw_ipm_create) and runs the loop
            r = solvesdp_mw(f, ctx=c, limbs=4, duality_gap_threshold=1e-10, dual_error_threshold=1e-20, primal_error_threshold=1e-20)
            assert r.error_code == 0
        t0 = time.time()
        its = 0
        first = None
        for c in ctxs:
            r = solvesdp_mw(f, ctx=c, limbs=4, duality_gap_threshold=1e-10, dual_error_threshold=1e-20, primal_error_threshold=1e-20)
            # the functional signal: a wait that ran out of polls reports a failed factorisation (error code 1), never numbers -- and every context must
            # produce the SAME solve, whatever queue its streams landed on
            assert r.error_code == 0 and r.status == "Optimal"
            first = first or r
            assert r.iterations == first.iterations and np.array_equal(r.history, first.history) and np.array_equal(r.y, first.y)
            its += r.iterations
        # (a loose clock on top: a wait whose producer sat behind it in the same queue costs its whole bound, seconds per iteration; an ordinary solve on a
        # shared host stays orders of magnitude below this)
        assert (time.time() - t0) / its < 0.25, ("seconds per iteration", (time.time() - t0) / its)
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("K", [4, 5, 6])
def test_inverse_factor_columns_of_a_large_block_are_shared(K, oracle_built):
    """k_mw_potrf_x: a PSD block of more than 32 rows whose inverse factor still fits in LDS beside it (48 x 48 in Nsphere_packing(8,15,[1/2,1/2,1/2]) at
    4-6 limbs) is eliminated by four workgroups that share out the columns of the inverse: factor against the oracle's, and the assembly that reads the
    inverse factors (Z = Xi V) against the oracle's from the same factors -- every column of every inverse must have been written, by its own workgroup."""
    from clrs_amd.mw import MwSchurContext
    from oracle.oracle import Oracle
    f = flat("ns_8_15_3")
    assert int(max(f.block_n)) == 48
    X, Y = _iterates(f, K)
    X, Y = _sym_limbs(f, X), _sym_limbs(f, Y)
    o = Oracle(f, mp_bits=320 if K <= 5 else 640)
    ctx = MwSchurContext(f, limbs=K)
    Xc = ctx.cholesky_blocks(X)
    st, Xc_ref = o.cholesky_blocks_mw(np.vstack([X, np.zeros((1, f.xy_len))]))
    assert st == 0
    assert mw_relerr(Xc, Xc_ref) <= tol(K, 14), ("chol X", mw_relerr(Xc, Xc_ref))
    S, AY = ctx.compute_S_integrated(Xc, Y)
    S_ref, AY_ref = o.schur_assemble_mw(np.vstack([Xc, np.zeros((1, f.xy_len))]), np.vstack([Y, np.zeros((1, f.xy_len))]))
    assert mw_relerr(S, S_ref) <= tol(K, 22), ("S", mw_relerr(S, S_ref))
    ctx.close()


@pytest.mark.parametrize("K", [3, 5, 6, 10])
@pytest.mark.parametrize("name", ["ce_8_15", "ce_8_3", "polyopt8", "delsarte_3_10", "polyopt40", "threepoint_4", "sdpa_small", "ns_8_15_2", "polyopt_scaled_100"])
def test_pipelined_factorisation_is_bit_identical(name, K, oracle_built):
    """csrc/clrs_mw_pipe.hip.h: chol(S_j), chol(Q) and their inverse factors as pipelines of workgroups (column blocks of eight as stages, four more
    workgroups for the inverse, pivot columns handed on as tagged granules) do the arithmetic of the one-workgroup elimination entry by entry and
    pivot by pivot: factors, reciprocal diagonals (through LinvB and the solves) and solutions agree BIT FOR BIT with `pipeline=False`, on matrix sides
    1 ... 32 including sides that are no multiple of the stage width (31, 22, 9), with and without free variables; clusters of 33 ... 64 rows (polyopt40:
    41, threepoint_4: 50) through the 64-row form of the pipeline (k_mw_factor_pipe64, up to six limbs).  Matrices beyond LDS (the last two
    instances: P = 96 with N = 97 free variables beside clusters that ride on the first launch; P = 201): the diagonal blocks of the blocked factorisation
    go through the same pipeline (k_mw_bp_diag_pipe), last blocks of 1 and 9 columns included."""
    from clrs_amd.mw import MwSchurContext
    f = flat(name)
    if K == 3 and name == "ce_8_15":
        pytest.skip("S_j of cohnelkies(8,15) is not positive definite at 3 limbs on these iterates: the failure path is the next test's")
    X, Y = _iterates(f, K)
    X, Y = _sym_limbs(f, X), _sym_limbs(f, Y)
    rng = np.random.default_rng(11)
    rx, ry = mw_with_tails(rng.standard_normal(f.x_len), K, 1), mw_with_tails(rng.standard_normal(max(f.n_free, 1)), K, 2)[:, :f.n_free]
    out = []
    for pipe in (False, True):
        ctx = MwSchurContext(f, limbs=K, pipeline=pipe)
        for rep in range(2):                                  # twice: the second launch meets the granules of the first (older epoch)
            Xc = ctx.cholesky_blocks(X)
            ctx.compute_S_integrated(Xc, Y)
            assert ctx.factor() == 0
            fac = ctx.get_factor()
            sol = ctx.solve(rx, ry)
        out.append((fac, sol))
        ctx.close()
    for a, b in zip(out[0][0] + out[0][1], out[1][0] + out[1][1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("K", [5, 6])
@pytest.mark.parametrize("name", ["ce_8_15", "polyopt40", "threepoint_4", "ns_8_15_2"])
def test_reduced_factor_limbs_in_the_stand_alone_entry_points(name, K, oracle_built):
    """clrs_mw_options.factor_limbs = limbs - 1 (mixed-precision refinement, clrs_mw_kernels.hip.h::mw_kf_of): the factor stage and the inverse-factor products
    of the solve stage in one limb less, the residuals of the refinement step and the solution in all limbs.  Every factorisation form (one workgroup,
    pipeline of 32 / 64 rows, blocked) computes the same reduced factors bit for bit; their upper limb plane is zero; the refined solution has the KKT
    backward error of the working precision to within 20 bits of 53 K on these synthetic iterates (12 with full-limb factors: inside the interior-point loop the
    measured first-pass accuracy decides when the reduced form is left)."""
    from clrs_amd.mw import MwSchurContext
    from oracle.oracle import Oracle
    f = flat(name)
    X, Y = _iterates(f, K)
    X, Y = _sym_limbs(f, X), _sym_limbs(f, Y)
    rng = np.random.default_rng(5)
    rx, ry = mw_with_tails(rng.standard_normal(f.x_len), K, 1), mw_with_tails(rng.standard_normal(max(f.n_free, 1)), K, 2)[:, :f.n_free]
    o = Oracle(f, mp_bits=320 if K <= 5 else 640)
    pad = lambda a: np.vstack([a, np.zeros((1, a.shape[1]))])
    out = []
    for pipe in (False, True):
        ctx = MwSchurContext(f, limbs=K, pipeline=pipe, factor_limbs=K - 1)
        Xc = ctx.cholesky_blocks(X)
        ctx.compute_S_integrated(Xc, Y)
        S_ref, _ = o.schur_assemble_mw(pad(Xc), pad(Y))
        assert ctx.factor() == 0
        fac = ctx.get_factor()
        dx, dy = ctx.solve(rx, ry)
        ctx.close()
        out.append((fac, (dx, dy)))
        for a in fac:
            if a.size:
                assert np.all(a[K - 1] == 0.0) and np.any(a[K - 2] != 0.0)
        bx, by = o.kkt_backward_error_mw(S_ref, dx, dy, rx, ry if f.n_free else np.zeros((K, 0)))
        slack = 60 if name == "ns_8_15_2" else 20      # (Nsphere_packing: the products alone lose 82 bits on this iterate -- the loop would leave the reduced form here)
        assert 53 * K + np.log2(bx) <= slack and (f.n_free == 0 or 53 * K + np.log2(by) <= slack), (53 * K + np.log2(bx), 53 * K + np.log2(by) if f.n_free else None)
    for a, b in zip(out[0][0] + out[0][1], out[1][0] + out[1][1]):
        assert np.array_equal(a, b)
    with pytest.raises(Exception):
        MwSchurContext(f, limbs=K, factor_limbs=K - 2)


def test_pipelined_factorisation_reports_a_nonpositive_pivot_and_does_not_hang(oracle_built):
    """every workgroup of a matrix stops at the pivot its producer found non-positive (approx_cholesky!'s test, src/tools.jl:92-95): the status is the
    one-workgroup kernel's, and it comes at once: the stage that owns the failing column publishes it before it stops (without that its consumers polled
    for the column until their bound, 1.3 s -- found on the 16-cluster weak-scaling instance of bench.py, whose solve ends with a failed factorisation)."""
    import time
    # S_j not positive definite: at 2 limbs by rounding (a pivot in the middle of a cluster); Y negated (the first pivot of every cluster); the same on the
    # instance whose large cluster and Q go through the blocked path (k_mw_bp_diag_pipe: a matrix that failed is eliminated again in every later block
    # column, and fails again with the same code)
    for name, K, flip in (("ce_8_15", 2, False), ("ce_8_15", 5, True), ("ns_8_15_2", 2, False), ("ns_8_15_2", 5, True)):
        f = flat(name)
        X, Y = _iterates(f, K)
        X, Y = _sym_limbs(f, X), _sym_limbs(f, Y)
        if flip:
            Y = -Y
        _pivot_failure_case(f, K, X, Y, time)


def test_a_solve_that_ends_with_a_failed_factorisation_does_not_stall():
    """cohnelkies_multi(8, 15) with 15 radius scalings in steps of 1/8 (16 clusters: the first form of bench.py's eight-rank instance) ends at 5 limbs with a
    factorisation that fails at mu = 6e-15 (the reference's SolverFailure inside the loop, src/solver.jl:1249: status by the gap reached, code 1).  Through
    the pipelined factorisation that last iteration took 1.3 s -- 18.7 ms per iteration over the solve -- before the owner of a failing column published it."""
    import clrs_amd
    from clrs_amd.mw import solvesdp_mw
    from clrs_amd.problems import cohnelkies_multi
    full = clrs_amd.flatten(cohnelkies_multi(8, 15, [1.0 + 0.125 * k for k in range(15)]))
    r = solvesdp_mw(full, limbs=5, dual_error_threshold=1e-30, primal_error_threshold=1e-30, duality_gap_threshold=1e-15)
    assert r.error_code == 1 and r.status == "NearOptimal" and 70 <= r.iterations <= 80, (r.status, r.error_code, r.iterations)
    assert r.time_total / r.iterations < 3e-3, r.time_total / r.iterations


def _pivot_failure_case(f, K, X, Y, time):
    from clrs_amd.mw import MwSchurContext
    st = []
    for pipe in (False, True):
        ctx = MwSchurContext(f, limbs=K, pipeline=pipe)
        for rep in range(2):                                  # (the second factorisation is the timed one)
            Xc = ctx.cholesky_blocks(X)
            ctx.compute_S_integrated(Xc, Y)
            t0 = time.perf_counter()
            code = ctx.factor()
            dt = time.perf_counter() - t0
        st.append(code)
        # the owner of the failing column publishes it before it stops: nobody polls for it until the bound (1.3 s before that was so)
        assert dt < 0.2, (pipe, dt)
        ctx.close()
    assert st[0] == st[1] and st[0] > 0, st


@pytest.mark.parametrize("K", [4, 5])
def test_mw_factors_the_north_star_instance_where_fp64_fails(K, oracle_built):
    """cohnelkies(8,15) at the first iterate X = Y = Omega I (src/solver.jl:187-201): the fp64 path reports the reference's
    SolverFailure (tests/test_hip_parity.py), 113 bits fail too; 4 and 5 limbs factor S_j and solve to the oracle's answer."""
    from clrs_amd.mw import MwSchurContext
    from clrs_amd.solver import SchurContext
    from oracle.oracle import Oracle
    f = flat("ce_8_15")
    X = np.zeros(f.xy_len); Y = np.zeros(f.xy_len)
    for b in range(f.n_blocks):
        n = int(f.block_n[b]); o0 = int(f.block_off[b])
        X[o0:o0 + n * n] = (1e10 * np.eye(n)).reshape(-1); Y[o0:o0 + n * n] = (1e10 * np.eye(n)).reshape(-1)
    c64 = SchurContext(f)
    c64.compute_S_integrated(c64.cholesky_blocks(X), Y)
    assert c64.factor() > 0
    c64.close()
    ctx = MwSchurContext(f, limbs=K)
    Xm, Ym = mw_from_double(X, K), mw_from_double(Y, K)
    Xc = ctx.cholesky_blocks(Xm)
    S, _ = ctx.compute_S_integrated(Xc, Ym)
    assert ctx.factor() == 0
    o = Oracle(f, mp_bits=320)
    S_ref, _ = o.schur_assemble_mw(np.vstack([Xc, np.zeros((1, f.xy_len))]), np.vstack([Ym, np.zeros((1, f.xy_len))]))
    assert mw_relerr(S, S_ref) <= tol(K, 22)
    assert o.schur_factor() == 0
    rx, ry = mw_from_double(np.ones(f.x_len), K), mw_from_double(np.ones(f.n_free), K)
    dx, dy = ctx.solve(rx, ry)
    dx_ref, dy_ref = o.schur_solve_mw(np.vstack([rx, np.zeros((1, f.x_len))]), np.vstack([ry, np.zeros((1, f.n_free))]))
    assert_backward_stable(o, S_ref, dx, dy, rx, ry, K, "ce_8_15")
    # cond(S) ~ 1e37 here (lambda_min/lambda_max of the sampled form): what is left of 53 K bits
    assert tol(K, 150) < 1e-6
    assert mw_relerr(dx, dx_ref) <= tol(K, 150), mw_relerr(dx, dx_ref)
    assert mw_relerr(dy, dy_ref) <= tol(K, 150), mw_relerr(dy, dy_ref)
    ctx.close()


def test_mw_reports_failures_like_the_reference(oracle_built):
    """A non-positive pivot in S_j returns j+1, in a block of X b+1 (src/solver.jl:395-397, 1249)."""
    from clrs_amd.mw import MwSchurContext
    from clrs_amd.solver import SolverFailure
    f = flat("polyopt8")
    K = 3
    ctx = MwSchurContext(f, limbs=K)
    X, Y = spd_iterates(f, seed=2)
    Xm, Ym = mw_from_double(X, K), mw_from_double(-Y, K)          # Y negative definite: S is too
    Xc = ctx.cholesky_blocks(Xm)
    ctx.compute_S_integrated(Xc, Ym)
    assert ctx.factor() == 1
    with pytest.raises(SolverFailure, match="block \\(1,1\\)"):
        ctx.cholesky_blocks(mw_from_double(-X, K))
    ctx.close()


# ---- the whole interior-point loop at the reference's precision ------------------------------------------------------
PI4_384 = 0.25366950790104804          # pi^4 / 384, test/runtests_solver.jl:20,22


def test_north_star_instance_solves_to_the_pinned_objective(oracle_built):
    """cohnelkies(8,15) with the reference's DEFAULT options (Omega = 1e10, gap 1e-15, errors 1e-30; prec = 256 -> 5 limbs):
    test/runtests_solver.jl:19-20 pins the objective to pi^4/384 within 1e-4.  The oracle at 256 bits takes 56 iterations."""
    from clrs_amd.mw import solvesdp_mw
    from oracle.oracle import Oracle
    f = flat("ce_8_15")
    r = solvesdp_mw(f, prec=256)
    assert r.error_code == 0 and r.status == "Optimal", (r.status, r.error_code, r.iterations)
    assert abs(r.primal_objective - PI4_384) <= 1e-4 and abs(r.dual_objective - PI4_384) <= 1e-4
    ro = Oracle(f, mp_bits=256).solvesdp()
    assert ro["error_code"] == 0
    assert abs(r.primal_objective - ro["p_obj"]) <= 1e-12 and abs(r.dual_objective - ro["d_obj"]) <= 1e-12
    assert abs(r.iterations - ro["iterations"]) <= 2, (r.iterations, ro["iterations"])
    # the early iterations follow the oracle's trace to the digits the step-length eigenvalue (fp64, -1e-5) leaves
    h, ho = r.history, ro["hist"]
    for it in range(5):
        assert abs(h[it, 1] - ho[it, 1]) <= 1e-6 * ho[it, 1]              # mu
        assert abs(h[it, 8] - ho[it, 8]) <= 1e-6 and abs(h[it, 9] - ho[it, 9]) <= 1e-6   # alpha_d, alpha_p


def test_four_limbs_reach_the_objective_with_thresholds_for_209_bits(oracle_built):
    """4 limbs (~209 bits): the oracle's sweep reaches gap 1e-14 and errors 5e-29 / 5e-38 at 212 bits, so the
    error thresholds are set to 1e-25 instead of the 256-bit defaults 1e-30."""
    from clrs_amd.mw import solvesdp_mw
    r = solvesdp_mw(flat("ce_8_15"), limbs=4, dual_error_threshold=1e-25, primal_error_threshold=1e-25, duality_gap_threshold=1e-12)
    assert r.error_code == 0 and r.status == "Optimal", (r.status, r.error_code, r.iterations)
    assert abs(r.primal_objective - PI4_384) <= 1e-4


def test_fp64_rounded_problem_data_is_a_different_problem(oracle_built):
    """Why the ABI takes multi-word problem DATA too (clrs_mw_create_ex): with B, c, the sampled vectors rounded to fp64 the
    cohnelkies(8,15) SDP is no longer the same problem -- p = b - B^T x stalls near 1e9 and mu blows up (error code 3, "mu too
    large") at 5 limbs, in the oracle at 256 bits exactly as on the GPU."""
    from clrs_amd.mw import solvesdp_mw
    from oracle.oracle import Oracle
    f = flat("ce_8_15")
    r = solvesdp_mw(f, limbs=5, data_limbs=1)
    ro = Oracle(f, mp_bits=256, use_lo=False).solvesdp()
    assert r.error_code == 3 and ro["error_code"] == 3
    assert abs(r.iterations - ro["iterations"]) <= 1


def test_nsphere_packing_prec_300_instance(oracle_built):
    """Nsphere_packing(8,15,[1/2,1/2],2) (test/runtests_solver.jl:21-22) at the reference's prec = 300 -> 6 limbs (315 bits):
    7 clusters, m = 2 sub-blocks, P = 96 (blocked path); same pinned value, and the same objective as the 5-limb solve."""
    from clrs_amd.mw import solvesdp_mw, limbs_for_precision
    assert limbs_for_precision(300) == 6 and limbs_for_precision(256) == 5 and limbs_for_precision(400) == 8 and limbs_for_precision(512) == 10
    r = solvesdp_mw(flat("ns_8_15_2"), prec=300)
    assert r.error_code == 0 and r.timings["limbs"] == 6, (r.status, r.error_code, r.iterations)
    assert abs(r.primal_objective - PI4_384) <= 1e-4, r.primal_objective
    r5 = solvesdp_mw(flat("ns_8_15_2"), limbs=5)
    assert r5.error_code == 0 and abs(r5.primal_objective - r.primal_objective) <= 1e-12


def test_prec_512_and_gap_1e60_as_in_the_reference_tutorial():
    """docs/src/tutorial.md:169: solvesdp(problem; prec=512, duality_gap_threshold=1e-60) -- prec = 512 maps to 10 limbs (525 bits).  On
    the north-star instance (whose generator defaults to prec = 512 in examples/SpherePacking.jl:117; at 10 limbs its 32 x 32
    clusters take the blocked path with 16-wide panels) the loop reaches the 1e-60 gap and the pinned objective.  The problem DATA
    stay double-double (106 bits), so digits beyond ~1e-30 describe that problem, not the exact one."""
    from clrs_amd.mw import solvesdp_mw
    r = solvesdp_mw(flat("ce_8_15"), prec=512, duality_gap_threshold=1e-60, primal_error_threshold=1e-60, dual_error_threshold=1e-60)
    assert r.timings["limbs"] == 10
    assert r.error_code == 0 and r.status == "Optimal" and r.duality_gap <= 1e-60, (r.status, r.error_code, r.duality_gap, r.iterations)
    assert abs(r.primal_objective - PI4_384) <= 1e-4
    r5 = solvesdp_mw(flat("ce_8_15"), limbs=5)
    assert abs(r.primal_objective - r5.primal_objective) <= 1e-13


def test_problem_data_at_the_working_precision(oracle_built):
    """The reference holds the sampled problem at `prec` bits (convert_to_prec, src/interface.jl:1078-1112); `data_limbs = limbs` does the same here
    (round-4 review: with 2-limb data a prec = 512, gap 1e-60 solve describes a neighbouring problem).  cohnelkies(8,15) generated with ten limb planes
    (`sdp.data_planes`): the 10-limb solve on 10-limb data agrees with the 640-bit oracle on the same data to 1e-55; the solve on the first two planes only
    is the solution of another problem -- how far away is measured here (its entries reach 1e27: a relative 2^-106 is 1e-5 absolute)."""
    import mpmath as mp
    from clrs_amd.mw import solvesdp_mw
    from clrs_amd.problems import cohnelkies
    from clrs_amd.sdp import data_planes, flatten
    from oracle.oracle import Oracle
    with data_planes(10):
        f = flatten(cohnelkies(8, 15))
    assert set(f.tails) >= {"B", "c", "b", "C", "term_lambda", "term_vs", "term_ws", "dense_A"} and f.tails["B"].shape == (8, f.B.size)
    assert np.max(np.abs(f.tails["B"][0])) > 0 and np.max(np.abs(f.tails["B"][0])) < 1e-30 * np.max(np.abs(f.B))
    kw = dict(duality_gap_threshold=1e-60, primal_error_threshold=1e-60, dual_error_threshold=1e-60)
    r10 = solvesdp_mw(f, prec=512, data_limbs=10, **kw)
    assert r10.timings["limbs"] == 10 and r10.error_code == 0 and r10.status == "Optimal" and r10.duality_gap <= 1e-60
    o = Oracle(f, mp_bits=640)
    o.set_num_threads(8)
    ro = o.solvesdp(**kw)
    assert ro["error_code"] == 0 and ro["gap"] <= 1e-60

    def val(limbs):
        return mp.fsum(mp.mpf(float(v)) for v in limbs)
    with mp.workprec(800):
        po, do = val(ro["objectives_limbs"][1]), val(ro["objectives_limbs"][0])
        pg, dg = val(r10.timings["objectives_limbs"][1]), val(r10.timings["objectives_limbs"][0])
        assert abs(pg - po) <= mp.mpf(10) ** -55 and abs(dg - do) <= mp.mpf(10) ** -55, (mp.nstr(pg - po, 5), mp.nstr(dg - do, 5))
        r2 = solvesdp_mw(f, prec=512, data_limbs=2, **kw)
        assert r2.error_code == 0 and r2.status == "Optimal"
        far = abs(val(r2.timings["objectives_limbs"][1]) - po)
        assert mp.mpf(10) ** -20 < far < mp.mpf(10) ** -6, mp.nstr(far, 5)      # measured: 5.5e-11 -- ten digits, not thirty
    # the same at the reference's default precision: 5 limbs of data against 2
    r5 = solvesdp_mw(f, limbs=5, data_limbs=5)
    assert r5.error_code == 0 and r5.status == "Optimal" and abs(r5.primal_objective - float(po)) <= 1e-13, (r5.primal_objective, float(po))


def test_duality_gap_1e30_as_in_the_reference_rounding_test(oracle_built):
    """test/runtests_solver.jl:90: three_point_spherical_codes(4, 1//6, -1, 4, prec=256, duality_gap_threshold=1e-30, omega=10^3)
    -- the solve the reference rounds to the exact optimum 10.  5 limbs reach the 1e-30 gap; 8 limbs (420 bits) reach 1e-45 and
    agree with the 5-limb solution to the accuracy of the latter."""
    from clrs_amd.mw import solvesdp_mw
    kw = dict(omega_p=1e3, omega_d=1e3)
    r = solvesdp_mw(flat("threepoint_4"), prec=256, duality_gap_threshold=1e-30, **kw)
    assert r.error_code == 0 and r.status == "Optimal" and r.duality_gap <= 1e-30, (r.status, r.duality_gap)
    assert abs(r.primal_objective - 10.0) <= 1e-12
    r8 = solvesdp_mw(flat("threepoint_4"), limbs=8, duality_gap_threshold=1e-45, primal_error_threshold=1e-50, dual_error_threshold=1e-50, **kw)
    assert r8.error_code == 0 and r8.status == "Optimal" and r8.duality_gap <= 1e-45, (r8.status, r8.duality_gap)
    assert abs(r8.primal_objective - 10.0) <= 1e-12


@pytest.mark.parametrize("name,expected,tol_,kw", [
    ("x2p1", 1.0, 1e-10, {}),
    ("delsarte_3_10", 13.158314, 1e-5, {}),
    ("delsarte_8_3", 240.0, 1e-8, {}),
    ("threepoint_4", 10.0, 1e-5, dict(omega_p=1e3, omega_d=1e3)),
    ("polyopt40", None, 1e-10, {}),
    ("sdpa_example", None, 1e-10, {}),
])
def test_mw_loop_reaches_the_pinned_objectives(name, expected, tol_, kw, oracle_built):
    """The reference's own known answers (test/runtests_solver.jl:15, 86-87, 26-27; README.md:149) with its default options,
    and agreement with the 256-bit oracle loop on the rest."""
    from clrs_amd.mw import solvesdp_mw
    from oracle.oracle import Oracle
    f = flat(name)
    r = solvesdp_mw(f, limbs=5, **kw)
    assert r.error_code == 0 and r.status == "Optimal", (name, r.status, r.error_code)
    if expected is not None:
        assert abs(r.primal_objective - expected) <= tol_, (name, r.primal_objective)
    ro = Oracle(f, mp_bits=256).solvesdp(**kw)
    assert ro["error_code"] == 0
    assert abs(r.primal_objective - ro["p_obj"]) <= 1e-10 * max(1.0, abs(ro["p_obj"]))
    assert abs(r.iterations - ro["iterations"]) <= 2


def _golden_config(name):
    """tests/golden/configs_256.npz (written by tests/golden/make_golden_configs.py from the 256-bit CPU oracle with the reference's default options):
    dict(iterations, error_code, p_obj, d_obj, gap, dual_error, primal_error, hist)"""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "configs_256.npz"))
    v = z[name + "/summary"]
    return dict(iterations=int(v[0]), error_code=int(v[1]), p_obj=float(v[2]), d_obj=float(v[3]), gap=float(v[4]), dual_error=float(v[5]),
                primal_error=float(v[6]), hist=z[name + "/hist"])


@pytest.mark.parametrize("name,kw", [
    ("threepoint_3_8_8", dict(omega_p=1e3, omega_d=1e3)),
    ("sdpa_x64", {}),
    ("ns_8_15_2", {}),
    ("ns_8_15_3", {}),
])
def test_baseline_configs_as_named_solve_like_the_256_bit_oracle(name, kw):
    """BASELINE configs 4 ("ThreePointBound n=3, 2d=16": P = 221, PSD blocks to 54x54 -- inverse factors formed in place in memory,
    cluster beyond LDS -- blocked path), 5 ("SDPA x64 blocks": 64 dense 32x32 blocks, P = 256) and 3 ("many small clusters": Nsphere_packing(8,15) with
    two radii -- 7 clusters -- and three -- 11 clusters, P = 192, 193 free variables) through the whole device-resident loop at 5 limbs = the reference's
    default prec = 256 with its default thresholds, against the committed runs of the 256-bit oracle (143 / 279 / 15 / 43 s on 8 threads: too long
    for the suite): Optimal, the oracle's iteration count, objectives, and its mu / step-length columns."""
    from clrs_amd.mw import solvesdp_mw
    g = _golden_config(name)
    assert g["error_code"] == 0
    r = solvesdp_mw(flat(name), limbs=5, **kw)
    assert r.error_code == 0 and r.status == "Optimal", (name, r.status, r.error_code, r.iterations, r.duality_gap, r.dual_error, r.primal_error)
    assert r.iterations == g["iterations"], (name, r.iterations, g["iterations"])
    assert abs(r.primal_objective - g["p_obj"]) <= 1e-12 * max(1.0, abs(g["p_obj"])) and abs(r.dual_objective - g["d_obj"]) <= 1e-12 * max(1.0, abs(g["d_obj"]))
    assert abs(r.duality_gap - g["gap"]) <= 1e-3 * g["gap"]
    n = r.iterations
    for col in (1, 8, 9, 10):                       # mu, both step lengths, beta_c of every iteration
        assert np.allclose(r.history[:n, col], g["hist"][:n, col], rtol=1e-5, atol=1e-300), (name, col)
    if name.startswith("ns_"):
        assert abs(r.primal_objective - PI4_384) <= 1e-4                                   # test/runtests_solver.jl:21-22


def test_mw_loop_beta_follows_the_reference_order_across_the_feasibility_flip(oracle_built):
    """beta_c of the iteration in which the iterate becomes feasible is still chosen with the PREVIOUS feasibility
    (src/solver.jl:429-434 before :441-447): the per-iteration beta_c column agrees with the oracle's."""
    from clrs_amd.mw import solvesdp_mw
    from oracle.oracle import Oracle
    f = flat("delsarte_3_10")
    r = solvesdp_mw(f, limbs=5)
    ro = Oracle(f, mp_bits=256).solvesdp()
    n = min(len(r.history), len(ro["hist"]))
    assert n > 10
    flips = 0
    for it in range(n):
        assert abs(r.history[it, 10] - ro["hist"][it, 10]) <= 1e-6 * max(1.0, ro["hist"][it, 10]), (it, r.history[it, 10], ro["hist"][it, 10])
        if it and (r.history[it, 10] == 0.1) != (r.history[it - 1, 10] == 0.1):
            flips += 1
    assert flips >= 1


# ---- kernel-level parity on real interior-point iterates ---------------------------------------------
COND_X_BITS = {1: 0, 2: 2, 28: 40, 55: 56}          # bits of S lost to cond(X) at iterations 1, 2, 28, 55 of the fixture
COND_S_BITS = {1: 122, 2: 126, 28: 120, 55: 165}    # bits of (dx, dy) lost to cond(S)


@pytest.mark.parametrize("K", [4, 5])
def test_mw_path_on_the_trajectory_fixture(K, oracle_built):
    """tests/golden/ce_8_15_traj.npz: (X, Y, rhs) at iterations 1, 2, K/2, K-1 of the mpmath restatement of the whole loop on
    cohnelkies(8,15) (mu from 1e20 down to 2e-16), with S from the dense trace formula and (dx, dy) from an LU solve of the KKT matrix
    at 456 bits -- an answer that shares neither arithmetic nor algorithm with the HIP path.
    Tolerances: S carries cond(X) of the iterate, (dx, dy) carry cond(S) (up to ~1e37 on this problem, which is why it needs
    the precision it needs): 2^-(53 K - slack) with the slack stated per quantity."""
    import os
    from clrs_amd.mw import MwSchurContext
    f = flat("ce_8_15")
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ce_8_15_traj.npz"))
    ctx = MwSchurContext(f, limbs=K)
    from oracle.oracle import Oracle
    o = Oracle(f, mp_bits=320)
    worst, bwd = {}, {}
    for s, it in enumerate(g["iters"]):
        X, Y = np.ascontiguousarray(g["X"][s][:K]), np.ascontiguousarray(g["Y"][s][:K])
        Xc = ctx.cholesky_blocks(X)
        S, _ = ctx.compute_S_integrated(Xc, Y)
        eS = mw_relerr(S, g["S"][s])
        if K == 4 and it == g["iters"][-1]:
            # the iterate of iteration K-1 (gap 8e-15, mu 2e-16) belongs to the 256-bit run: at ~209 bits S_j is no longer numerically
            # positive definite there -- the 212-bit oracle run stops at gap 1e-14 for the same reason
            assert eS <= 2.0 ** -(53 * K - 6 - COND_X_BITS[int(it)]), (it, eS)
            assert ctx.factor() > 0
            continue
        assert ctx.factor() == 0, it
        dx, dy = ctx.solve(np.ascontiguousarray(g["rhs_x"][s][:K]), np.ascontiguousarray(g["rhs_y"][s][:K]))
        # backward error against the FIXTURE's S (dense trace formula at 456 bits, from the 6-limb iterate: the K-limb truncation of X enters with cond(X))
        bx, by = o.kkt_backward_error_mw(g["S"][s], dx, dy, g["rhs_x"][s][:K], g["rhs_y"][s][:K])
        bwd[int(it)] = (np.log2(max(bx, 1e-300)), np.log2(max(by, 1e-300)))
        assert bx <= tol(K, 16 + COND_X_BITS[int(it)]) and by <= tol(K, 16 + COND_X_BITS[int(it)]), (it, bwd)
        edx, edy = mw_relerr(dx, g["dx"][s]), mw_relerr(dy, g["dy"][s])
        worst[int(it)] = (np.log2(eS), np.log2(max(edx, 1e-300)), np.log2(max(edy, 1e-300)))
        # measured (scripts/traj_errors.py, K = 3, 4, 5): every limb buys 52-54 bits on all three quantities; what is lost is the conditioning
        # of the iterate, the same number of bits at every K: log2 cond(X) for S, log2 cond(S) (~1e35 from the first iterate on) for dx, dy
        assert eS <= 2.0 ** -(53 * K - 6 - COND_X_BITS[int(it)]), (it, worst)
        assert max(edx, edy) <= 2.0 ** -(53 * K - 6 - COND_S_BITS[int(it)]), (it, worst)
    print("log2 relative errors (S, dx, dy) per iteration:", worst, "log2 backward errors (x rows, y rows):", bwd)
    ctx.close()


# ---- cluster sharding of the multi-word path ----------------------------------------------------------
def _take(flat_full, shard_ids, M, kind):
    """rows of a planar full-problem array that belong to the clusters `shard_ids`: kind 'xy' (blocks) or 'x' (constraints)"""
    f = flat_full
    if kind == "x":
        idx = np.concatenate([np.arange(int(f.cluster_off[j]), int(f.cluster_off[j + 1])) for j in shard_ids])
    else:
        idx = np.concatenate([np.arange(int(f.block_off[b]), int(f.block_off[b + 1])) for b in range(f.n_blocks) if int(f.block_cluster[b]) in shard_ids])
    return np.ascontiguousarray(M[:, idx])


def test_mw_two_shards_on_one_gpu_match_the_unsharded_path(oracle_built):
    """Clusters split over two contexts ("ranks") on one GPU, the exchange of the partial Q and u done by copying the gather slots
    (what an all-gather does): every rank ends with the same dy bit for bit, and (dx, dy) agree with the unsharded context and with
    the 320-bit oracle.  cohnelkies_multi(8, 3, 3 radii): 4 clusters, N = 7."""
    import torch
    import clrs_amd
    from clrs_amd.mw import MwSchurContext
    from clrs_amd.problems import cohnelkies_multi
    from clrs_amd.sdp import shard_clusters
    from clrs_amd.sharded import _DevArray
    from oracle.oracle import Oracle
    K = 4
    full = clrs_amd.flatten(cohnelkies_multi(8, 3, [1.0, 1.125, 1.25]))
    assert full.n_clusters == 4
    X, Y = _iterates(full, K)
    X, Y = _sym_limbs(full, X), _sym_limbs(full, Y)
    rng = np.random.default_rng(3)
    rx, ry = mw_with_tails(rng.standard_normal(full.x_len), K, 5), mw_with_tails(rng.standard_normal(full.n_free), K, 6)
    ref = MwSchurContext(full, limbs=K)
    Xc = ref.cholesky_blocks(X)
    ref.compute_S_integrated(Xc, Y)
    assert ref.factor() == 0
    dx_ref, dy_ref = ref.solve(rx, ry)
    parts = [[0, 2], [1, 3]]
    dev = "cuda:0"
    N = full.n_free
    ranks = []
    for r, ids in enumerate(parts):
        sh = shard_clusters(full, ids)
        c = MwSchurContext(sh, limbs=K)
        c.set_shard(r, 2)
        t = dict(X=torch.tensor(_take(full, ids, X, "xy"), device=dev), Y=torch.tensor(_take(full, ids, Y, "xy"), device=dev),
                 rx=torch.tensor(_take(full, ids, rx, "x"), device=dev), ry=torch.tensor(ry, device=dev))
        t["Xc"], t["dx"], t["dy"] = torch.empty_like(t["X"]), torch.empty_like(t["rx"]), torch.empty_like(t["ry"])
        t["Qg"] = torch.as_tensor(_DevArray(c.q_gather(), 2 * K * N * N), device=dev).view(2, K * N * N)
        t["ug"] = torch.as_tensor(_DevArray(c.u_gather(), 2 * K * N), device=dev).view(2, K * N)
        ranks.append((c, t, ids))
    for c, t, _ in ranks:
        c.cholesky_blocks_dev(t["X"].data_ptr(), t["Xc"].data_ptr())
        c.assemble_dev(t["Xc"].data_ptr(), t["Y"].data_ptr())
        c.factor_local_dev()
    torch.cuda.synchronize()
    for c in ranks:
        torch.cuda.current_stream().wait_stream(torch.cuda.ExternalStream(c[0].stream()))
    ranks[0][1]["Qg"][1].copy_(ranks[1][1]["Qg"][1]); ranks[1][1]["Qg"][0].copy_(ranks[0][1]["Qg"][0])      # the all-gather
    torch.cuda.synchronize()
    for c, t, _ in ranks:
        c.factor_finish_dev()
        assert c.sync_status() == 0
        c.solve_fwd_dev(t["rx"].data_ptr())
    torch.cuda.synchronize()
    ranks[0][1]["ug"][1].copy_(ranks[1][1]["ug"][1]); ranks[1][1]["ug"][0].copy_(ranks[0][1]["ug"][0])
    torch.cuda.synchronize()
    for c, t, _ in ranks:
        c.solve_bwd_dev(t["ry"].data_ptr(), t["dx"].data_ptr(), t["dy"].data_ptr())
        c.sync_status()
    torch.cuda.synchronize()
    dy_plain = ranks[0][1]["dy"].cpu().numpy().copy()
    # the refinement step of the split-phase protocol: the backward call left every rank's partial u' in its slot; one more exchange, then the correction
    ranks[0][1]["ug"][1].copy_(ranks[1][1]["ug"][1]); ranks[1][1]["ug"][0].copy_(ranks[0][1]["ug"][0])
    torch.cuda.synchronize()
    for c, t, _ in ranks:
        c.solve_refine_dev(t["ry"].data_ptr(), t["dx"].data_ptr(), t["dy"].data_ptr())
        c.sync_status()
    torch.cuda.synchronize()
    dy0, dy1 = ranks[0][1]["dy"].cpu().numpy(), ranks[1][1]["dy"].cpu().numpy()
    assert not np.array_equal(dy0, dy_plain)                                  # (the correction did something)
    assert np.array_equal(dy0, dy1)                                           # replicated bit for bit
    assert mw_relerr(dy0, dy_ref) <= tol(K, 60), mw_relerr(dy0, dy_ref)
    for c, t, ids in ranks:
        assert mw_relerr(t["dx"].cpu().numpy(), _take(full, ids, dx_ref, "x"), scale=np.max(np.abs(dx_ref[0]))) <= tol(K, 60)
    o = Oracle(full, mp_bits=320)
    pad = lambda a: np.vstack([a, np.zeros((1, a.shape[1]))])
    o.schur_assemble_mw(pad(Xc), pad(Y))
    assert o.schur_factor() == 0
    _, dy_o = o.schur_solve_mw(pad(rx), pad(ry))
    assert mw_relerr(dy0, dy_o) <= tol(K, 60)
    for c, _, _ in ranks:
        c.close()
    ref.close()


def test_mw_rccl_exchange_inside_the_c_abi_one_rank(oracle_built):
    """clrs_comm_unique_id / clrs_mw_comm_init with a one-rank communicator: clrs_mw_schur_factor_dev and clrs_mw_schur_solve_dev run
    their RCCL all-gathers on the context stream (a Julia host reaches the sharded path through these two calls alone) and give the
    results of the plain path bit for bit."""
    import torch
    from clrs_amd.mw import MwSchurContext
    K = 4
    f = flat("ce_8_3")
    X, Y = _iterates(f, K)
    X, Y = _sym_limbs(f, X), _sym_limbs(f, Y)
    rx, ry = mw_from_double(np.ones(f.x_len), K), mw_from_double(np.ones(f.n_free), K)
    plain = MwSchurContext(f, limbs=K)
    Xc = plain.cholesky_blocks(X)
    plain.compute_S_integrated(Xc, Y)
    assert plain.factor() == 0
    dx0, dy0 = plain.solve(rx, ry)
    plain.close()
    c = MwSchurContext(f, limbs=K)
    c.comm_init(MwSchurContext.comm_unique_id(), 0, 1)
    Xc2 = c.cholesky_blocks(X)
    c.compute_S_integrated(Xc2, Y)
    assert c.factor() == 0                     # host entry point -> clrs_mw_schur_factor_dev -> ncclAllGather -> finish
    dx1, dy1 = c.solve(rx, ry)
    assert np.array_equal(dx0, dx1) and np.array_equal(dy0, dy1)
    c.comm_destroy()
    c.close()


def test_solvesdp_device_with_the_reference_prec_keyword():
    """`solvesdp_device(sdp, prec=256)`: the reference's `prec` keyword (src/solver.jl:73) selects the limb count; the north-star
    instance ends Optimal at the pinned objective (test/runtests_solver.jl:19-20)."""
    from clrs_amd.solver import solvesdp_device
    r = solvesdp_device(flat("ce_8_15"), prec=256)
    assert r.error_code == 0 and r.status == "Optimal" and r.timings["limbs"] == 5
    assert abs(r.primal_objective - PI4_384) <= 1e-4


def test_fp64_assembly_paths_on_the_trajectory_fixture():
    """The fp64 assembly kernels (every path) on the fp64 heads of the trajectory iterates against the fixture's S: fp64 accuracy times
    cond(X) of the iterate (1, 4, 2^38, 2^52 at iterations 1, 2, 28, 55) -- the late iterates are beyond fp64 by their conditioning
    alone, which is the reason for the multi-word path; the factorisation of that S fails in fp64 at every iterate."""
    import os
    from clrs_amd.solver import SchurContext, SolverFailure
    f = flat("ce_8_15")
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ce_8_15_traj.npz"))
    for kw in (dict(), dict(wave3=False, wave2=True), dict(wave2=False), dict(wave=False), dict(fused=False)):
        ctx = SchurContext(f, **kw)
        for s, it in enumerate(g["iters"]):
            X, Y = g["X"][s][0].copy(), g["Y"][s][0].copy()
            try:
                Xc = ctx.cholesky_blocks(X)
            except SolverFailure:
                assert COND_X_BITS[int(it)] >= 38, it        # X itself is not positive definite to fp64 accuracy on the late iterates
                continue
            S, _ = ctx.compute_S_integrated(Xc, Y)
            err = np.max(np.abs(S - g["S"][s][0])) / np.max(np.abs(g["S"][s][0]))
            assert err <= 2.0 ** -(53 - 10 - COND_X_BITS[int(it)]) or COND_X_BITS[int(it)] >= 43, (kw, it, err)
            assert ctx.factor() > 0
        ctx.close()


# ---- breadth: random structures, handmade shapes, size limits, malformed input -------------------------------------------------
def _check_mw_against_oracle(f, K, seed, amp=40, DL=2, bw_slack=(24, 24)):
    from clrs_amd.mw import MwSchurContext
    from oracle.oracle import Oracle
    X, Y = _iterates(f, K, seed=seed)
    X, Y = _sym_limbs(f, X), _sym_limbs(f, Y)
    o = Oracle(f, mp_bits=320, use_lo=(DL == 2))
    ctx = MwSchurContext(f, limbs=K, data_limbs=DL)
    pad = lambda a: np.vstack([a, np.zeros((1, a.shape[1]))])
    Xc = ctx.cholesky_blocks(X)
    S, AY = ctx.compute_S_integrated(Xc, Y)
    S_ref, AY_ref = o.schur_assemble_mw(pad(Xc), pad(Y))
    assert mw_relerr(S, S_ref) <= tol(K, 22), mw_relerr(S, S_ref)
    if f.n_terms:
        assert mw_relerr(AY, AY_ref, scale=max(1.0, np.max(np.abs(AY_ref[0])))) <= tol(K, 16)
    o.set_S_mw(pad(S))
    assert o.schur_factor() == 0 and ctx.factor() == 0
    rng = np.random.default_rng(seed)
    rx = mw_with_tails(rng.standard_normal(f.x_len), K, 1)
    ry = mw_with_tails(rng.standard_normal(max(f.n_free, 1)), K, 2)[:, :f.n_free]
    dx, dy = ctx.solve(rx, ry)
    assert_backward_stable(o, S_ref, dx, dy, rx, ry, K, slack=bw_slack)
    dx_ref, dy_ref = o.schur_solve_mw(pad(rx), pad(ry) if f.n_free else np.zeros((K + 1, 0)))
    assert_forward_close(dx, dx_ref, tol(K, 22 + 3 * amp), "dx")
    if f.n_free:
        assert_forward_close(dy, dy_ref, tol(K, 22 + 3 * amp), "dy")
    ctx.close()


@pytest.mark.parametrize("seed", range(6))
def test_mw_random_structures(seed, oracle_built):
    """Clusters of different sizes P <= 32 in one context, 1-3 low-rank blocks of different sides n <= 16 per cluster, 0-2 dense 1 x 1 blocks
    touching a subset of the constraints, placed anywhere among them; 0-3 free variables (tests/util.py::random_simple_sdp)."""
    import clrs_amd
    from tests.util import random_simple_sdp
    f = clrs_amd.flatten(random_simple_sdp(seed, J=3 + seed % 3, n_free=seed % 4, definite=True))
    _check_mw_against_oracle(f, K=3 + seed % 3, seed=seed + 100)


@pytest.mark.parametrize("kw", [dict(), dict(rank2=True), dict(m=2), dict(m=2, rank2=True)])
def test_mw_handmade_shapes(kw, oracle_built):
    """rank-2 terms, sub-blocks (m = 2), a dense block touching only some constraints, de-duplicated vectors (the shapes of
    tests/test_hip_parity.py::_mini_sdp)."""
    import clrs_amd
    from tests.test_hip_parity import _mini_sdp
    _check_mw_against_oracle(clrs_amd.flatten(_mini_sdp(**kw)), K=4, seed=3)


@pytest.mark.parametrize("P,N", [(64, 64), (65, 7), (33, 65), (70, 0)])
def test_mw_size_limits(P, N, oracle_built):
    """The limits where the solve stage switches between the single-wave register form (P, N <= 64) and the workgroup form, and where
    clusters leave LDS (4 limbs: P > ~68) for the blocked multi-workgroup factorisation: two clusters of exactly P constraints (four
    low-rank blocks of side 16), N free variables."""
    import clrs_amd
    from tests.util import random_simple_sdp
    f = clrs_amd.flatten(random_simple_sdp(2000 + P + N, J=2, n_free=N, fixed_P=P, max_n=16, lr_blocks=4))
    _check_mw_against_oracle(f, K=4, seed=P + 2 * N, amp=50)


@pytest.mark.parametrize("J", [90, 200])
def test_mw_many_clusters(J, oracle_built):
    """Many small clusters in one context: the workgroups that share the columns of an inverse factor are 4 per cluster only while
    4 J <= 256 compute units (J = 90: 2 per cluster, J = 200: 1), and every per-cluster kernel runs with a grid of J."""
    import clrs_amd
    from tests.util import random_simple_sdp
    f = clrs_amd.flatten(random_simple_sdp(4000 + J, J=J, n_free=3, fixed_P=6, max_n=4, lr_blocks=2))
    _check_mw_against_oracle(f, K=5, seed=J, amp=40)


def test_mw_malformed_descriptions_and_call_order():
    """The multi-word ABI validates like the fp64 one: a term without its transposed partner (src/solver.jl:1009), calls out of order,
    unsupported limb counts -- negative codes with a message, never a crash."""
    import clrs_amd
    from clrs_amd._lib import ClrsError
    from clrs_amd.mw import MwSchurContext
    from tests.test_hip_parity import _mini_sdp
    sdp = _mini_sdp(m=2, drop_partner=True)
    sdp.check = lambda: None
    with pytest.raises(ClrsError, match="transposed partner"):
        MwSchurContext(clrs_amd.flatten(sdp), limbs=4)
    f = flat("polyopt8")
    with pytest.raises(ClrsError, match="limbs"):
        MwSchurContext(f, limbs=7)
    ctx = MwSchurContext(f, limbs=3)
    K = 3
    with pytest.raises(ClrsError, match="before clrs_mw_schur_factor"):
        ctx.solve(np.zeros((K, f.x_len)), np.zeros((K, f.n_free)))
    with pytest.raises(ClrsError, match="before clrs_mw_schur_assemble"):
        ctx.factor()
    with pytest.raises(ValueError, match="planar limbs"):
        ctx.cholesky_blocks(np.zeros(f.xy_len))
    ctx.close()


def test_one_call_loop_equals_the_step_by_step_loop():
    """`clrs_mw_ipm_solve` (iterations enqueued one ahead of the record the host reads, termination test evaluated on the device, the
    iterate frozen once it holds) against one `clrs_mw_ipm_iterate` per iteration with the test on the host: the same kernels in the same
    order on the same data -- every table row, the objectives and the final iterate agree bit for bit; a maximum number of iterations
    gives error code 2 (src/solver.jl:362-366) and the loop can be continued from where it stopped."""
    from clrs_amd.mw import MwSchurContext, solvesdp_mw
    f = flat("ce_8_15")
    a = solvesdp_mw(f, limbs=5)
    b = solvesdp_mw(f, limbs=5, step_by_step=True)
    assert a.status == b.status == "Optimal" and a.iterations == b.iterations and a.error_code == b.error_code == 0
    assert np.array_equal(a.history, b.history)
    for u, v in ((a.x, b.x), (a.y, b.y), (a.X, b.X), (a.Y, b.Y), (a.timings["objectives_limbs"], b.timings["objectives_limbs"])):
        assert np.array_equal(u, v)
    c = solvesdp_mw(f, limbs=5, maxiterations=7)
    assert c.error_code == 2 and c.iterations == 7 and np.array_equal(c.history, a.history[:7])
    for kw in (dict(need_primal_feasible=True), dict(need_dual_feasible=True)):
        u, v = solvesdp_mw(f, limbs=5, **kw), solvesdp_mw(f, limbs=5, step_by_step=True, **kw)
        assert u.iterations == v.iterations and np.array_equal(u.history, v.history) and np.array_equal(u.X, v.X)



def _solve_sharded_in_threads(full, world, K=5, **kw):
    """`world` ranks of one process on one GPU, one thread each, exchanging through the in-process group: the whole sharded solve."""
    import threading
    from clrs_amd.mw import LocalGroup, MwSchurContext, shard_problem, solvesdp_mw
    group = LocalGroup(world)
    out, err = [None] * world, [None] * world

    def run(rank):
        try:
            shard, info = shard_problem(full, rank, world)
            ctx = MwSchurContext(shard, limbs=K)
            ctx.comm_init_local(group, rank)
            out[rank] = (solvesdp_mw(shard, ctx=ctx, shard_info=info, **kw), info)
            ctx.close()
        except Exception as e:          # a failing rank must not leave the others waiting in a collective: nothing to do but report
            err[rank] = e
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert all(e is None for e in err), err
    assert all(o is not None for o in out)
    group.close()
    return out


@pytest.mark.parametrize("which,world", [("multi3", 2), ("ns3", 2), ("ns3", 3)])
def test_sharded_interior_point_solve_on_one_gpu(which, world, oracle_built):
    """SURVEY.md section 8e, third row: a whole interior-point solve with the clusters sharded over `world` ranks (here: contexts of one
    process, one thread each, the all-gathers through the in-process group -- the same slots and the same rank-order reductions the RCCL
    path uses).  mu, the errors, p, beta_c, the step lengths and the objectives are reduced over the ranks inside the library; every rank
    ends with bit-identical y, objectives and table rows, and the solve agrees with the unsharded one (whose sums run in block order,
    not in rank order: the last bits differ).  `ns3` = Nsphere_packing(8, 15, [1/2, 1/2, 1/2]) with its 11 = N(N+1)/2 + N + 2 clusters
    (BASELINE config 3: "many small clusters sharded"), pinned like the N = 2 instance to pi^4/384 (test/runtests_solver.jl:21-22)."""
    import clrs_amd
    from clrs_amd.mw import solvesdp_mw
    from clrs_amd.problems import cohnelkies_multi, nsphere_packing
    from clrs_amd.sharded import partition_clusters
    if which == "multi3":
        full = clrs_amd.flatten(cohnelkies_multi(8, 15, [1.0, 1.125, 1.25]))
    else:
        full = clrs_amd.flatten(nsphere_packing(8, 15, [0.5, 0.5, 0.5]))
        assert full.n_clusters == 11
    parts = partition_clusters(full, world)
    assert sorted(j for p in parts for j in p) == list(range(full.n_clusters)) and all(parts)
    K = 5                                      # the reference's default prec = 256 (its own Nsphere_packing test asks for prec = 300: 6 limbs)
    ref = solvesdp_mw(full, limbs=K)
    assert ref.error_code == 0 and ref.status == "Optimal", (ref.status, ref.error_code, ref.iterations, ref.duality_gap)
    if which == "ns3":
        g = _golden_config("ns_8_15_3")          # the 256-bit oracle: 60 iterations, 0.25374045328578, errors 1.4e-37 / 5.2e-44
        assert abs(ref.primal_objective - PI4_384) <= 1e-4 and ref.iterations == g["iterations"]
        assert abs(ref.primal_objective - g["p_obj"]) <= 1e-12
    res = _solve_sharded_in_threads(full, world, K=K)
    r0 = res[0][0]
    assert r0.error_code == 0 and r0.status == "Optimal", (r0.status, r0.error_code)
    for r, _ in res[1:]:
        assert np.array_equal(r.y, r0.y) and np.array_equal(r.history, r0.history)
        assert np.array_equal(r.timings["objectives_limbs"], r0.timings["objectives_limbs"])
    assert r0.iterations == ref.iterations
    assert abs(r0.primal_objective - ref.primal_objective) <= 1e-12 * max(1.0, abs(ref.primal_objective))
    assert np.allclose(r0.history[:, [1, 8, 9]], ref.history[:, [1, 8, 9]], rtol=1e-9, atol=0)
    # max|P|, max|p|, max|d| of every iteration: maxima over the ranks' records (stage 15: -B^T x and max|P| with the objectives; stage 2: max|d|)
    # (above the rounding floor of the residuals, where sums in rank order and sums in block order differ by factors)
    he, hr = r0.history[:, [5, 6, 7]], ref.history[:, [5, 6, 7]]
    big = hr > 1e-30 * np.max(hr, axis=0)
    assert big.sum() > 3 * 10 and np.allclose(he[big], hr[big], rtol=1e-6, atol=0)
    # the shards' x, X, Y are the unsharded solution's rows / blocks
    for r, info in res:
        cols = np.concatenate([np.arange(int(full.block_off[b]), int(full.block_off[b + 1])) for b in info["block_ids"]])
        rows = np.concatenate([np.arange(int(full.cluster_off[j]), int(full.cluster_off[j + 1])) for j in info["cluster_ids"]])
        assert np.allclose(r.Y[0], ref.Y[0][cols], rtol=1e-9, atol=1e-300) and np.allclose(r.x[0], ref.x[0][rows], rtol=1e-7, atol=1e-300)


@pytest.mark.parametrize("K", [3, 4, 5, 6])
@pytest.mark.parametrize("name", ["ce_8_15", "ns_8_15_2", "polyopt8", "delsarte_3_10", "threepoint_4", "polyopt40"])
def test_exact_product_pairings_match_the_oracle(name, K, oracle_built):
    """k_mws_pair (csrc/clrs_mw_exact.hip.h): the pairing matrices V^T X^-1 V and V^T Y V through 23-bit slices whose products and k-sums
    are exact in the fp64 MFMA accumulators, instead of K-limb expansions -- the same S_j and A_Y as the oracle to the same tolerance as the
    expansion kernels (the scheme cuts at 2^-(23 S) relative to row / column maxima, S = 8 / 10 / 12 / 15 slices for K = 3 / 4 / 5 / 6: below
    the expansions' own rounding), on blocks of sides 1 ... 32 with and without sub-blocks, rank-1 and rank-2 terms."""
    import torch
    from clrs_amd.mw import MwSchurContext
    from oracle.oracle import Oracle
    f = flat(name)
    X, Y = _iterates(f, K)
    X, Y = _sym_limbs(f, X), _sym_limbs(f, Y)
    ctx = MwSchurContext(f, limbs=K, exact_products=True)
    dX, dY = torch.tensor(X, device="cuda:0"), torch.tensor(Y, device="cuda:0")
    dXc = torch.empty_like(dX)
    ctx.cholesky_blocks_dev(dX.data_ptr(), dXc.data_ptr())        # the exact-product kernel uses the inverse factors this call leaves in the context
    ctx.assemble_dev(dXc.data_ptr(), dY.data_ptr())
    S, AY = ctx.get_S()
    Xc = dXc.cpu().numpy()
    o = Oracle(f, mp_bits=320 if K <= 5 else 640)
    pad = lambda a: np.vstack([a, np.zeros((1, a.shape[1]))])
    S_ref, AY_ref = o.schur_assemble_mw(pad(Xc), pad(Y))
    assert mw_relerr(S, S_ref) <= tol(K, 22), ("S", np.log2(mw_relerr(S, S_ref)))
    if f.n_terms:
        assert mw_relerr(AY, AY_ref, scale=max(1.0, np.max(np.abs(AY_ref[0])))) <= tol(K, 16)
    for j in range(f.n_clusters):
        P = int(f.cluster_P[j]); o0 = int(f.S_off[j])
        for l in range(K):
            Sj = S[l, o0:o0 + P * P].reshape(P, P, order="F")
            assert np.array_equal(Sj, Sj.T)
    assert ctx.factor() == (0 if not (name in ("ce_8_15", "ns_8_15_2") and K <= 3) else ctx.sync_status())
    ctx.close()


@pytest.mark.parametrize("kw", [
    dict(seed=1, J=3, n_free=1, definite=True),                              # the shapes of the named problems
    dict(seed=10, J=2, n_free=5, fixed_P=70, max_n=40, lr_blocks=2),         # tiled block products (n = 40), blocked factorisation (P = 70), row-parallel solve
    dict(seed=11, J=1, n_free=0, fixed_P=60, max_n=33, lr_blocks=3),         # no free variables, one cluster beyond LDS
    dict(seed=12, J=40, n_free=2, fixed_P=6, max_n=4, lr_blocks=2),          # many small clusters (k_mw_saccum_one)
    dict(seed=13, J=2, n_free=70, fixed_P=40, max_n=20, lr_blocks=2),        # Q larger than the clusters (blocked Q, N = 70)
])
def test_device_loop_follows_the_oracle_on_random_sdps(kw, oracle_built):
    """The device-resident loop against the 256-bit oracle on random SDPs of awkward shapes -- most are infeasible or unbounded, and both must
    walk the same trajectory: mu, both step lengths and beta_c of the first 12 iterations to 1e-8, the same error code at the end."""
    import clrs_amd
    from tests.util import random_simple_sdp
    from clrs_amd.mw import solvesdp_mw
    from oracle.oracle import Oracle
    from clrs_amd.mw import MwSchurContext
    f = clrs_amd.flatten(random_simple_sdp(**kw))
    o = Oracle(f, mp_bits=256)
    o.set_num_threads(8)
    ro = o.solvesdp(maxiterations=12)
    for exact in ((None, True) if kw["seed"] in (1, 12) else (None,)):        # also with the pairing matrices forced through the exact slice products
        ctx = MwSchurContext(f, limbs=5, exact_products=exact)
        r = solvesdp_mw(f, limbs=5, maxiterations=12, ctx=ctx)
        ctx.close()
        assert r.error_code == ro["error_code"] and r.iterations == ro["iterations"], exact
        for it in range(len(ro["hist"])):
            for col in (1, 8, 9, 10):
                a, b = r.history[it, col], ro["hist"][it, col]
                assert abs(a - b) <= 1e-8 * max(abs(b), 1e-300), (exact, it, col, a, b)


@pytest.mark.parametrize("K", [4, 5, 6])
def test_exact_product_pairings_of_a_large_block(K, oracle_built):
    """k_mwx_slice / k_mwx_gram (csrc/clrs_mw_exact.hip.h): the pairing matrices of a block beyond the shapes of k_mws_pair -- 41 x 41 with 81 unique
    vectors: digits of Z and T in global memory, 6 x 6 tiles of 16 x 16 per matrix, 44 rows of k in two exact accumulations of at most 32 -- against
    the oracle like the expansion kernels; the same context with the exact products switched off gives the same S to the tolerance of either."""
    import torch
    from clrs_amd.mw import MwSchurContext
    from oracle.oracle import Oracle
    f = flat("polyopt80")
    X, Y = _iterates(f, K)
    X, Y = _sym_limbs(f, X), _sym_limbs(f, Y)
    o = Oracle(f, mp_bits=320 if K <= 5 else 640)
    pad = lambda a: np.vstack([a, np.zeros((1, a.shape[1]))])
    res = {}
    for exact in (True, False):
        ctx = MwSchurContext(f, limbs=K, exact_products=exact)
        dX, dY = torch.tensor(X, device="cuda:0"), torch.tensor(Y, device="cuda:0")
        dXc = torch.empty_like(dX)
        ctx.cholesky_blocks_dev(dX.data_ptr(), dXc.data_ptr())
        ctx.assemble_dev(dXc.data_ptr(), dY.data_ptr())
        S, AY = ctx.get_S()
        if exact:
            S_ref, AY_ref = o.schur_assemble_mw(pad(dXc.cpu().numpy()), pad(Y))
        assert mw_relerr(S, S_ref) <= tol(K, 22), (exact, np.log2(mw_relerr(S, S_ref)))
        assert mw_relerr(AY, AY_ref, scale=max(1.0, np.max(np.abs(AY_ref[0])))) <= tol(K, 16), exact
        res[exact] = S
        ctx.close()
    assert not np.array_equal(res[True], res[False])        # two different arithmetics (the exact path was taken), one answer
    assert mw_relerr(res[True], res[False]) <= tol(K, 22)


@pytest.mark.parametrize("K", [4, 5, 6])
def test_exact_products_of_dense_blocks(K, oracle_built):
    """k_mwx_dense (csrc/clrs_mw_exact.hip.h): T_e = X^-1 (A_e Y) of dense 32 x 32 blocks as three chained exact slice products per matrix (static
    digits of A_e, digits of Y, Xi^T, Xi and of the intermediate results in LDS) -- S_j against the oracle within the tolerance of the expansion
    kernels, and against the expansion kernels themselves (sdpa_scaled(8, 32, 40): 8 dense blocks, 40 constraints on random block pairs)."""
    import torch
    from clrs_amd.mw import MwSchurContext
    from oracle.oracle import Oracle
    f = flat("sdpa_mid")
    X, Y = _iterates(f, K)
    X, Y = _sym_limbs(f, X), _sym_limbs(f, Y)
    o = Oracle(f, mp_bits=320 if K <= 5 else 640)
    pad = lambda a: np.vstack([a, np.zeros((1, a.shape[1]))])
    res = {}
    for exact in (True, False):
        ctx = MwSchurContext(f, limbs=K, exact_products=exact)
        dX, dY = torch.tensor(X, device="cuda:0"), torch.tensor(Y, device="cuda:0")
        dXc = torch.empty_like(dX)
        ctx.cholesky_blocks_dev(dX.data_ptr(), dXc.data_ptr())
        ctx.assemble_dev(dXc.data_ptr(), dY.data_ptr())
        S, _ = ctx.get_S()
        if exact:
            S_ref, _ = o.schur_assemble_mw(pad(dXc.cpu().numpy()), pad(Y))
        assert mw_relerr(S, S_ref) <= tol(K, 22), (exact, np.log2(mw_relerr(S, S_ref)))
        res[exact] = S
        ctx.close()
    assert not np.array_equal(res[True], res[False])        # two different arithmetics (the exact path was taken), one answer
    assert mw_relerr(res[True], res[False]) <= tol(K, 22)


@pytest.mark.parametrize("K", [4, 5])
def test_exact_product_pairings_on_the_trajectory_fixture(K):
    """The same on real interior-point iterates (tests/golden/ce_8_15_traj.npz, mu from 1e20 to 2e-16, cond(X) up to 2^56): S against the
    456-bit dense trace formula within the bound of the expansion kernels."""
    import os
    import torch
    from clrs_amd.mw import MwSchurContext
    f = flat("ce_8_15")
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ce_8_15_traj.npz"))
    ctx = MwSchurContext(f, limbs=K, exact_products=True)
    for s, it in enumerate(g["iters"]):
        X, Y = np.ascontiguousarray(g["X"][s][:K]), np.ascontiguousarray(g["Y"][s][:K])
        dX, dY = torch.tensor(X, device="cuda:0"), torch.tensor(Y, device="cuda:0")
        dXc = torch.empty_like(dX)
        ctx.cholesky_blocks_dev(dX.data_ptr(), dXc.data_ptr())
        ctx.assemble_dev(dXc.data_ptr(), dY.data_ptr())
        S, _ = ctx.get_S()
        eS = mw_relerr(S, g["S"][s])
        assert eS <= 2.0 ** -(53 * K - 6 - COND_X_BITS[int(it)]), (it, np.log2(eS))
    ctx.close()
