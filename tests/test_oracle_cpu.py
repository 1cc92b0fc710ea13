"""CPU tests (no GPU): the oracle (oracle/clrs_oracle.c) against
  * the 256-bit golden vectors of tests/golden/ (independent dense / LU restatement, make_golden.py),
  * the objective values the reference's own tests pin for this path,
  * the structural identities of the path.
The oracle is the checker of the GPU parity tests; these tests pin the checker."""
import json
import os

import numpy as np
import pytest

from tests.util import chol_blocks_np, flat, spd_iterates

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FULL = ["x2p1", "polyopt8", "delsarte_8_3", "ce_8_3", "ns_8_3_2", "sdpa_small", "polyopt40", "delsarte_3_10", "threepoint_4"]
S_ONLY = ["ce_8_15", "ns_8_15_2"]


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def fingerprint(f):
    parts = [f.cluster_P, f.block_m, f.block_delta, f.block_kind, f.term_p, f.term_r, f.term_s]
    h = sum(int(np.sum(np.asarray(a, dtype=np.int64) * (np.arange(len(a)) + 1))) for a in parts)
    v = float(np.sum(f.term_vs)) + float(np.sum(f.term_lambda)) + float(np.sum(f.dense_A)) + float(np.sum(f.B))
    return np.array([h, v])


@pytest.mark.parametrize("name", FULL + S_ONLY)
def test_golden_inputs_match_generators(name):
    """The fixtures were produced from the problems the generators produce today."""
    g = load(name)
    fp = fingerprint(flat(name))
    assert fp[0] == g["fingerprint"][0]
    assert abs(fp[1] - g["fingerprint"][1]) <= 1e-9 * max(1.0, abs(fp[1]))


@pytest.mark.parametrize("quad", [False, True])
@pytest.mark.parametrize("name", FULL + S_ONLY)
def test_oracle_assembly_matches_256bit_golden(name, quad, oracle_built):
    """S from bilinear pairings (oracle) == S from the dense definition at 256 bits.
    Tolerance: fp64 oracle 1e-12 * max|S| (accumulated rounding of ~n + U term sums), quad oracle 1e-15."""
    from oracle.oracle import Oracle
    f, g = flat(name), load(name)
    S, _ = Oracle(f, quad=quad, use_lo=False).schur_assemble(g["Xchol"], g["Y"])
    tol = 1e-15 if quad else 1e-12
    assert np.max(np.abs(S - g["S"])) <= tol * np.max(np.abs(g["S"]))


@pytest.mark.parametrize("name", FULL)
def test_oracle_solve_matches_256bit_golden(name, oracle_built):
    """(dx, dy) from the block-Cholesky path == LU solve of the full KKT system at 256 bits.
    Tolerance 1e-7 relative: the test iterates give cond(S) up to ~1e8 (ce_8_3), fp64 eps * cond."""
    from oracle.oracle import Oracle
    f, g = flat(name), load(name)
    loose = 1e-5 if name == "threepoint_4" else 1e-7     # cond(S) ~ 1e10 at these iterates for the three-point instance
    for quad, tol in ((False, loose), (True, 1e-12)):
        o = Oracle(f, quad=quad, use_lo=False)
        o.schur_assemble(g["Xchol"], g["Y"])
        assert o.schur_factor() == 0
        dx, dy = o.schur_solve(g["rhs_x"], g["rhs_y"])
        scale = max(1.0, np.max(np.abs(g["dx"])), np.max(np.abs(g["dy"])) if f.n_free else 0.0)
        assert np.max(np.abs(dx - g["dx"])) <= tol * scale, (quad, np.max(np.abs(dx - g["dx"])) / scale)
        if f.n_free:
            assert np.max(np.abs(dy - g["dy"])) <= tol * scale


@pytest.mark.parametrize("name", ["polyopt8", "delsarte_8_3", "ns_8_3_2", "sdpa_small", "ce_8_3"])
def test_lowrank_assembly_equals_dense_trace_formula(name, oracle_built):
    """SURVEY.md section 8c identity: low-rank S == Tr(A_p X^-1 A_q Y) with A_p = Matrix(::LowRankMat)."""
    from oracle.oracle import Oracle
    f = flat(name)
    X, Y = spd_iterates(f, seed=21)
    Xc = chol_blocks_np(f, X)
    o = Oracle(f, quad=True, use_lo=False)
    S, _ = o.schur_assemble(Xc, Y)
    Sd = o.schur_dense_check(Xc, Y)
    assert np.max(np.abs(S - Sd)) <= 1e-14 * np.max(np.abs(S))
    for j in range(f.n_clusters):
        P = int(f.cluster_P[j])
        Sj = S[f.S_off[j]:f.S_off[j + 1]].reshape(P, P)
        assert np.array_equal(Sj, Sj.T)                     # symmetric! (src/tools.jl:43-57)
        assert np.linalg.eigvalsh(Sj)[0] > 0                # PSD blocks + SPD iterates => S positive definite


# objective values pinned by the reference's own tests / docs for problems that run through this path
PINNED = [
    ("x2p1", 1.0, 1e-6, "README.md:149 (min of x^2+1)"),
    ("delsarte_3_10", 13.158314, 1e-5, "test/runtests_solver.jl:15"),
    ("delsarte_8_3", 240.0, 1e-4, "test/runtests_solver.jl:86-87 (exact 240)"),
]


def test_sdpa_writer_round_trips(tmp_path):
    """`write_sdpa` (the export half of src/SDPAtoCLRS.jl's format): the reference's example file and a synthetic instance with a diagonal block
    survive write -> read unchanged (block sizes incl. the sign of a diagonal block, c, every matrix), and flatten to the same arrays."""
    import clrs_amd
    from clrs_amd.problems import read_sdpa, sdpa_scaled, sdpa_to_sdp, write_sdpa
    from clrs_amd.problems.sdpa import SDPAData
    a = read_sdpa(os.path.join(GOLD, "example.dat-s"))
    b = sdpa_scaled(nb=3, bs=4, m=5, seed=7)
    rng = np.random.default_rng(3)
    diag = [np.diag(rng.standard_normal(3)) for _ in range(b.m + 1)]
    b = SDPAData(b.m, b.block_sizes + [-3], b.c, [F + [diag[k]] for k, F in enumerate(b.F)])
    for i, d in enumerate((a, b)):
        path = str(tmp_path / f"rt{i}.dat-s")
        write_sdpa(path, d)
        e = read_sdpa(path)
        assert e.m == d.m and e.block_sizes == d.block_sizes and np.array_equal(e.c, d.c)
        assert all(np.array_equal(x, y) for Fe, Fd in zip(e.F, d.F) for x, y in zip(Fe, Fd))
        fe, fd = clrs_amd.flatten(sdpa_to_sdp(e)), clrs_amd.flatten(sdpa_to_sdp(d))
        assert np.array_equal(fe.dense_A, fd.dense_A) and np.array_equal(fe.C, fd.C) and np.array_equal(fe.c, fd.c) and list(fe.block_n) == list(fd.block_n)


def test_sdpa_example_file_parses_like_the_reference(oracle_built):
    """BASELINE config 5: test/example.dat-s (copied as a data fixture).  The reference pins only the parse
    (test/runtests_solver.jl:228-235): F0 block 2 = [3 0; 0 4], F2 block 2 = [5 2; 2 6]; the empty third constraint is
    dropped (src/SDPAtoCLRS.jl:66-78).  The solve itself is unpinned by the reference: fp64 and quad oracle must agree
    and satisfy weak duality."""
    from clrs_amd.problems import read_sdpa
    from oracle.oracle import Oracle
    d = read_sdpa(os.path.join(GOLD, "example.dat-s"))
    assert d.m == 3 and d.block_sizes == [2, 2] and list(d.c) == [10.0, 20.0, 0.0]
    assert np.array_equal(d.F[0][1], [[3, 0], [0, 4]]) and np.array_equal(d.F[2][1], [[5, 2], [2, 6]])
    f = flat("sdpa_example")
    assert list(f.cluster_P) == [2] and f.n_free == 0 and list(f.block_kind) == [1, 1]
    res = [Oracle(f, quad=q).solvesdp(omega_p=1e2, omega_d=1e2, duality_gap_threshold=1e-7, dual_error_threshold=1e-9,
                                      primal_error_threshold=1e-9) for q in (False, True)]
    assert all(r["error_code"] == 0 for r in res)
    assert abs(res[0]["p_obj"] - res[1]["p_obj"]) <= 1e-6 * max(1.0, abs(res[1]["p_obj"]))
    assert all(abs(r["p_obj"] - r["d_obj"]) <= 1e-6 * max(1.0, abs(r["p_obj"])) for r in res)


def test_threepoint_generator_reproduces_the_reference_instance(oracle_built):
    """BASELINE config 4: three_point_spherical_codes(4, 1//6, -1, 4) (test/runtests_solver.jl:26-27): block structure as the
    reference builds it and the pinned objective 10 +- 1e-5 with omega = 1e3, as in the reference's test."""
    from oracle.oracle import Oracle
    f = flat("threepoint_4")
    assert list(f.cluster_P) == [50] and f.n_free == 0
    assert [int(n) for n, kd in zip(f.block_n, f.block_kind) if kd == 1] == [5, 4, 3, 2, 1]
    assert [int(n) for n, kd in zip(f.block_n, f.block_kind) if kd == 0] == [5, 4, 11, 2, 11, 7, 1, 6, 4, 3, 2, 1, 4, 3]
    r = Oracle(f, quad=True).solvesdp(omega_p=1e3, omega_d=1e3, duality_gap_threshold=1e-7, dual_error_threshold=1e-9,
                                      primal_error_threshold=1e-9)
    assert r["error_code"] == 0
    assert abs(r["p_obj"] - 10.0) <= 1e-5 and abs(r["d_obj"] - 10.0) <= 1e-5


@pytest.mark.parametrize("name,expected,tol,src", PINNED)
def test_oracle_loop_reproduces_reference_pinned_objectives(name, expected, tol, src, oracle_built):
    from oracle.oracle import Oracle
    for quad in (False, True):
        r = Oracle(flat(name), quad=quad).solvesdp(omega_p=1e4, omega_d=1e4, duality_gap_threshold=1e-8,
                                                   dual_error_threshold=1e-9, primal_error_threshold=1e-9)
        assert r["error_code"] == 0, (src, quad, r["error_code"])
        assert abs(r["p_obj"] - expected) <= tol * max(1.0, abs(expected)), (src, quad, r["p_obj"])
        assert abs(r["d_obj"] - expected) <= tol * max(1.0, abs(expected)), (src, quad, r["d_obj"])


def test_oracle_loop_polyopt_matches_independent_minimum(oracle_built):
    """polyopt 2d=40 (BASELINE config 2): the SOS bound equals the true minimum of the univariate polynomial,
    found independently by root-finding of f' (numpy companion matrix)."""
    from clrs_amd.problems import polyopt_random
    from oracle.oracle import Oracle
    sdp, coef = polyopt_random(20, seed=0)
    import clrs_amd
    r = Oracle(clrs_amd.flatten(sdp), quad=True).solvesdp(omega_p=1e4, omega_d=1e4, duality_gap_threshold=1e-8,
                                                          dual_error_threshold=1e-9, primal_error_threshold=1e-9)
    assert r["error_code"] == 0
    c = np.polynomial.chebyshev.Chebyshev(coef)
    crit = c.deriv().roots()
    crit = crit[np.abs(crit.imag) < 1e-7].real               # f - lambda is SOS on all of R: global minimum
    fmin = min(c(x) for x in crit)
    assert abs(r["p_obj"] - fmin) <= 1e-6 * max(1.0, abs(fmin))


def test_dedup_counts_follow_the_reference_convention(oracle_built):
    """precompute_matrices_bilinear_pairings de-duplicates by exact equality (src/solver.jl:985-1059, comment :988):
    the 96-constraint cluster of Nsphere_packing(8,15,[1/2,1/2],2) collapses to 32 unique vectors per sub-block row."""
    from oracle.oracle import Oracle
    f = flat("ns_8_15_2")
    o = Oracle(f)
    big = [b for b in range(f.n_blocks) if f.block_n[b] == 32]
    assert len(big) == 2
    for b in big:
        UR, UL = o.unique_counts(b)
        assert list(UR) == [32, 32] and list(UL) == [32, 32]


def test_oracle_reports_nonpositive_pivot_like_the_reference(oracle_built):
    """approx_cholesky! returns 0 on a non-positive pivot (src/tools.jl:92-95) -> SolverFailure naming the block
    (src/solver.jl:1249); the oracle returns j+1."""
    from oracle.oracle import Oracle
    f = flat("ns_8_3_2")
    X, Y = spd_iterates(f, seed=5)
    Xc = chol_blocks_np(f, X)
    b = 1
    n = int(f.block_n[b])
    Yb = Y[f.block_off[b]:f.block_off[b + 1]].reshape(n, n)
    Yb[:] = -Yb                                              # Y block negative definite => S_j not PD
    o = Oracle(f)
    o.schur_assemble(Xc, Y)
    assert o.schur_factor() == int(f.block_cluster[b]) + 1


def test_sdp_variant_helpers_against_the_oracle(oracle_built):
    """Host logic used by the GPU parity tests and by bench.py (cluster replication, constraint relabelling, block duplication),
    checked on the CPU against the oracle's own algebra: relabelling the constraints permutes S_j, a duplicated dense 1 x 1 block with
    half the entries adds a quarter of its contribution, replicated clusters assemble independently."""
    from clrs_amd.sdp import replicate_clusters
    from oracle.oracle import Oracle
    from tests.util import chol_blocks_np, duplicate_block, flat, permute_cluster_constraints, spd_iterates
    f = flat("ce_8_3")
    X, Y = spd_iterates(f, seed=3)
    Xc = chol_blocks_np(f, X)
    S, _ = Oracle(f, quad=False).schur_assemble(Xc, Y)
    # relabelled constraints: S'[perm p, perm q] = S[p, q]
    g = permute_cluster_constraints(f, seed=5)
    Sg, _ = Oracle(g, quad=False).schur_assemble(Xc, Y)
    for j in range(f.n_clusters):
        P = int(f.cluster_P[j])
        a = S[f.S_off[j]:f.S_off[j + 1]].reshape(P, P, order="F")
        b = Sg[g.S_off[j]:g.S_off[j + 1]].reshape(P, P, order="F")
        assert np.allclose(np.sort(np.linalg.eigvalsh(a)), np.sort(np.linalg.eigvalsh(b)), rtol=1e-10, atol=1e-12 * np.max(np.abs(a)))
        assert np.isclose(np.trace(a), np.trace(b), rtol=1e-12)
    # duplicated dense 1 x 1 block (entries x 0.5): its contribution a a^T Y / X is added once more, scaled by 0.25, when the copy's
    # iterates equal the original's
    d = [b for b in range(f.n_blocks) if f.block_kind[b] == 1][0]
    h = duplicate_block(f, d, 0.5)
    x0, x1 = int(f.block_off[d]), int(f.block_off[d + 1])
    Xc2 = np.concatenate([Xc[:x1], Xc[x0:x1], Xc[x1:]])
    Y2 = np.concatenate([Y[:x1], Y[x0:x1], Y[x1:]])
    Sh, _ = Oracle(h, quad=False).schur_assemble(Xc2, Y2)
    # S without the dense block, from a copy whose dense entries are zero
    import copy
    z = copy.copy(f)
    z.dense_A = f.dense_A.copy()
    z.dense_A[int(f.dense_A_ptr[f.dense_ptr[d]]):int(f.dense_A_ptr[f.dense_ptr[d + 1]])] = 0.0
    S0, _ = Oracle(z, quad=False).schur_assemble(Xc, Y)
    assert np.allclose(Sh - S0, 1.25 * (S - S0), rtol=1e-10, atol=1e-13 * np.max(np.abs(S)))
    # replication: cluster k of the big instance is the small problem at its own iterates
    big = replicate_clusters(f, 3)
    Xb, Yb = spd_iterates(big, seed=8)
    Xcb = chol_blocks_np(big, Xb)
    Sb, _ = Oracle(big, quad=False).schur_assemble(Xcb, Yb)
    o = Oracle(f, quad=False)
    for k in range(3):
        Sk, _ = o.schur_assemble(Xcb[k * f.xy_len:(k + 1) * f.xy_len], Yb[k * f.xy_len:(k + 1) * f.xy_len])
        assert np.array_equal(Sb[k * f.S_len:(k + 1) * f.S_len], Sk)


# ---- the multi-precision oracle (oracle/mpx.hpp): the stand-in for the reference's Arb arithmetic --------------------------
PI4_384 = 0.25366950790104804          # pi^4 / 384 (test/runtests_solver.jl:19-22)


def _limbs(vals, K):
    import mpmath as mp
    out = np.zeros((K, len(vals)))
    for i, v in enumerate(vals):
        r = mp.mpf(v)
        for l in range(K):
            h = float(r); out[l, i] = h; r -= mp.mpf(h)
    return out


def _vals(a):
    import mpmath as mp
    return [mp.fsum(mp.mpf(float(a[l, i])) for l in range(a.shape[0])) for i in range(a.shape[1])]


@pytest.mark.parametrize("bits", [128, 256, 320])
def test_mpx_arithmetic_against_mpmath(bits, oracle_built):
    """add / sub / mul / div / sqrt of oracle/mpx.hpp, truncated to `bits` bits: within 2 units of 2^-bits of the exact result
    (truncation, like Arb's approx_* functions), including near-cancelling subtractions."""
    import mpmath as mp
    from oracle.oracle import real_op
    mp.mp.prec = 1200
    rng = np.random.default_rng(bits)
    K, n = 7, 200
    av, bv = [], []
    for i in range(n):
        a = mp.fsum(mp.mpf(float(rng.standard_normal())) * mp.mpf(2) ** (int(rng.integers(-30, 30)) - 50 * l) for l in range(K))
        b = mp.fsum(mp.mpf(float(rng.standard_normal())) * mp.mpf(2) ** (int(rng.integers(-30, 30)) - 50 * l) for l in range(K))
        if i % 5 == 0:
            b = a * (1 + mp.mpf(2) ** -int(rng.integers(20, bits - 20)))
        av.append(a); bv.append(b)
    A, B = _limbs(av, K), _limbs(bv, K)
    # the oracle reads the limbs exactly when they fit: keep operands of at most `bits` bits
    def trunc(vals):
        out = []
        for v in vals:
            m, e = mp.frexp(v)
            out.append(mp.ldexp(mp.floor(abs(m) * mp.mpf(2) ** bits), e - bits) * (1 if v >= 0 else -1))
        return out
    av, bv = trunc(_vals(A)), trunc(_vals(B))
    A, B = _limbs(av, K), _limbs(bv, K)
    unit = mp.mpf(2) ** -bits
    for op, ex in (("add", [x + y for x, y in zip(av, bv)]), ("sub", [x - y for x, y in zip(av, bv)]), ("mul", [x * y for x, y in zip(av, bv)]),
                   ("div", [x / y for x, y in zip(av, bv)]), ("sqrt", [mp.sqrt(abs(x)) for x in av])):
        a_in = _limbs([abs(x) for x in av], K) if op == "sqrt" else A
        got = _vals(real_op(op, a_in, B, mp_bits=bits))
        for g, e, x, y in zip(got, ex, av, bv):
            scale = abs(e) if op not in ("add", "sub") else max(abs(x), abs(y))
            assert abs(g - e) <= 2.5 * unit * scale, (op, float(abs(g - e) / scale / unit))


def test_mp_oracle_reproduces_the_reference_pinned_objective_of_the_north_star_config(oracle_built):
    """cohnelkies(8,15) at 256 bits with the reference's default options: test/runtests_solver.jl:19-20 pins pi^4/384 +- 1e-4."""
    from oracle.oracle import Oracle
    r = Oracle(flat("ce_8_15"), mp_bits=256).solvesdp()
    assert r["error_code"] == 0 and r["pd_feas"]
    assert abs(r["p_obj"] - PI4_384) <= 1e-4 and abs(r["d_obj"] - PI4_384) <= 1e-4
    assert r["gap"] < 1e-15 and r["dual_error"] < 1e-30 and r["primal_error"] < 1e-30
    assert 50 <= r["iterations"] <= 62


def test_precision_sweep_of_the_north_star_config(oracle_built):
    """What working precision this problem needs: 106 and 113 bits lose positive definiteness of S at once,
    160 bits reach the pinned objective but not feasibility, 212 bits reach gap 1e-12, 256 bits the reference's default thresholds."""
    from oracle.oracle import Oracle
    f = flat("ce_8_15")
    for bits in (106, 113):
        r = Oracle(f, mp_bits=bits).solvesdp()
        assert r["error_code"] == 1 and r["iterations"] <= 2
    r = Oracle(f, mp_bits=160).solvesdp()
    assert abs(r["p_obj"] - PI4_384) <= 1e-4 and r["error_code"] == 1          # SolverFailure later on: current iterate returned
    r = Oracle(f, mp_bits=212).solvesdp(dual_error_threshold=1e-25, primal_error_threshold=1e-25, duality_gap_threshold=1e-12)
    assert r["error_code"] == 0 and abs(r["p_obj"] - PI4_384) <= 1e-4


def test_fp64_rounded_data_is_a_different_problem(oracle_built):
    """The same SDP with B, c, vectors rounded to fp64 diverges (mu too large, code 3): the problem data need more than fp64 too."""
    from oracle.oracle import Oracle
    r = Oracle(flat("ce_8_15"), mp_bits=256, use_lo=False).solvesdp()
    assert r["error_code"] == 3


def test_mp_oracle_loop_follows_the_independent_mpmath_loop(oracle_built):
    """tests/golden/ce_8_15_ipm.npz: the whole interior-point loop restated in mpmath (dense trace formula, LU of the KKT matrix)
    at 256 bits.  The oracle (bilinear pairings, block Cholesky, its own arithmetic) takes the same number of iterations to the same
    objective and follows its mu / step-length trace (the step lengths go through an fp64 eigenvalue in both: 1e-6)."""
    from oracle.oracle import Oracle
    g = np.load(os.path.join(GOLD, "ce_8_15_ipm.npz"))
    r = Oracle(flat("ce_8_15"), mp_bits=256).solvesdp()
    assert r["iterations"] == int(g["iterations"][0])
    assert abs(r["p_obj"] - float(np.sum(g["p_obj"]))) <= 1e-13 and abs(r["d_obj"] - float(np.sum(g["d_obj"]))) <= 1e-13
    assert abs(float(np.sum(g["p_obj"])) - PI4_384) <= 1e-4
    h, hg = r["hist"], g["hist"]
    for it in range(min(len(h), len(hg))):
        assert abs(h[it, 1] - hg[it, 1]) <= 1e-6 * hg[it, 1], (it, h[it, 1], hg[it, 1])                 # mu
        assert abs(h[it, 8] - hg[it, 8]) <= 1e-6 and abs(h[it, 9] - hg[it, 9]) <= 1e-6, it             # alpha_d, alpha_p


def test_mp_oracle_on_the_trajectory_fixture(oracle_built):
    """tests/golden/ce_8_15_traj.npz: (X, Y, rhs) at iterations 1, 2, K/2, K-1 of that mpmath run (mu from 1e20 to 1e-15) with S from
    the dense definition and (dx, dy) from an LU solve of the KKT matrix at 456 bits.  The oracle at 320 bits: S to 2^-250 relative (2^-316 on the first iterate);
    dx, dy to what cond(S) leaves (stated per iterate)."""
    from oracle.oracle import Oracle
    from tests.util import mw_relerr
    f = flat("ce_8_15")
    g = np.load(os.path.join(GOLD, "ce_8_15_traj.npz"))
    o = Oracle(f, mp_bits=320)
    for s, it in enumerate(g["iters"]):
        X, Y = g["X"][s], g["Y"][s]
        st, Xc = o.cholesky_blocks_mw(X)
        assert st == 0
        S, _ = o.schur_assemble_mw(Xc, Y)
        assert mw_relerr(S, g["S"][s]) <= 2.0 ** -250, (it, mw_relerr(S, g["S"][s]))      # cond(X) of the late iterates: 2^-316 at iteration 1, 2^-263 at 55
        assert o.schur_factor() == 0
        dx, dy = o.schur_solve_mw(g["rhs_x"][s], g["rhs_y"][s])
        assert mw_relerr(dx, g["dx"][s]) <= 2.0 ** -150, (it, mw_relerr(dx, g["dx"][s]))
        assert mw_relerr(dy, g["dy"][s]) <= 2.0 ** -150, (it, mw_relerr(dy, g["dy"][s]))
