"""The tightest vectors the REFERENCE itself holds for the path, reproduced by the CPU oracle (CPU tests) and by the HIP path
(`-m gpu`), every one of them through `compute_S_integrated!` + factorisation + solve on every iteration:

* `min_f(2)` (examples/PolyOpt.jl:40-86; rank-2 term in a two-block cluster): the solver log of docs/src/solving.md:38-51 --
  56 iterations; mu, dual / primal objective, gap, the three errors and both step lengths of iterations 1-3 and 55-56 to the printed
  digits (+ 1e-3 relative: the reference's step length comes from a randomly started Float64 Lanczos with a 1e-5 safety margin,
  src/solver.jl:1659-1686, ours from an exact fp64 eigenvalue with the same margin); the 77-digit final objectives to 1e-14 (the
  iterate the loop stops at is an O(gap) = 8e-16 neighbour of the optimum, so more digits are not comparable between two
  implementations of the step length); BASELINE.md section 1 "use as golden value / trace check".
* theta(C_5) = sqrt(5) and the POVM value sqrt(2)/4 + 1/2 at `duality_gap_threshold = 1e-30` (examples/jump.jl:4-55,
  test/moi_tests.jl:5-10: atol 1e-30 after rounding) -- the dense branch, to 1e-28 on the unrounded iterate.
* the one-constraint toy of test/runtests_solver.jl:30-51 (objective 1 to 1e-10) and its `model_psd_variables_as_free_variables` twin.

An interior-point iterate at iteration k is a smooth function of S^-1 at iterations < k: a wrong Schur complement moves mu, the
errors and the step lengths of iteration 2 at the first digit, so this trace pins the hot path, not only the loop around it.
"""
import os

import mpmath as mp
import numpy as np
import pytest

from tests.util import load_flat

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# docs/src/solving.md:39-44 -- iter: (mu, D-obj, P-obj, gap, D-error, d-error, p-error, alpha_d, alpha_p, beta) exactly as printed
REF_LOG = {
    1: ("1.000e+20", "0.000e+00", "0.000e+00", "0.00e+00", "1.00e+10", "1.00e+00", "1.95e+10", "7.42e-01", "7.10e-01", "3.00e-01"),
    2: ("3.995e+19", "1.999e+11", "-2.907e+09", "1.03e+00", "2.58e+09", "2.58e-01", "5.65e+09", "7.46e-01", "7.17e-01", "3.00e-01"),
    3: ("1.576e+19", "3.079e+11", "-4.779e+09", "1.03e+00", "6.53e+08", "6.53e-02", "1.60e+09", "7.32e-01", "7.31e-01", "3.00e-01"),
    55: ("5.066e-14", "-2.113e+00", "-2.113e+00", "8.39e-14", None, None, None, "1.00e+00", "1.00e+00", "1.00e-01"),
    56: ("5.067e-15", "-2.113e+00", "-2.113e+00", "8.39e-15", None, None, None, "1.00e+00", "1.00e+00", "1.00e-01"),
}
REF_ITERATIONS = 56                                                                                   # docs/src/solving.md:44-45
REF_DUAL = "-2.112913881423601867325289796075301826150007716044362101360781221096092533872562"       # :49
REF_PRIMAL = "-2.112913881423605414349991239275382883067580432169230529548206052006356176913883"    # :50
REF_GAP = "8.393680245626824434313082297089851809408852609517159688543365552836941907249006e-16"      # :51


def _printed_close(ours, printed):
    """equal to the printed digits: half a unit of the last printed digit + 1e-3 relative"""
    ref = float(printed)
    mant = printed.split("e")[0]
    digits = len(mant.split(".")[1])
    expo = int(printed.split("e")[1])
    half_ulp = 0.5 * 10.0 ** (expo - digits)
    return abs(ours - ref) <= half_ulp + 1e-3 * abs(ref)


def _check_log(hist, iterations, who):
    assert iterations == REF_ITERATIONS, (who, iterations)
    for it, row in REF_LOG.items():
        ours = hist[it - 1]
        assert int(ours[0]) == it
        for col, printed in enumerate(row, start=1):
            if printed is None:
                continue            # errors of the last iterations are rounding noise at 1e-77 (256 bits): not comparable
            assert _printed_close(float(ours[col]), printed), (who, "iteration", it, "column", col, float(ours[col]), printed)


def _limbs_to_mp(row):
    with mp.workprec(700):
        return mp.fsum(mp.mpf(float(v)) for v in row)


def _check_final_objectives(obj_limbs, who):
    with mp.workprec(700):
        d, p, gap = (_limbs_to_mp(obj_limbs[i]) for i in range(3))
        assert abs(d - mp.mpf(REF_DUAL)) <= mp.mpf("1e-14"), (who, mp.nstr(d, 30))
        assert abs(p - mp.mpf(REF_PRIMAL)) <= mp.mpf("1e-14"), (who, mp.nstr(p, 30))
        assert abs(gap - mp.mpf(REF_GAP)) <= mp.mpf("1e-3") * mp.mpf(REF_GAP), (who, mp.nstr(gap, 10))
        # the optimum lies between the two objectives the reference printed (weak duality; maximisation: dual >= primal)
        assert mp.mpf(REF_PRIMAL) - mp.mpf("1e-15") <= p <= d <= mp.mpf(REF_DUAL) + mp.mpf("1e-15") or abs(d - p) <= mp.mpf("1e-15")


@pytest.fixture(scope="module")
def min_f_2():
    return load_flat(os.path.join(GOLDEN, "min_f_2.npz"))


def test_min_f_generator_reproduces_the_committed_instance(min_f_2):
    """P = 11 constraints, blocks 4 x 4 (rank 1) and 3 x 3 (rank 2), one free variable; the generator picks the
    committed samples and data (to fp64 rounding of the QR steps)."""
    import clrs_amd
    from clrs_amd.problems import min_f
    f0, extra = min_f_2
    sdp = min_f(2)
    f = clrs_amd.flatten(sdp)
    assert list(f.cluster_P) == [11] and list(f.block_n) == [4, 3] and f.n_free == 1
    assert sorted(set(int(k) for k in f.term_rank[f.term_ptr[1]:f.term_ptr[2]])) == [0, 1]          # the rank-2 block
    samples = np.array([[float(t) for t in s] for s in sdp.names["samples"]])
    assert np.allclose(samples, extra["samples"], atol=1e-15)
    for name in ("term_vs", "term_lambda", "c", "B", "b"):
        assert np.allclose(getattr(f, name), getattr(f0, name), rtol=1e-9, atol=1e-12), name


def test_oracle_reproduces_the_reference_log_of_min_f_2(min_f_2, oracle_built):
    from oracle.oracle import Oracle
    f, _ = min_f_2
    r = Oracle(f, mp_bits=256).solvesdp()              # the reference's defaults: prec 256, omega 1e10, gap 1e-15, errors 1e-30
    assert r["error_code"] == 0 and r["pd_feas"]
    _check_log(r["hist"], r["iterations"], "oracle")
    _check_final_objectives(r["objectives_limbs"], "oracle")
    assert abs(r["p_obj"] + 2.113) <= 1e-2            # test/runtests_solver.jl:10-11


DENSE = [("theta_c5", lambda: mp.sqrt(5)), ("povm_2x2", lambda: mp.sqrt(2) / 4 + mp.mpf(1) / 2)]


@pytest.mark.parametrize("name,value", DENSE)
def test_oracle_reaches_the_1e30_answers_of_the_dense_examples(name, value, oracle_built):
    import clrs_amd
    from clrs_amd import problems
    from oracle.oracle import Oracle
    f = clrs_amd.flatten(getattr(problems, name)())
    assert np.all(f.block_kind == 1)
    r = Oracle(f, mp_bits=256).solvesdp(duality_gap_threshold=1e-30)
    assert r["error_code"] == 0 and r["gap"] < 1e-30
    with mp.workprec(700):
        for i in (0, 1):
            assert abs(_limbs_to_mp(r["objectives_limbs"][i]) - value()) <= mp.mpf("1e-28"), (name, i)


@pytest.mark.parametrize("name", ["toy_z", "toy_z_as_free"])
def test_oracle_solves_the_toy_problem(name, oracle_built):
    import clrs_amd
    from clrs_amd import problems
    from oracle.oracle import Oracle
    f = clrs_amd.flatten(getattr(problems, name)())
    r = Oracle(f, mp_bits=256).solvesdp()
    assert r["error_code"] == 0 and abs(r["p_obj"] - 1.0) <= 1e-10 and abs(r["d_obj"] - 1.0) <= 1e-10
    r = Oracle(f, mp_bits=256).solvesdp(need_primal_feasible=1)                 # test/runtests_solver.jl:43-44
    assert r["primal_error"] < 1e-30
    r = Oracle(f, mp_bits=256).solvesdp(need_dual_feasible=1)                   # :45-46
    assert r["dual_error"] < 1e-30


# ------------------------------------------------------------------------------------------------------------------------------------
# the HIP path (5 limbs = 262 bits: the reference's default precision)
# ------------------------------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_hip_path_reproduces_the_reference_log_of_min_f_2(min_f_2, oracle_built):
    from clrs_amd.mw import solvesdp_mw
    f, extra = min_f_2
    r = solvesdp_mw(f, prec=256)
    assert r.error_code == 0 and r.status == "Optimal", (r.status, r.error_code)
    _check_log(r.history, r.iterations, "hip f64x5")
    _check_final_objectives(r.timings["objectives_limbs"], "hip f64x5")
    # and against the 256-bit oracle's own history on the same instance, far inside the printed digits
    oh = extra["oracle_hist"]
    assert r.history.shape == oh.shape
    for col in (1, 2, 3, 8, 9):
        assert np.allclose(r.history[:, col], oh[:, col], rtol=1e-6, atol=1e-12), col


@pytest.mark.gpu
@pytest.mark.parametrize("name,value", DENSE)
def test_hip_path_reaches_the_1e30_answers_of_the_dense_examples(name, value):
    import clrs_amd
    from clrs_amd import problems
    from clrs_amd.mw import solvesdp_mw
    f = clrs_amd.flatten(getattr(problems, name)())
    r = solvesdp_mw(f, prec=256, duality_gap_threshold=1e-30)
    assert r.error_code == 0 and r.status == "Optimal" and r.duality_gap < 1e-30, (r.status, r.error_code, r.duality_gap)
    with mp.workprec(700):
        for i in (0, 1):
            assert abs(_limbs_to_mp(r.timings["objectives_limbs"][i]) - value()) <= mp.mpf("1e-28"), (name, i)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["toy_z", "toy_z_as_free"])
def test_hip_path_solves_the_toy_problem(name):
    import clrs_amd
    from clrs_amd import problems
    from clrs_amd.mw import solvesdp_mw
    f = clrs_amd.flatten(getattr(problems, name)())
    r = solvesdp_mw(f, prec=256)
    assert r.error_code == 0 and abs(r.primal_objective - 1.0) <= 1e-10 and abs(r.dual_objective - 1.0) <= 1e-10
    r = solvesdp_mw(f, prec=256, need_primal_feasible=True)
    assert r.primal_error < 1e-30
    r = solvesdp_mw(f, prec=256, need_dual_feasible=True)
    assert r.dual_error < 1e-30


# ------------------------------------------------------------------------------------------------------------------------------------
# warm start (dualsol / primalsol, src/solver.jl:202-239; the reference's test: test/runtests_solver.jl:166-173)
# ------------------------------------------------------------------------------------------------------------------------------------

def test_oracle_warm_start_continues_from_the_given_iterate(oracle_built):
    import clrs_amd
    from clrs_amd import problems
    from oracle.oracle import Oracle
    f = clrs_amd.flatten(problems.toy_z())
    o = Oracle(f, mp_bits=256)
    r1 = o.solvesdp(duality_gap_threshold=1e-5)
    cold = o.solvesdp(duality_gap_threshold=1e-10)
    r2 = o.solvesdp(duality_gap_threshold=1e-10, start=(r1["x"], r1["y"], r1["X"], r1["Y"]))
    assert r1["error_code"] == 0 and r2["error_code"] == 0 and r2["gap"] < 1e-10
    assert abs(r1["p_obj"] - r2["p_obj"]) <= 1e-4                                 # test/runtests_solver.jl:172
    assert 0 < r2["iterations"] < cold["iterations"] - r1["iterations"] + 4        # it continues, it does not start over
    again = o.solvesdp(duality_gap_threshold=1e-10)                                 # the start is used once
    assert again["iterations"] == cold["iterations"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["toy_z", "ce_8_3"])
def test_hip_path_warm_start_matches_the_oracle_started_from_the_same_iterate(name, oracle_built):
    """solve to gap 1e-5, restart from the returned iterate (clrs_mw_ipm_set) to 1e-10: the reference's warm-start test on its own toy, and on a
    sampled sphere-packing instance with free variables; the 320-bit oracle loop started from the SAME K-limb iterate takes the same iterations to
    the same objectives, step lengths and mu."""
    import clrs_amd
    from clrs_amd import problems
    from clrs_amd.mw import solvesdp_mw
    from oracle.oracle import Oracle
    from tests.util import flat
    f = clrs_amd.flatten(problems.toy_z()) if name == "toy_z" else flat(name)
    r1 = solvesdp_mw(f, limbs=5, duality_gap_threshold=1e-5)
    assert r1.error_code == 0 and r1.duality_gap < 1e-5
    r2 = solvesdp_mw(f, limbs=5, dualsol=r1, primalsol=r1, duality_gap_threshold=1e-10)
    assert r2.error_code == 0 and r2.status in ("Optimal", "NearOptimal") and r2.duality_gap < 1e-10, (r2.status, r2.duality_gap)
    assert abs(r1.primal_objective - r2.primal_objective) <= 1e-4                 # test/runtests_solver.jl:172
    cold = solvesdp_mw(f, limbs=5, duality_gap_threshold=1e-10)
    assert 0 < r2.iterations < cold.iterations
    y1 = r1.y if f.n_free else np.zeros((5, 1))
    ro = Oracle(f, mp_bits=320).solvesdp(duality_gap_threshold=1e-10, start=(r1.x, y1, r1.X, r1.Y))
    assert ro["error_code"] == 0 and ro["iterations"] == r2.iterations, (ro["iterations"], r2.iterations)
    assert abs(ro["p_obj"] - r2.primal_objective) <= 1e-12 * max(1.0, abs(ro["p_obj"])) and abs(ro["d_obj"] - r2.dual_objective) <= 1e-12 * max(1.0, abs(ro["d_obj"]))
    n = r2.iterations
    for col in (1, 2, 3, 8, 9):                                                    # mu, objectives of the iterate at the top of each iteration, step lengths
        assert np.allclose(r2.history[:n, col], ro["hist"][:n, col], rtol=1e-6, atol=1e-12), (col, r2.history[:n, col], ro["hist"][:n, col])
    # only one of the two given: ignored, as in the reference (`if !isnothing(dualsol) && !isnothing(primalsol)`, src/solver.jl:202)
    half = solvesdp_mw(f, limbs=5, dualsol=r1, duality_gap_threshold=1e-10)
    assert half.iterations == cold.iterations
