"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerances (fp64 path): S assembly is compared in the norm-wise sense |S_hip - S_ref| <= 1e-11 * max|S_ref|
against the __float128 oracle (the fp64 oracle itself differs from quad by ~1e-13 on these instances);
factor/solve outputs are compared through residuals and against the fp64 oracle where the problem is
well conditioned."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests.util import chol_blocks_np, flat, spd_iterates  # noqa: E402


def _lib():
    from clrs_amd import _lib
    return _lib.load()


def _dp(a):
    from clrs_amd import _lib
    return a.ctypes.data_as(_lib.p_d)


@pytest.mark.parametrize("ta,tb", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (16, 16, 4), (64, 64, 16), (37, 53, 29), (130, 70, 100), (3, 200, 65),
                                   (256, 256, 16), (300, 260, 70), (513, 257, 33),
                                   # two rounds of 128 x 128 tiles against four of 64 x 64: these take the large tile (gemm_prefers_large_tiles),
                                   # with a ragged last tile column / row and a partial last chunk of k
                                   (4096, 4000, 40), (4000, 4096, 19)])
def test_gemm_kernel(ta, tb, M, N, K):
    rng = np.random.default_rng(M * 1000 + N * 10 + K)
    A = rng.standard_normal((K, M) if ta else (M, K))
    B = rng.standard_normal((N, K) if tb else (K, N))
    C = rng.standard_normal((M, N))
    ref = 0.7 * (A.T if ta else A) @ (B.T if tb else B) - 0.3 * C
    Af, Bf, Cf = np.asfortranarray(A), np.asfortranarray(B), np.asfortranarray(C)
    rc = _lib().clrs_test_gemm(0, ta, tb, M, N, K, 0.7, _dp(Af), Af.shape[0], _dp(Bf), Bf.shape[0], -0.3, _dp(Cf), M)
    assert rc == 0
    assert np.max(np.abs(Cf - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref)))


@pytest.mark.parametrize("which", ["A", "B"])
def test_gemm_kernel_with_a_leading_dimension_beyond_32_bit_offsets(which):
    """The staged GEMM addresses a tile's operands by 32-bit byte offsets (64 or 128 leading dimensions of reach); the Gram product of a
    dense block with n >= 2048 has lda = ldb = n^2 >= 2^22.  Products beyond the reach of the large tile take the small one, beyond that
    the 64-bit-offset instantiation -- instead of failing the whole context (round-2 behaviour).  Here ld = 2^23 + 16 on the operand that
    is walked along the tile (A transposed: K x M; B as stored: K x N), 64 ld doubles = 4.3 GB of reach needed."""
    ld = (1 << 23) + 16
    M, N, K = 5, 7, 100
    rng = np.random.default_rng(11 + (which == "B"))
    ta = 1 if which == "A" else 0
    if which == "A":
        Af = np.zeros((ld, M), order="F"); Af[:K, :] = rng.standard_normal((K, M))
        opA = Af[:K, :].T
        Bf = np.asfortranarray(rng.standard_normal((K, N))); opB = Bf
    else:
        Af = np.asfortranarray(rng.standard_normal((M, K))); opA = Af
        Bf = np.zeros((ld, N), order="F"); Bf[:K, :] = rng.standard_normal((K, N))
        opB = Bf[:K, :]
    C = rng.standard_normal((M, N))
    ref = 0.7 * opA @ opB - 0.3 * C
    Cf = np.asfortranarray(C)
    rc = _lib().clrs_test_gemm(0, ta, 0, M, N, K, 0.7, _dp(Af), Af.shape[0], _dp(Bf), Bf.shape[0], -0.3, _dp(Cf), M)
    assert rc == 0
    assert np.max(np.abs(Cf - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref)))


@pytest.mark.parametrize("n", [1, 2, 17, 64, 65, 150, 300, 513, 600, 1100])
def test_potrf_and_trsm(n):
    """n > 512: the triangular solves run through inverted 512 x 512 diagonal blocks (plan_trsm_blockinv), with ragged last blocks."""
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n, n))
    A = np.asfortranarray(np.eye(n) + G @ G.T / n)
    L = A.copy(order="F")
    assert _lib().clrs_test_potrf(0, n, _dp(L), n) == 0
    Lr = np.linalg.cholesky(A)
    assert np.max(np.abs(np.tril(L) - Lr)) <= 1e-12 * np.max(np.abs(Lr))
    Lf = np.asfortranarray(np.tril(L))
    for trans in (0, 1):
        for nrhs in (1, 70):
            B = np.asfortranarray(rng.standard_normal((n, nrhs)))
            Bs = B.copy(order="F")
            assert _lib().clrs_test_trsm(0, trans, n, nrhs, _dp(Lf), n, _dp(Bs), n) == 0
            ref = np.linalg.solve(Lf.T if trans else Lf, B)
            assert np.max(np.abs(Bs - ref)) <= 1e-10 * max(1.0, np.max(np.abs(ref)))


def test_stream_probe_measures_a_plausible_rate():
    """clrs_test_stream (the copy roof bench.py quotes beside the assembly kernel): runs, rejects bad sizes, and reports a rate
    below 12 TB/s (and not absurdly low) for a footprint of 96 MB."""
    import ctypes as C
    us = C.c_double(0.0)
    # (this can be the first GPU work of the process on a box that has been idle: once a launch of the probe was seen to take 15 ms
    # there -- the first call is a warm-up, and the lower bound only rules out nonsense)
    assert _lib().clrs_test_stream(0, 64 << 20, 32 << 20, 20, C.byref(us)) == 0
    assert _lib().clrs_test_stream(0, 64 << 20, 32 << 20, 20, C.byref(us)) == 0
    rate = (96 << 20) / (us.value * 1e-6) / 1e12
    assert 0.05 < rate < 12.0, (us.value, rate)
    assert _lib().clrs_test_stream(0, 0, 0, 1, C.byref(us)) < 0


def test_host_pointer_entry_points_staged_and_direct_copies_agree():
    """The host-pointer entry points stage the caller's buffers through a pinned arena (from the second call of a context on: the
    arena is sized by what the first call asked for); "pin_limit" = 0 keeps the direct copies.  Same bits either way, call after call."""
    from clrs_amd.solver import SchurContext, compute_T_decomposition, solve_system
    f = flat("ce_8_3")
    X, Y = spd_iterates(f, seed=9)
    Xc = chol_blocks_np(f, X)
    rng = np.random.default_rng(10)
    rx, ry = rng.standard_normal(f.x_len), rng.standard_normal(f.n_free)
    results = []
    for limit in (None, 0):
        if limit is not None:
            assert _lib().clrs_config_set(b"pin_limit", limit) == 0
        try:
            ctx = SchurContext(f)
            out = []
            for _ in range(3):
                _, S, AY = compute_T_decomposition(ctx, Xc, Y, want_S=True)
                dx, dy = solve_system(ctx, rx, ry)
                out.append((S.copy(), np.array(AY, copy=True), dx.copy(), dy.copy()))
            ctx.close()
        finally:
            assert _lib().clrs_config_set(b"pin_limit", 64 << 20) == 0
        for o in out[1:]:
            assert all(np.array_equal(a, b) for a, b in zip(o, out[0]))
        results.append(out[0])
    assert all(np.array_equal(a, b) for a, b in zip(results[0], results[1]))


def test_potrf_reports_failure():
    A = np.asfortranarray(np.array([[1.0, 2.0], [2.0, 1.0]]))
    assert _lib().clrs_test_potrf(0, 2, _dp(A), 2) == 1
    # beyond one block (k_chol_level): a pivot that turns negative in the third block column
    n = 200
    B = np.asfortranarray(np.eye(n) * 4.0)
    B[150, 150] = -1.0
    assert _lib().clrs_test_potrf(0, n, _dp(B), n) == 1


def test_several_large_matrices_per_level_launch(oracle_built):
    """Three clusters whose X blocks (101 x 101) and Schur blocks (201 x 201) are all beyond one 64-wide block: the per-column launches
    of k_chol_level then carry several matrices (work tables instead of the single job in the kernel arguments)."""
    from clrs_amd.sdp import replicate_clusters
    from clrs_amd.solver import SchurContext, compute_T_decomposition
    from oracle.oracle import Oracle
    f = flat("polyopt_scaled_100")
    big = replicate_clusters(f, 3)
    X, Y = spd_iterates(big, seed=5)
    ctx = SchurContext(big, fused=False)
    Xc = ctx.cholesky_blocks(X)
    assert np.max(np.abs(Xc - chol_blocks_np(big, X))) <= 1e-12 * np.max(np.abs(Xc))
    _, S, _ = compute_T_decomposition(ctx, Xc, Y, want_S=True)
    L, LinvB, LQ = ctx.get_factor()
    # the factor stage of the three clusters is one augmented factorisation each, their shares of Q summed: against the oracle on the
    # replicated problem, with a solve
    ob = Oracle(big, quad=False)
    ob.schur_assemble(Xc, Y)
    assert ob.schur_factor() == 0
    _, LinvB_ref, LQ_ref = ob.get_factor()
    assert np.max(np.abs(LinvB - LinvB_ref)) <= 1e-9 * max(1.0, np.max(np.abs(LinvB_ref)))
    assert np.max(np.abs(LQ - LQ_ref)) <= 1e-9 * max(1.0, np.max(np.abs(LQ_ref)))
    rng = np.random.default_rng(9)
    rx, ry = rng.standard_normal(big.x_len), rng.standard_normal(big.n_free)
    from clrs_amd.solver import solve_system
    dx, dy = solve_system(ctx, rx, ry)
    dx_ref, dy_ref = ob.schur_solve(rx, ry)
    assert np.max(np.abs(dx - dx_ref)) <= 1e-8 * max(1.0, np.max(np.abs(dx_ref)))
    assert np.max(np.abs(dy - dy_ref)) <= 1e-8 * max(1.0, np.max(np.abs(dy_ref)))
    ctx.close()
    nxy, nS = f.xy_len, f.S_len
    P = int(f.cluster_P[0])
    for k in range(3):
        o = Oracle(f, quad=False)
        Sk, _ = o.schur_assemble(Xc[k * nxy:(k + 1) * nxy], Y[k * nxy:(k + 1) * nxy])
        assert np.max(np.abs(S[k * nS:(k + 1) * nS] - Sk)) <= 1e-10 * np.max(np.abs(Sk))
        Lk = np.linalg.cholesky(Sk.reshape(P, P, order="F"))
        got = np.tril(L[k * nS:(k + 1) * nS].reshape(P, P, order="F"))
        assert np.max(np.abs(got - Lk)) <= 1e-9 * np.max(np.abs(Lk))


ASSEMBLE_CASES = ["x2p1", "polyopt8", "polyopt40", "delsarte_3_10", "delsarte_8_3", "ce_8_15", "ce_8_3", "ns_8_3_2", "ns_8_15_2",
                  "threepoint_4", "sdpa_small", "sdpa_mid", "polyopt_scaled_100", "polyopt_scaled_300"]


PATHS = {"wave3": dict(fused=True, wave=True, wave2=True, wave3=True), "wave2": dict(fused=True, wave=True, wave2=True, wave3=False, wave4=False, wave5=False),
         "wave": dict(fused=True, wave=True, wave2=False, wave4=False, wave5=False), "fused": dict(fused=True, wave=False), "staged": dict(fused=False)}
WAVE5_CASES = {"ns_8_15_2": 1}            # clusters taken by k_cluster_assemble_w5 (2 x 2 blocks of 16 x 16 sub-blocks on shared sample vectors) on the default path
WAVE4_CASES = {"polyopt40": 1}            # clusters taken by k_cluster_assemble_w4 (simple blocks of 17-32 rows or 33-64 constraints) on the default path
WAVE2_CASES = {"x2p1": 1, "polyopt8": 1, "delsarte_8_3": 0, "ce_8_15": 2, "ce_8_3": 2}    # clusters taken by k_cluster_assemble_w2
WAVE_CASES = {"x2p1", "polyopt8", "delsarte_3_10", "delsarte_8_3", "ce_8_15", "ce_8_3", "sdpa_small"}   # every cluster takes k_cluster_assemble_w1


@pytest.mark.parametrize("path", list(PATHS))
@pytest.mark.parametrize("name", ASSEMBLE_CASES)
def test_schur_assemble_matches_oracle(name, path, oracle_built):
    from clrs_amd.solver import SchurContext
    from oracle.oracle import Oracle
    f = flat(name)
    X, Y = spd_iterates(f, seed=1)
    Xc = chol_blocks_np(f, X)
    ctx = SchurContext(f, **PATHS[path])
    fused = path != "staged"
    assert (ctx.fused_clusters() > 0) == (fused and name not in ("sdpa_mid", "polyopt_scaled_100", "polyopt_scaled_300"))
    if path in ("wave", "wave2", "wave3") and name in WAVE_CASES:
        assert ctx.wave_clusters() == f.n_clusters
    if path in ("wave2", "wave3") and name in WAVE2_CASES:
        assert ctx.wave2_clusters() == WAVE2_CASES[name]
    assert ctx.wave4_clusters() == (WAVE4_CASES.get(name, 0) if path == "wave3" else 0)
    assert ctx.wave5_clusters() == (WAVE5_CASES.get(name, 0) if path == "wave3" else 0)
    if path == "wave":
        assert ctx.wave2_clusters() == 0
    if path in ("fused", "staged"):
        assert ctx.wave_clusters() == 0
    S, AY = ctx.compute_S_integrated(Xc, Y)
    o = Oracle(f, quad=True, use_lo=False)
    S_ref, AY_ref = o.schur_assemble(Xc, Y)
    scale = np.max(np.abs(S_ref))
    assert np.max(np.abs(S - S_ref)) <= 1e-11 * scale
    if f.n_terms:
        assert np.max(np.abs(AY - AY_ref)) <= 1e-11 * max(1.0, np.max(np.abs(AY_ref)))
    # exact symmetry, like symmetric! (src/tools.jl:43-57)
    for j in range(f.n_clusters):
        P = int(f.cluster_P[j])
        Sj = S[f.S_off[j]:f.S_off[j + 1]].reshape(P, P, order="F")
        assert np.array_equal(Sj, Sj.T)
    ctx.close()


@pytest.mark.parametrize("name", ["threepoint_4", "ns_8_15_2", "ns_8_3_2", "sdpa_small", "delsarte_8_3", "x2p1"])
def test_block_split_assembly_is_bit_identical(name):
    """General fused assembly with one workgroup per PSD block + k_sum_S_slabs (the default for few clusters) against one workgroup
    per cluster: the same additions in the same order, so S and A_Y agree to the last bit."""
    from clrs_amd.solver import SchurContext
    f = flat(name)
    X, Y = spd_iterates(f, seed=5)
    Xc = chol_blocks_np(f, X)
    out = []
    for split in (True, False):
        ctx = SchurContext(f, wave=False, split_blocks=split)
        S, AY = ctx.compute_S_integrated(Xc, Y)
        out.append((S.copy(), np.array(AY, copy=True)))
        ctx.close()
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize("name,copies", [("ce_8_15", 1100), ("ce_8_3", 1100), ("polyopt8", 2500)])
def test_cluster_per_wave_assembly_many_clusters(name, copies, oracle_built):
    """k_cluster_assemble_w3 with several clusters per wave (contiguous cluster ranges, loads of the next cluster's first block in
    flight across the cluster boundary): more clusters than resident waves, every cluster checked against the oracle."""
    import torch
    torch.cuda.set_device(0)       # torch's HIP runtime first (as in bench.py); the library then shares the device with it
    from clrs_amd.sdp import replicate_clusters
    from clrs_amd.solver import SchurContext
    from oracle.oracle import Oracle
    f = flat(name)
    big = replicate_clusters(f, copies)
    X, Y = spd_iterates(big, seed=7)
    Xc = chol_blocks_np(big, X)
    ctx = SchurContext(big)
    assert ctx.wave2_clusters() == big.n_clusters
    S, AY = ctx.compute_S_integrated(Xc, Y)
    # the device-pointer entry reads the caller's buffers directly: same result from other iterates held in torch tensors
    from clrs_amd.sharded import _DevArray
    X2, Y2 = spd_iterates(big, seed=8)
    Xc2 = chol_blocks_np(big, X2)
    tX, tY = torch.from_numpy(Xc2).to("cuda:0"), torch.from_numpy(Y2).to("cuda:0")
    torch.cuda.synchronize()
    ctx.assemble_dev(tX.data_ptr(), tY.data_ptr())
    torch.cuda.synchronize()
    S2 = torch.as_tensor(_DevArray(ctx.S_buffer(), big.S_len), device="cuda:0").cpu().numpy()
    ctx.close()
    o = Oracle(f, quad=False)
    nxy, nS, nT = f.xy_len, f.S_len, f.n_terms
    for k in (0, copies // 2, copies - 1):
        Sk, _ = o.schur_assemble(Xc2[k * nxy:(k + 1) * nxy], Y2[k * nxy:(k + 1) * nxy])
        assert np.max(np.abs(S2[k * nS:(k + 1) * nS] - Sk)) <= 1e-11 * np.max(np.abs(Sk))
    worst = 0.0
    for k in range(copies):
        Sk, AYk = o.schur_assemble(Xc[k * nxy:(k + 1) * nxy], Y[k * nxy:(k + 1) * nxy])
        worst = max(worst, np.max(np.abs(S[k * nS:(k + 1) * nS] - Sk)) / np.max(np.abs(Sk)))
        if nT:
            worst = max(worst, np.max(np.abs(AY[k * nT:(k + 1) * nT] - AYk)) / max(1.0, np.max(np.abs(AYk))))
    assert worst <= 1e-11


@pytest.mark.parametrize("copies", [1, 700])
@pytest.mark.parametrize("name", ["ce_8_15", "ce_8_3"])
def test_cluster_per_wave_assembly_rare_paths(name, copies, oracle_built):
    """The rarely taken paths of k_cluster_assemble_w3: constraints of a cluster in a permuted order (S_j stored through the
    vector -> constraint table), two 1 x 1 dense blocks in one cluster (the second one is fetched at the cluster's end), three
    low-rank blocks in one cluster -- alone and with several clusters per wave (the loads of the next block are in flight across
    these paths)."""
    from clrs_amd.sdp import replicate_clusters
    from clrs_amd.solver import SchurContext
    from oracle.oracle import Oracle
    from tests.util import duplicate_block, permute_cluster_constraints
    f = flat(name)
    dense = [b for b in range(f.n_blocks) if f.block_kind[b] == 1]
    assert len(dense) == 1
    g = duplicate_block(f, dense[0], 0.5)                 # second 1 x 1 dense block in that cluster
    g = duplicate_block(g, 0, -0.75)                      # third low-rank block in cluster 0 (negative lambdas too)
    g = permute_cluster_constraints(g, seed=11)
    big = replicate_clusters(g, copies) if copies > 1 else g
    X, Y = spd_iterates(big, seed=9)
    Xc = chol_blocks_np(big, X)
    ctx = SchurContext(big, wave2=True)
    assert ctx.wave2_clusters() == big.n_clusters
    S, AY = ctx.compute_S_integrated(Xc, Y)
    ctx.close()
    o = Oracle(g, quad=False)
    nxy, nS, nT = g.xy_len, g.S_len, g.n_terms
    for k in range(copies):
        Sk, AYk = o.schur_assemble(Xc[k * nxy:(k + 1) * nxy], Y[k * nxy:(k + 1) * nxy])
        assert np.max(np.abs(S[k * nS:(k + 1) * nS] - Sk)) <= 1e-11 * np.max(np.abs(Sk))
        assert np.max(np.abs(AY[k * nT:(k + 1) * nT] - AYk)) <= 1e-11 * max(1.0, np.max(np.abs(AYk)))
        Sj = S[k * nS:k * nS + int(g.cluster_P[0]) ** 2].reshape(int(g.cluster_P[0]), -1, order="F")
        assert np.array_equal(Sj, Sj.T)


@pytest.mark.parametrize("seed", range(8))
def test_random_simple_structures(seed, oracle_built):
    """Randomised structure for the general (partial-tile) paths of the wave kernels: clusters of different sizes P <= 32 in one
    context, 1-3 low-rank blocks of different sides n <= 16 per cluster, 0-2 dense 1 x 1 blocks touching a subset of the constraints,
    placed anywhere among them; a few free variables.  Assembly against the oracle on every wave path, then factor + solve."""
    import clrs_amd
    from clrs_amd.solver import SchurContext, compute_T_decomposition, solve_system
    from oracle.oracle import Oracle
    from tests.util import random_simple_sdp
    f = clrs_amd.flatten(random_simple_sdp(seed, J=3 + seed % 3, n_free=seed % 4, definite=True))
    X, Y = spd_iterates(f, seed=seed + 100)
    Xc = chol_blocks_np(f, X)
    o = Oracle(f, quad=False)
    S_ref, AY_ref = o.schur_assemble(Xc, Y)
    for kw in (dict(wave2=True, wave3=True), dict(wave2=True, wave3=False), dict(wave2=False), dict(wave=False)):
        ctx = SchurContext(f, **kw)
        if "wave3" in kw:
            assert ctx.wave2_clusters() == f.n_clusters
        S, AY = ctx.compute_S_integrated(Xc, Y)
        for j in range(f.n_clusters):
            sl = slice(int(f.S_off[j]), int(f.S_off[j + 1]))
            assert np.max(np.abs(S[sl] - S_ref[sl])) <= 1e-11 * np.max(np.abs(S_ref[sl])), (kw, j)
        assert np.max(np.abs(AY - AY_ref)) <= 1e-11 * max(1.0, np.max(np.abs(AY_ref))), kw
        ctx.close()
    assert o.schur_factor() == 0
    rng = np.random.default_rng(seed)
    rx, ry = rng.standard_normal(f.x_len), rng.standard_normal(f.n_free)
    dx_ref, dy_ref = o.schur_solve(rx, ry)
    for kw in (dict(), dict(solve_small2=False), dict(factor_small=2)):
        ctx = SchurContext(f, **kw)
        compute_T_decomposition(ctx, Xc, Y)
        dx, dy = solve_system(ctx, rx, ry)
        assert np.max(np.abs(dx - dx_ref)) <= 1e-7 * max(1.0, np.max(np.abs(dx_ref))), kw
        if f.n_free:
            assert np.max(np.abs(dy - dy_ref)) <= 1e-7 * max(1.0, np.max(np.abs(dy_ref))), kw
        ctx.close()


@pytest.mark.parametrize("P,N", [(127, 127), (128, 128), (129, 7), (80, 129), (128, 0)])
def test_size_limits_of_the_fused_regime(P, N, oracle_built):
    """Clusters and free-variable counts at the limits where the plan switches between the LDS-resident kernels (P, N <= 128) and the
    staged ones: two clusters of exactly P constraints (four low-rank blocks of side 16 each, so S_j is definite), N free variables.
    Assembly, factorisation and solve against the oracle."""
    import clrs_amd
    from clrs_amd.solver import SchurContext, compute_T_decomposition, solve_system
    from oracle.oracle import Oracle
    from tests.util import random_simple_sdp
    f = clrs_amd.flatten(random_simple_sdp(1000 + P + N, J=2, n_free=N, fixed_P=P, max_n=16, lr_blocks=4))
    X, Y = spd_iterates(f, seed=P + 2 * N)
    Xc = chol_blocks_np(f, X)
    o = Oracle(f, quad=False)
    S_ref, AY_ref = o.schur_assemble(Xc, Y)
    assert o.schur_factor() == 0
    rng = np.random.default_rng(P * 131 + N)
    rx, ry = rng.standard_normal(f.x_len), rng.standard_normal(f.n_free)
    dx_ref, dy_ref = o.schur_solve(rx, ry)
    ctx = SchurContext(f)
    _, S, AY = compute_T_decomposition(ctx, Xc, Y, want_S=True)
    assert np.max(np.abs(S - S_ref)) <= 1e-11 * np.max(np.abs(S_ref))
    assert np.max(np.abs(AY - AY_ref)) <= 1e-11 * max(1.0, np.max(np.abs(AY_ref)))
    dx, dy = solve_system(ctx, rx, ry)
    assert np.max(np.abs(dx - dx_ref)) <= 1e-7 * max(1.0, np.max(np.abs(dx_ref)))
    if N:
        assert np.max(np.abs(dy - dy_ref)) <= 1e-7 * max(1.0, np.max(np.abs(dy_ref)))
    ctx.close()


def test_dedup_counts_match_oracle(oracle_built):
    from clrs_amd.solver import SchurContext
    from oracle.oracle import Oracle
    f = flat("ns_8_15_2")
    ctx = SchurContext(f)
    o = Oracle(f)
    for b in range(f.n_blocks):
        if f.block_kind[b] == 0:
            UR, UL = o.unique_counts(b)
            assert ctx.unique_counts(b) == (list(UR), list(UL))
    # the 96-constraint cluster collapses to 32 unique vectors per sub-block row (src/solver.jl:988)
    big = [b for b in range(f.n_blocks) if f.block_n[b] == 32]
    assert big and all(ctx.unique_counts(b)[0] == [32, 32] for b in big)
    ctx.close()


FACTOR_CASES = ["x2p1", "polyopt8", "polyopt40", "delsarte_3_10", "delsarte_8_3", "ns_8_3_2", "threepoint_4", "sdpa_small", "sdpa_mid",
                "polyopt_scaled_100", "polyopt_scaled_300"]


@pytest.mark.parametrize("fused", [True, "k_solve_small", "k_factor_small_waves", False],
                         ids=["fused", "fused_k_solve_small", "fused_k_factor_small_wave_per_cluster", "staged"])
@pytest.mark.parametrize("name", FACTOR_CASES)
def test_factor_and_solve_match_oracle(name, fused, oracle_built):
    from clrs_amd.solver import SchurContext, compute_T_decomposition, solve_system
    from oracle.oracle import Oracle
    f = flat(name)
    X, Y = spd_iterates(f, seed=2)
    Xc = chol_blocks_np(f, X)
    ctx = SchurContext(f, fused=bool(fused), solve_small2=(False if fused == "k_solve_small" else None),
                       factor_small=(2 if fused == "k_factor_small_waves" else None))
    _, S, _ = compute_T_decomposition(ctx, Xc, Y, want_S=True)
    L, LinvB, LQ = ctx.get_factor()
    o = Oracle(f, quad=False)
    o.schur_assemble(Xc, Y)
    assert o.schur_factor() == 0
    L_ref, LinvB_ref, LQ_ref = o.get_factor()
    amp = 1e3 if name == "threepoint_4" else 1.0          # cond(S) ~ 1e10 at these iterates for the three-point instance
    tol = 1e-9 * amp
    assert np.max(np.abs(L - L_ref)) <= tol * np.max(np.abs(L_ref))
    if f.n_free:
        assert np.max(np.abs(LinvB - LinvB_ref)) <= tol * max(1.0, np.max(np.abs(LinvB_ref)))
        assert np.max(np.abs(LQ - LQ_ref)) <= tol * max(1.0, np.max(np.abs(LQ_ref)))
    rng = np.random.default_rng(7)
    rx, ry = rng.standard_normal(f.x_len), rng.standard_normal(f.n_free)
    dx, dy = solve_system(ctx, rx, ry)
    dx_ref, dy_ref = o.schur_solve(rx, ry)
    assert np.max(np.abs(dx - dx_ref)) <= 1e-8 * amp * max(1.0, np.max(np.abs(dx_ref)))
    if f.n_free:
        assert np.max(np.abs(dy - dy_ref)) <= 1e-8 * amp * max(1.0, np.max(np.abs(dy_ref)))
    # structural identity: S dx - B dy = rhs_x ; B^T dx = rhs_y   (src/solver.jl:1527)
    N = f.n_free
    bty = np.zeros(N)
    for j in range(f.n_clusters):
        P = int(f.cluster_P[j])
        sl = slice(int(f.cluster_off[j]), int(f.cluster_off[j + 1]))
        Sj = S[f.S_off[j]:f.S_off[j + 1]].reshape(P, P, order="F")
        r = Sj @ dx[sl] - rx[sl]
        if N:
            Bj = f.B[int(f.cluster_off[j]) * N:int(f.cluster_off[j + 1]) * N].reshape(P, N, order="F")
            r -= Bj @ dy
            bty += Bj.T @ dx[sl]
        assert np.max(np.abs(r)) <= 1e-8 * amp * max(1.0, np.max(np.abs(Sj)) * np.max(np.abs(dx)))
    if N:
        assert np.max(np.abs(bty - ry)) <= 1e-8 * max(1.0, np.max(np.abs(dx)))
    ctx.close()


def test_factor_failure_is_reported_like_the_reference():
    """cohnelkies(8,15) is not factorisable in fp64 (cond(S) ~ 1e30): the path must report it, not crash."""
    from clrs_amd.solver import SchurContext, SolverFailure, compute_T_decomposition
    f = flat("ce_8_15")
    X, Y = spd_iterates(f, seed=3)
    ctx = SchurContext(f)
    with pytest.raises(SolverFailure, match="was not decomposed"):
        compute_T_decomposition(ctx, chol_blocks_np(f, X), Y)
    ctx.close()


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "staged"])
def test_cholesky_blocks(fused, oracle_built):
    from clrs_amd.solver import SchurContext, SolverFailure
    f = flat("delsarte_3_10")
    X, _ = spd_iterates(f, seed=4)
    ctx = SchurContext(f, fused=fused)
    Xc = ctx.cholesky_blocks(X)
    assert np.max(np.abs(Xc - chol_blocks_np(f, X))) <= 1e-12 * np.max(np.abs(Xc))
    X[f.block_off[3]] = -1.0
    with pytest.raises(SolverFailure, match=r"block \(1,4\)"):
        ctx.cholesky_blocks(X)
    ctx.close()


@pytest.mark.parametrize("name", ["ns_8_3_2", "polyopt_scaled_300"])
def test_graph_mode_matches_eager(name):
    """polyopt_scaled_300: the staged plans (one launch per block column of the factorisations, the factor stage as one factorisation,
    substitutions through inverted diagonal blocks) replayed as hipGraphs."""
    from clrs_amd.solver import SchurContext
    f = flat(name)
    X, Y = spd_iterates(f, seed=5)
    Xc = chol_blocks_np(f, X)
    a = SchurContext(f)
    b = SchurContext(f, graph=True)
    Sa, _ = a.compute_S_integrated(Xc, Y)
    for _ in range(3):
        Sb, _ = b.compute_S_integrated(Xc, Y)
    assert np.array_equal(Sa, Sb)
    assert a.factor() == 0 and b.factor() == 0
    rx, ry = np.ones(f.x_len), np.ones(f.n_free)
    for u, v in zip(a.solve(rx, ry), b.solve(rx, ry)):
        assert np.array_equal(u, v)
    a.close(); b.close()


@pytest.mark.parametrize("name,expected,tol", [("x2p1", 1.0, 1e-6), ("delsarte_3_10", 13.158314, 1e-5), ("delsarte_8_3", 240.0, 1e-4)])
def test_solvesdp_known_answers(name, expected, tol):
    """End-to-end through the path with the reference's pinned objectives (test/runtests_solver.jl:15,86-87; README.md:149)."""
    from clrs_amd.solver import solvesdp
    r = solvesdp(flat(name))
    assert r.error_code == 0, (r.status, r.iterations)
    assert abs(r.primal_objective - expected) <= tol * max(1.0, abs(expected))
    assert abs(r.dual_objective - expected) <= tol * max(1.0, abs(expected))


def test_solvesdp_matches_oracle_loop(oracle_built):
    """Same fp64 algorithm, host+HIP vs the C oracle loop: objectives and iteration counts agree."""
    from clrs_amd.solver import solvesdp
    from oracle.oracle import Oracle
    f = flat("polyopt40")
    r = solvesdp(f)
    o = Oracle(f, quad=False).solvesdp(omega_p=1e4, omega_d=1e4, duality_gap_threshold=1e-7, dual_error_threshold=1e-9,
                                        primal_error_threshold=1e-9)
    assert r.error_code == 0 and o["error_code"] == 0
    assert abs(r.primal_objective - o["p_obj"]) <= 1e-6
    assert abs(r.iterations - o["iterations"]) <= 2
    n = min(3, len(r.history))
    assert np.allclose(r.history[:n, 1], o["hist"][:n, 1], rtol=1e-6)      # mu trace
    assert np.allclose(r.history[:n, 8:10], o["hist"][:n, 8:10], rtol=1e-3)  # step lengths


GOLDEN_FULL = ["x2p1", "polyopt8", "delsarte_8_3", "ce_8_3", "ns_8_3_2", "sdpa_small", "polyopt40", "delsarte_3_10", "threepoint_4"]


@pytest.mark.parametrize("path", list(PATHS))
@pytest.mark.parametrize("name", GOLDEN_FULL + ["ce_8_15", "ns_8_15_2"])
def test_schur_assemble_matches_256bit_golden(name, path):
    """HIP assembly against the committed 256-bit vectors (tests/golden/make_golden.py): 1e-12 * max|S|,
    the same bar the fp64 oracle meets (tests/test_oracle_cpu.py)."""
    import os
    from clrs_amd.solver import SchurContext
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    ctx = SchurContext(flat(name), **PATHS[path])
    S, _ = ctx.compute_S_integrated(g["Xchol"], g["Y"], want_AY=False)
    assert np.max(np.abs(S - g["S"])) <= 1e-12 * np.max(np.abs(g["S"]))
    ctx.close()


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "staged"])
@pytest.mark.parametrize("name", GOLDEN_FULL)
def test_factor_solve_matches_256bit_golden(name, fused):
    """(dx, dy) against the 256-bit LU solution of the full KKT system; 1e-7 relative (fp64 eps x cond ~1e8)."""
    import os
    from clrs_amd.solver import SchurContext, compute_T_decomposition, solve_system
    f = flat(name)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    ctx = SchurContext(f, fused=fused)
    compute_T_decomposition(ctx, g["Xchol"], g["Y"])
    dx, dy = solve_system(ctx, g["rhs_x"], g["rhs_y"])
    scale = max(1.0, np.max(np.abs(g["dx"])), np.max(np.abs(g["dy"])) if f.n_free else 0.0)
    tol = 1e-5 if name == "threepoint_4" else 1e-7       # cond(S) ~ 1e10 at these iterates for the three-point instance
    assert np.max(np.abs(dx - g["dx"])) <= tol * scale
    if f.n_free:
        assert np.max(np.abs(dy - g["dy"])) <= tol * scale
    ctx.close()


def test_split_phase_device_api_two_shards_on_one_gpu(oracle_built):
    """The sharded protocol (clrs_schur_factor_local_dev / _finish_dev, clrs_schur_solve_fwd_dev / _bwd_dev) with two
    'ranks' living in one process on one GPU: partial Q and u are summed by hand where RCCL would all-reduce."""
    import torch
    from clrs_amd.sharded import HipLocal, ShardedSchur
    from oracle.oracle import Oracle
    f = flat("ns_8_3_2")
    X, Y = spd_iterates(f, seed=41)
    Xc = chol_blocks_np(f, X)
    rng = np.random.default_rng(42)
    rx, ry = rng.standard_normal(f.x_len), rng.standard_normal(f.n_free)
    torch.cuda.set_device(0)
    ranks = [ShardedSchur(f, r, 2, lambda s: HipLocal(s, 0)) for r in range(2)]
    for sh in ranks:
        sh.world = 1                      # no process group here: the exchange is done below
    dev = "cuda:0"
    tx = [torch.from_numpy(sh.take_xy(Xc)).to(dev) for sh in ranks]
    ty = [torch.from_numpy(sh.take_xy(Y)).to(dev) for sh in ranks]
    qs = []
    for sh, a, b in zip(ranks, tx, ty):
        sh.local.assemble(a, b)
        qs.append(sh.local.factor_local())
    total = qs[0] + qs[1]
    for q in qs:
        q.copy_(total)
    for sh in ranks:
        sh.local.factor_finish()
        assert sh.local.status() == 0
    us = [sh.local.solve_fwd(torch.from_numpy(sh.take_x(rx)).to(dev)) for sh in ranks]
    total = us[0] + us[1]
    for u in us:
        u.copy_(total)
    o = Oracle(f)
    o.schur_assemble(Xc, Y)
    assert o.schur_factor() == 0
    dx_ref, dy_ref = o.schur_solve(rx, ry)
    try_y = torch.from_numpy(ry).to(dev)
    for sh in ranks:
        dx = torch.empty(sh.shard.x_len, dtype=torch.float64, device=dev)
        dy = torch.empty(f.n_free, dtype=torch.float64, device=dev)
        sh.local.solve_bwd(try_y, dx, dy)
        torch.cuda.synchronize()
        ref = np.concatenate([dx_ref[f.cluster_off[j]:f.cluster_off[j + 1]] for j in sh.clusters])
        assert np.max(np.abs(dx.cpu().numpy() - ref)) <= 1e-7 * max(1.0, np.max(np.abs(dx_ref)))
        assert np.max(np.abs(dy.cpu().numpy() - dy_ref)) <= 1e-7 * max(1.0, np.max(np.abs(dy_ref)))
        sh.close()


@pytest.mark.parametrize("name", ["ns_8_3_2", "ce_8_15_well"])
def test_merged_exchange_order_two_shards_on_one_gpu(name, oracle_built):
    """The order ShardedSchur uses by default: factor_local, solve_fwd, ONE exchange of the contiguous [Q | u] buffer,
    factor_finish, solve_bwd -- two 'ranks' in one process on one GPU, the all-reduce replaced by a sum by hand; then a second
    solve (u alone) through the same objects."""
    import torch
    from clrs_amd.sharded import HipLocal, ShardedSchur
    from oracle.oracle import Oracle
    f = flat("ce_8_3" if name == "ce_8_15_well" else name)
    X, Y = spd_iterates(f, seed=43)
    Xc = chol_blocks_np(f, X)
    rng = np.random.default_rng(44)
    rx, ry = rng.standard_normal(f.x_len), rng.standard_normal(f.n_free)
    torch.cuda.set_device(0)
    dev = "cuda:0"
    ranks = [ShardedSchur(f, r, 2, lambda s: HipLocal(s, 0)) for r in range(2)]
    bufs = [sh.local.qu for sh in ranks]

    def fake_all_reduce(t):                      # called by rank 1 last: every buffer becomes the sum
        if t.data_ptr() == bufs[1].data_ptr() or t.data_ptr() == ranks[1].local.u.data_ptr():
            n, off = t.numel(), (0 if t.data_ptr() == bufs[1].data_ptr() else f.n_free * f.n_free)
            total = bufs[0][off:off + n] + bufs[1][off:off + n]
            for b in bufs:
                b[off:off + n].copy_(total)
    for sh in ranks:
        sh._all_reduce = fake_all_reduce
        sh.force_split = True
    o = Oracle(f)
    o.schur_assemble(Xc, Y)
    assert o.schur_factor() == 0
    dx_ref, dy_ref = o.schur_solve(rx, ry)
    txs = [torch.from_numpy(sh.take_x(rx)).to(dev) for sh in ranks]
    t_y = torch.from_numpy(ry).to(dev)
    for sh in ranks:
        sh.local.assemble(torch.from_numpy(sh.take_xy(Xc)).to(dev), torch.from_numpy(sh.take_xy(Y)).to(dev))
        sh._q = sh.local.factor_local()
    for rep in range(2):
        # the two ranks step through solve() in lock step: fwd on both, exchange, finish + bwd on both
        us = [sh.local.solve_fwd(tx) for sh, tx in zip(ranks, txs)]
        if rep == 0:
            fake_all_reduce(bufs[1])
            for sh in ranks:
                sh.local.factor_finish()
                sh._q = None
                assert sh.local.status() == 0
        else:
            fake_all_reduce(us[1])
        for sh in ranks:
            dx = torch.empty(sh.shard.x_len, dtype=torch.float64, device=dev)
            dy = torch.empty(f.n_free, dtype=torch.float64, device=dev)
            sh.local.solve_bwd(t_y, dx, dy)
            torch.cuda.synchronize()
            ref = np.concatenate([dx_ref[f.cluster_off[j]:f.cluster_off[j + 1]] for j in sh.clusters])
            assert np.max(np.abs(dx.cpu().numpy() - ref)) <= 1e-7 * max(1.0, np.max(np.abs(dx_ref))), rep
            assert np.max(np.abs(dy.cpu().numpy() - dy_ref)) <= 1e-7 * max(1.0, np.max(np.abs(dy_ref))), rep
    for sh in ranks:
        sh.close()


def test_cholesky_blocks_device_entry():
    import torch
    from clrs_amd.solver import SchurContext
    f = flat("ns_8_3_2")
    X, _ = spd_iterates(f, seed=6)
    ctx = SchurContext(f)
    tX = torch.from_numpy(X).to("cuda:0")
    tL = torch.empty_like(tX)
    torch.cuda.synchronize()
    ctx.cholesky_blocks_dev(tX.data_ptr(), tL.data_ptr())
    assert ctx.sync_status_cholesky() == 0
    assert np.max(np.abs(tL.cpu().numpy() - chol_blocks_np(f, X))) <= 1e-13 * np.max(np.abs(X))
    ctx.close()


def test_solvesdp_three_point_bound_config4():
    """BASELINE config 4 end to end through the HIP path: three_point_spherical_codes(4, 1//6, -1, 4) = 10 +- 1e-5 with
    omega = 1e3 as in the reference's test (test/runtests_solver.jl:26-27).  fp64 reaches the value; it may stop on a
    failed Cholesky once the gap is below ~1e-7 (error_code 1, the reference's behaviour at too low a precision)."""
    from clrs_amd.solver import solvesdp
    r = solvesdp(flat("threepoint_4"), omega_p=1e3, omega_d=1e3, maxiterations=200)
    assert r.error_code in (0, 1), (r.status, r.iterations)
    assert abs(r.primal_objective - 10.0) <= 1e-5 and abs(r.dual_objective - 10.0) <= 1e-5, (r.primal_objective, r.dual_objective)


def test_solvesdp_sdpa_example_config5(oracle_built):
    """BASELINE config 5, test/example.dat-s: dense-constraint path, no free variables; objective 30 (oracle, unpinned by the reference)."""
    from clrs_amd.solver import solvesdp
    from oracle.oracle import Oracle
    f = flat("sdpa_example")
    r = solvesdp(f, omega_p=1e2, omega_d=1e2)
    o = Oracle(f, quad=True).solvesdp(omega_p=1e2, omega_d=1e2, duality_gap_threshold=1e-7, dual_error_threshold=1e-9,
                                      primal_error_threshold=1e-9)
    assert r.error_code == 0 and o["error_code"] == 0
    assert abs(r.primal_objective - o["p_obj"]) <= 1e-5 and abs(r.primal_objective - 30.0) <= 1e-4


# ---- device-resident interior-point loop (clrs_ipm_*) ------------------------------------------

DEVICE_LOOP_CASES = [("x2p1", 1.0, 1e-6, {}), ("polyopt40", None, 1e-6, {}), ("delsarte_3_10", 13.158314, 1e-5, {}),
                     ("delsarte_8_3", 240.0, 1e-6, {}), ("sdpa_example", 30.0, 1e-6, dict(omega_p=1e2, omega_d=1e2))]


@pytest.mark.parametrize("name,expected,tol,kw", DEVICE_LOOP_CASES)
def test_device_loop_reaches_the_pinned_objectives(name, expected, tol, kw, oracle_built):
    """The whole iteration on the GPU: same answers as the reference pins (test/runtests_solver.jl:15,86-87; README.md:149);
    polyopt 2d=40 (BASELINE config 2) against the quad-precision oracle loop."""
    from clrs_amd.solver import solvesdp_device
    from oracle.oracle import Oracle
    f = flat(name)
    r = solvesdp_device(f, **kw)
    assert r.error_code == 0 and r.status == "Optimal", (r.status, r.error_code, r.iterations)
    if expected is None:
        expected = Oracle(f, quad=True).solvesdp(omega_p=1e4, omega_d=1e4, duality_gap_threshold=1e-9, dual_error_threshold=1e-9,
                                                 primal_error_threshold=1e-9)["p_obj"]
    assert abs(r.primal_objective - expected) <= tol * max(1.0, abs(expected))
    assert abs(r.dual_objective - expected) <= tol * max(1.0, abs(expected))


@pytest.mark.parametrize("name", ["polyopt8", "delsarte_3_10", "polyopt40"])
def test_device_loop_follows_the_host_loop(name):
    """Same algorithm, same fp64 arithmetic up to summation order: the iteration traces (mu, step lengths) of the device loop and
    of the host-orchestrated loop agree to 1e-6 over the first iterations, and the iteration counts within 2."""
    from clrs_amd.solver import solvesdp, solvesdp_device
    f = flat(name)
    rd, rh = solvesdp_device(f), solvesdp(f)
    assert rd.error_code == 0 and rh.error_code == 0
    assert abs(rd.iterations - rh.iterations) <= 2
    n = min(6, len(rd.history), len(rh.history))
    assert np.allclose(rd.history[:n, 1], rh.history[:n, 1], rtol=1e-6)          # mu
    assert np.allclose(rd.history[:n, 8:10], rh.history[:n, 8:10], rtol=1e-5)    # alpha_d, alpha_p
    assert abs(rd.primal_objective - rh.primal_objective) <= 1e-6 * max(1.0, abs(rh.primal_objective))


def test_device_loop_three_point_bound():
    """BASELINE config 4 on the device loop: 10 +- 1e-5 (fp64 stalls on the step length near gap 1e-8, like the fp64 oracle)."""
    from clrs_amd.solver import solvesdp_device
    r = solvesdp_device(flat("threepoint_4"), omega_p=1e3, omega_d=1e3, maxiterations=200)
    assert r.error_code in (0, 1, 4)
    assert abs(r.primal_objective - 10.0) <= 1e-5 and abs(r.dual_objective - 10.0) <= 1e-5


def test_device_loop_reports_failure_like_the_reference():
    """cohnelkies(8,15) cannot be factored in fp64: the device loop stops with error_code 1 at the first iteration and leaves the
    starting point untouched (the reference returns the current iterate, src/solver.jl:594-623)."""
    from clrs_amd.solver import solvesdp_device
    f = flat("ce_8_15")
    r = solvesdp_device(f)
    assert r.error_code == 1 and r.iterations == 0
    assert np.all(np.isfinite(r.X)) and np.all(np.isfinite(r.x)) and np.allclose(r.x, 0.0)


# ---- malformed descriptions and edge shapes through the C ABI ------------------------------------------------------------------

def _mini_sdp(rank2=False, drop_partner=False, m=1):
    """One cluster, P = 3, N = 1, one low-rank block (optionally m = 2 sub-blocks), one 1 x 1 dense block."""
    import clrs_amd
    from clrs_amd.sdp import Block, ClusteredLowRankSDP, HiLo, LowRankMat
    rng = np.random.default_rng(7)
    dl = 3
    ent = {}
    for r in range(m):
        for s in range(m):
            if drop_partner and (r, s) == (1, 0):
                continue
            ent[(r, s)] = {}
    for p in range(3):
        v = rng.standard_normal((2 if rank2 else 1, dl))
        lam = np.array([1.0, 0.5][: v.shape[0]])
        for r in range(m):
            for s in range(m):
                if (r, s) in ent:
                    ent[(r, s)][p] = LowRankMat(lam, v, v)
    blocks = [Block(m, dl, ent, "lr"), Block(1, 1, {(0, 0): {0: HiLo.of(np.array([[2.0]])), 2: HiLo.of(np.array([[0.5]]))}}, "dense")]
    sdp = ClusteredLowRankSDP(maximize=True, constant=0.0, blocks=[blocks], B=[rng.standard_normal((3, 1))], c=[np.ones(3)],
                              C=[[np.zeros((m * dl, m * dl)), np.zeros((1, 1))]], b=np.ones(1))
    return sdp


@pytest.mark.parametrize("kw", [dict(), dict(rank2=True), dict(m=2), dict(m=2, rank2=True)])
def test_small_handmade_shapes_match_oracle(kw, oracle_built):
    """rank-2 terms, sub-blocks (m = 2), a dense block touching only some constraints, de-duplicated vectors."""
    import clrs_amd
    from clrs_amd.solver import SchurContext, compute_T_decomposition, solve_system
    from oracle.oracle import Oracle
    f = clrs_amd.flatten(_mini_sdp(**kw))
    X, Y = spd_iterates(f, seed=3)
    Xc = chol_blocks_np(f, X)
    o = Oracle(f, quad=True, use_lo=False)
    S_ref, AY_ref = o.schur_assemble(Xc, Y)
    assert o.schur_factor() == 0
    dx_ref, dy_ref = o.schur_solve(np.ones(f.x_len), np.ones(f.n_free))
    for path in PATHS.values():
        ctx = SchurContext(f, **path)
        _, S, AY = compute_T_decomposition(ctx, Xc, Y, want_S=True)
        assert np.max(np.abs(S - S_ref)) <= 1e-12 * np.max(np.abs(S_ref))
        assert np.max(np.abs(AY - AY_ref)) <= 1e-12 * max(1.0, np.max(np.abs(AY_ref)))
        dx, dy = solve_system(ctx, np.ones(f.x_len), np.ones(f.n_free))
        assert np.max(np.abs(dx - dx_ref)) <= 1e-9 * max(1.0, np.max(np.abs(dx_ref)))
        assert np.max(np.abs(dy - dy_ref)) <= 1e-9 * max(1.0, np.max(np.abs(dy_ref)))
        ctx.close()


def test_malformed_descriptions_are_rejected():
    """The ABI validates what it is given: a term without its transposed partner (the convention of src/solver.jl:1009), a solve
    before a factorisation, wrong call order -- negative codes with a message, never a crash."""
    import clrs_amd
    from clrs_amd._lib import ClrsError
    from clrs_amd.solver import SchurContext
    sdp = _mini_sdp(m=2, drop_partner=True)
    sdp.check = lambda: None                      # bypass the host-side check so that the C layer sees the malformed input
    with pytest.raises(ClrsError, match="transposed partner"):
        SchurContext(clrs_amd.flatten(sdp))
    f = flat("polyopt8")
    ctx = SchurContext(f)
    with pytest.raises(ClrsError, match="before clrs_schur_factor"):
        ctx.solve(np.ones(f.x_len), np.ones(f.n_free))
    with pytest.raises(ClrsError, match="before clrs_schur_assemble"):
        ctx.factor()
    ctx.close()
    # a dense constraint matrix that is not symmetric: the dense branch forms lower tiles and transposed stores only (both paths)
    import copy
    from clrs_amd.mw import MwSchurContext
    g = copy.copy(flat("sdpa_small"))
    g.dense_A = g.dense_A.copy()
    n = int(g.block_n[0])
    g.dense_A[1] += 0.5                            # entry (1, 0) of the first matrix, not its mirror
    assert n > 1
    with pytest.raises(ClrsError, match="must be symmetric"):
        SchurContext(g)
    with pytest.raises(ClrsError, match="must be symmetric"):
        MwSchurContext(g, limbs=3)


# ---- BASELINE configs 4 and 5 at their named sizes ---------------------------------------------------------------------------
def test_three_point_bound_n3_2d16_config4_as_named(oracle_built):
    """BASELINE config 4 as named ("ThreePointBound spherical code n=3, 2d=16": three_point_spherical_codes(3, 1/2, 8, 8),
    examples/ThreePointBound.jl:45-160): one cluster of P = 221 constraints, 9 dense blocks (sides 9..1), 17 rank-1 1x1 blocks and
    17 low-rank blocks up to 54 x 54 with rank-1 and rank-2 terms, N = 0.  The reference pins no answer for this instance (its test
    runs (4, 1/6, -1, 4)); parity is against the oracle on seeded iterates plus the structural identities.
    Like the 2d = 30 sphere-packing problems, S of this instance is not positive definite to fp64 accuracy even on well-conditioned
    iterates: the fp64 path assembles it (checked) and reports the reference's SolverFailure; factor and solve are checked on the
    multi-word path (4 limbs) against the 320-bit oracle."""
    from clrs_amd.mw import MwSchurContext
    from clrs_amd.solver import SchurContext
    from oracle.oracle import Oracle
    from tests.util import mw_from_double, mw_relerr
    f = flat("threepoint_3_8_8")
    P = 221
    assert f.n_clusters == 1 and int(f.cluster_P[0]) == P and f.n_free == 0 and int(np.max(f.block_n)) == 54
    X, Y = spd_iterates(f, seed=4)
    Xc = chol_blocks_np(f, X)
    o64 = Oracle(f, quad=False)
    S64, _ = o64.schur_assemble(Xc, Y)
    S_dense = o64.schur_dense_check(Xc, Y)                    # low-rank branch == dense trace formula (src/interface.jl:798-800)
    assert np.max(np.abs(S64 - S_dense)) <= 1e-10 * np.max(np.abs(S64))
    o = Oracle(f, quad=True, use_lo=False)
    S_ref, AY_ref = o.schur_assemble(Xc, Y)
    ctx = SchurContext(f)
    S, AY = ctx.compute_S_integrated(Xc, Y)
    Sm = S.reshape(P, P, order="F")
    assert np.array_equal(Sm, Sm.T)
    assert np.max(np.abs(S - S_ref)) <= 1e-11 * np.max(np.abs(S_ref)), np.max(np.abs(S - S_ref)) / np.max(np.abs(S_ref))
    assert np.max(np.abs(AY - AY_ref)) <= 1e-11 * max(1.0, np.max(np.abs(AY_ref)))
    assert ctx.factor() == 1                                  # "S was not decomposed succesfully in block 1" (src/solver.jl:1249)
    ctx.close()
    # factor + solve at 4 limbs
    K = 4
    om = Oracle(f, mp_bits=320)
    mctx = MwSchurContext(f, limbs=K)
    Xm, Ym = mw_from_double(X, K), mw_from_double(Y, K)
    Xcm = mctx.cholesky_blocks(Xm)
    Sm4, _ = mctx.compute_S_integrated(Xcm, Ym)
    pad = lambda a: np.vstack([a, np.zeros((1, a.shape[1]))])
    S_mp, _ = om.schur_assemble_mw(pad(Xcm), pad(Ym))
    assert mw_relerr(Sm4, S_mp) <= 2.0 ** -(53 * K - 24), mw_relerr(Sm4, S_mp)
    assert mctx.factor() == 0 and om.schur_factor() == 0
    rhs = mw_from_double(np.random.default_rng(9).standard_normal(f.x_len), K)
    dx, _ = mctx.solve(rhs, None)
    dx_ref, _ = om.schur_solve_mw(pad(rhs), np.zeros((K + 1, 0)))
    # cond(S) of this instance on these iterates is ~1e20 (why fp64 fails): what is left of 4 limbs
    assert mw_relerr(dx, dx_ref) <= 2.0 ** -(53 * K - 110), mw_relerr(dx, dx_ref)
    mctx.close()
    # the whole solve at the reference's precision (5 limbs), omega = 1e3 as in the reference's three-point test: a valid three-point
    # bound on the kissing number in dimension 3 (12 <= bound < 13; profiles/r02/c_configs_at_256_bits.txt: 12.5228 in 42 iterations)
    from clrs_amd.mw import solvesdp_mw
    r = solvesdp_mw(f, limbs=5, omega_p=1e3, omega_d=1e3)
    assert r.error_code == 0 and r.status == "Optimal", (r.status, r.error_code, r.iterations)
    assert 12.0 <= r.primal_objective < 13.0 and r.duality_gap < 1e-15


def test_sdpa_x64_config5_as_named(oracle_built):
    """BASELINE config 5 as named ("SDPA .dat-s import scaled x64 blocks": sdpa_scaled(nb=64, bs=32, m=256), SURVEY.md section 8d
    row 5; reader semantics of src/SDPAtoCLRS.jl:3-83): 64 dense 32 x 32 blocks, P = 256, no low-rank structure, N = 0 -- the dense
    ("high rank") branch of compute_S_integrated! (src/solver.jl:1089-1104) on every block."""
    from clrs_amd.solver import SchurContext, compute_T_decomposition, solve_system, solvesdp_device
    from oracle.oracle import Oracle
    f = flat("sdpa_x64")
    assert f.n_blocks == 64 and int(f.cluster_P[0]) == 256 and np.all(f.block_kind == 1) and np.all(f.block_n == 32)
    X, Y = spd_iterates(f, seed=6)
    Xc = chol_blocks_np(f, X)
    o = Oracle(f, quad=True, use_lo=False)
    S_ref, _ = o.schur_assemble(Xc, Y)
    assert o.schur_factor() == 0
    rhs = np.random.default_rng(10).standard_normal(f.x_len)
    dx_ref, _ = o.schur_solve(rhs, np.zeros(0))
    for kw in (dict(), dict(fused=False)):
        ctx = SchurContext(f, **kw)
        _, S, _ = compute_T_decomposition(ctx, Xc, Y, want_S=True)
        assert np.max(np.abs(S - S_ref)) <= 1e-11 * np.max(np.abs(S_ref)), (kw, np.max(np.abs(S - S_ref)) / np.max(np.abs(S_ref)))
        Sm = S.reshape(256, 256, order="F")
        assert np.array_equal(Sm, Sm.T)
        dx, _ = solve_system(ctx, rhs, np.zeros(0))
        assert np.max(np.abs(dx - dx_ref)) <= 1e-8 * max(1.0, np.max(np.abs(dx_ref)))
        ctx.close()
    # the whole solve, device resident, against the oracle loop (the reference pins no objective for a generated SDPA instance)
    r = solvesdp_device(f)
    ro = Oracle(f, quad=False).solvesdp(omega_p=1e4, omega_d=1e4, duality_gap_threshold=1e-7, dual_error_threshold=1e-9, primal_error_threshold=1e-9)
    assert r.error_code == 0 and ro["error_code"] == 0, (r.status, r.error_code, ro["error_code"])
    assert abs(r.primal_objective - ro["p_obj"]) <= 1e-6 * max(1.0, abs(ro["p_obj"]))


def test_device_loop_beta_follows_the_reference_order_across_the_feasibility_flip(oracle_built):
    """beta_c of the iteration in which the iterate becomes feasible is chosen with the PREVIOUS feasibility
    (src/solver.jl:429-434 before :441-447).  The per-iteration (mu, beta_c) columns of the fp64 device loop agree with the oracle's."""
    from clrs_amd.solver import solvesdp_device
    from oracle.oracle import Oracle
    f = flat("delsarte_3_10")
    r = solvesdp_device(f)
    ro = Oracle(f, quad=False).solvesdp(omega_p=1e4, omega_d=1e4, duality_gap_threshold=1e-7, dual_error_threshold=1e-9, primal_error_threshold=1e-9)
    assert r.iterations == ro["iterations"]
    flips = 0
    for it in range(r.iterations):
        assert abs(r.history[it, 10] - ro["hist"][it, 10]) <= 1e-6 * max(1.0, ro["hist"][it, 10]), (it, r.history[it, 10], ro["hist"][it, 10])
        assert abs(r.history[it, 1] - ro["hist"][it, 1]) <= 1e-5 * ro["hist"][it, 1]
        if it and (r.history[it, 10] == 0.1) != (r.history[it - 1, 10] == 0.1):
            flips += 1
    assert flips >= 1


def test_ipm_context_reused_for_another_objective(oracle_built):
    """clrs_ipm_create on a context that already holds an iteration state loads the new C, c, b (it used to solve the first
    problem again)."""
    import copy
    from clrs_amd.solver import SchurContext, solvesdp_device
    from oracle.oracle import Oracle
    f = flat("polyopt8")
    g = copy.copy(f)
    g.c = f.c * 2.0
    g.c_lo = f.c_lo * 2.0
    ctx = SchurContext(f)
    r1 = solvesdp_device(f, ctx=ctx)
    r2 = solvesdp_device(g, ctx=ctx)
    ro2 = Oracle(g, quad=False).solvesdp(omega_p=1e4, omega_d=1e4, duality_gap_threshold=1e-7, dual_error_threshold=1e-9, primal_error_threshold=1e-9)
    assert r1.error_code == 0 and r2.error_code == 0
    assert abs(r2.primal_objective - ro2["p_obj"]) <= 1e-6 * max(1.0, abs(ro2["p_obj"]))
    assert abs(r2.primal_objective - r1.primal_objective) > 1e-3 * max(1.0, abs(r1.primal_objective))
    ctx.close()


def test_sdpa_x64_full_block_diagonal_constraints(oracle_built):
    """Config 5 with every constraint matrix full block diagonal (both constraints of test/example.dat-s touch both blocks): P = 256 dense
    matrices in each of the 64 blocks -- the 7.5 GFLOP / 134 MB assembly of SURVEY.md section 8d, the instance `roofline_dense` of the bench
    line is measured on.  Assembly against the fp64 oracle, symmetry, factor + solve."""
    from clrs_amd.solver import SchurContext, compute_T_decomposition, solve_system
    from oracle.oracle import Oracle
    f = flat("sdpa_x64_full")
    assert int(f.dense_ptr[-1]) == 64 * 256
    X, Y = spd_iterates(f, seed=8)
    Xc = chol_blocks_np(f, X)
    o = Oracle(f, quad=False)
    S_ref, _ = o.schur_assemble(Xc, Y)
    assert o.schur_factor() == 0
    rhs = np.random.default_rng(11).standard_normal(f.x_len)
    dx_ref, _ = o.schur_solve(rhs, np.zeros(0))
    ctx = SchurContext(f)
    _, S, _ = compute_T_decomposition(ctx, Xc, Y, want_S=True)
    assert np.max(np.abs(S - S_ref)) <= 1e-12 * np.max(np.abs(S_ref))
    Sm = S.reshape(256, 256, order="F")
    assert np.array_equal(Sm, Sm.T)
    dx, _ = solve_system(ctx, rhs, np.zeros(0))
    assert np.max(np.abs(dx - dx_ref)) <= 1e-9 * max(1.0, np.max(np.abs(dx_ref)))
    ctx.close()


def test_dense_blocks_below_32_with_a_ragged_matrix_count(oracle_built):
    """k_dense_T32 off its 32 x 32 fast path: 8 blocks of 24 x 24 with 100 dense matrices each (not a multiple of the 8 matrices a
    workgroup takes), assembly against the fp64 oracle."""
    import clrs_amd
    from clrs_amd import problems as P
    from clrs_amd.solver import SchurContext, compute_T_decomposition
    from oracle.oracle import Oracle
    f = clrs_amd.flatten(P.sdpa_to_sdp(P.sdpa_scaled(nb=8, bs=24, m=100, seed=3, blocks_per_constraint=8)))
    assert int(f.dense_ptr[-1]) == 8 * 100
    X, Y = spd_iterates(f, seed=4)
    Xc = chol_blocks_np(f, X)
    S_ref, _ = Oracle(f, quad=False).schur_assemble(Xc, Y)
    ctx = SchurContext(f)
    ctx.set_graph_mode(False)
    ctx.set_kernel_timing(-1)
    _, S, _ = compute_T_decomposition(ctx, Xc, Y, want_S=True)
    assert "k_dense_T32" in ctx.kernel_times()
    ctx.close()
    assert np.max(np.abs(S - S_ref)) <= 1e-12 * np.max(np.abs(S_ref))


@pytest.mark.parametrize("P,N", [(129, 1), (150, 20), (200, 64), (321, 33), (260, 100), (200, 150)])
def test_one_large_cluster_factor_with_free_variables_in_one_factorisation(P, N, oracle_built):
    """One cluster beyond one 64-wide block with free variables: L, L^-1 B and Q come out of ONE blocked factorisation of
    [S .; B^T 0] that stops before the corner ("factor_aug": k_chol_pack, k_chol_level, k_chol_unpack), here with ragged last blocks,
    32 < N <= 64 (the corner's upper half is mirrored from the lower) and N beyond one block (several appended block rows).  Against
    the oracle, and against the three-stage plan."""
    import clrs_amd
    from clrs_amd import _lib
    from clrs_amd.solver import SchurContext, compute_T_decomposition, solve_system
    from oracle.oracle import Oracle
    from tests.util import random_simple_sdp
    lr = (P + 119) // 120                       # sides 16: 136 independent pairings per block
    f = clrs_amd.flatten(random_simple_sdp(3000 + P + N, J=1, n_free=N, fixed_P=P, max_n=16, lr_blocks=lr))
    X, Y = spd_iterates(f, seed=P + 3 * N)
    Xc = chol_blocks_np(f, X)
    o = Oracle(f, quad=False)
    o.schur_assemble(Xc, Y)
    assert o.schur_factor() == 0
    L_ref, LinvB_ref, LQ_ref = o.get_factor()
    rng = np.random.default_rng(P * 7 + N)
    rx, ry = rng.standard_normal(f.x_len), rng.standard_normal(f.n_free)
    dx_ref, dy_ref = o.schur_solve(rx, ry)
    got = {}
    for aug in (1, 0):
        assert _lib.load().clrs_config_set(b"factor_aug", aug) == 0
        try:
            ctx = SchurContext(f, fused=False)
            ctx.set_graph_mode(False)
            ctx.set_kernel_timing(-1)
            compute_T_decomposition(ctx, Xc, Y, want_S=False)
            assert ("k_chol_pack" in ctx.kernel_times()) == (aug == 1)
            got[aug] = ctx.get_factor() + solve_system(ctx, rx, ry)
            ctx.close()
        finally:
            _lib.load().clrs_config_set(b"factor_aug", 1)
    for aug in (1, 0):
        L, LinvB, LQ, dx, dy = got[aug]
        assert np.max(np.abs(L - L_ref)) <= 1e-9 * np.max(np.abs(L_ref))
        assert np.max(np.abs(LinvB - LinvB_ref)) <= 1e-9 * max(1.0, np.max(np.abs(LinvB_ref)))
        assert np.max(np.abs(LQ - LQ_ref)) <= 1e-9 * max(1.0, np.max(np.abs(LQ_ref)))
        assert np.max(np.abs(dx - dx_ref)) <= 1e-7 * max(1.0, np.max(np.abs(dx_ref)))
        assert np.max(np.abs(dy - dy_ref)) <= 1e-7 * max(1.0, np.max(np.abs(dy_ref)))


def test_inverse_blocks_of_S_are_formed_once_per_factorisation(oracle_built):
    """polyopt_scaled_300 (P = 601 > 512): the staged solves go through the inverted 512 x 512 diagonal blocks of L; the first solve after a
    factorisation forms them (k_trtri_diag), the second solve and both backward substitutions reuse them, and a new factorisation forms
    them again.  Every solve against the oracle."""
    from clrs_amd.solver import SchurContext, compute_T_decomposition, solve_system
    from oracle.oracle import Oracle
    f = flat("polyopt_scaled_300")
    ctx = SchurContext(f, fused=False)
    ctx.set_graph_mode(False)
    o = Oracle(f, quad=False)
    rng = np.random.default_rng(3)
    for round_ in range(2):
        X, Y = spd_iterates(f, seed=20 + round_)
        Xc = chol_blocks_np(f, X)
        compute_T_decomposition(ctx, Xc, Y)
        o.schur_assemble(Xc, Y)
        assert o.schur_factor() == 0
        ctx.set_kernel_timing(-1)
        for solve in range(2):
            rx, ry = rng.standard_normal(f.x_len), rng.standard_normal(f.n_free)
            dx, dy = solve_system(ctx, rx, ry)
            dx_ref, dy_ref = o.schur_solve(rx, ry)
            assert np.max(np.abs(dx - dx_ref)) <= 1e-8 * max(1.0, np.max(np.abs(dx_ref)))
            assert np.max(np.abs(dy - dy_ref)) <= 1e-8 * max(1.0, np.max(np.abs(dy_ref)))
            assert ctx.kernel_times()["k_trtri_diag"][2] == 1          # one launch per factorisation, by the first forward substitution
        ctx.set_kernel_timing(-2)
    ctx.close()



@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("max_n,max_P", [(32, 48), (24, 64), (16, 64)])
def test_wave4_random_structures(seed, max_n, max_P, oracle_built):
    """k_cluster_assemble_w4 on randomised structure: clusters of different sizes P <= 48 (three column tiles) or <= 64 (four) in one context, 1-3
    low-rank blocks of different sides n <= 32 per cluster (blocks of at most 16 rows beside blocks of two row tiles, partial tiles and partial
    k-steps), 0-2 dense 1 x 1 blocks touching a subset of the constraints, placed anywhere among them.  Clusters within reach of
    k_cluster_assemble_w3 (every n <= 16 and P <= 32) stay there.  Assembly against the oracle, entry by entry, and S_j exactly symmetric."""
    import clrs_amd
    from clrs_amd.solver import SchurContext
    from oracle.oracle import Oracle
    from tests.util import random_simple_sdp
    f = clrs_amd.flatten(random_simple_sdp(1000 + seed, J=4 + seed % 3, n_free=seed % 3, max_P=max_P, max_n=max_n))
    X, Y = spd_iterates(f, seed=seed + 300)
    Xc = chol_blocks_np(f, X)
    o = Oracle(f, quad=False)
    S_ref, AY_ref = o.schur_assemble(Xc, Y)
    beyond = 0
    for j in range(f.n_clusters):
        bl = [b for b in range(f.n_blocks) if f.block_cluster[b] == j and f.block_kind[b] == 0]
        if int(f.cluster_P[j]) > 32 or any(int(f.block_n[b]) > 16 for b in bl):
            beyond += 1
    ctx = SchurContext(f, wave2=True)
    assert ctx.wave4_clusters() == beyond
    assert ctx.wave4_clusters() + ctx.wave2_clusters() == f.n_clusters
    S, AY = ctx.compute_S_integrated(Xc, Y)
    ctx.close()
    for j in range(f.n_clusters):
        sl = slice(int(f.S_off[j]), int(f.S_off[j + 1]))
        assert np.max(np.abs(S[sl] - S_ref[sl])) <= 1e-11 * np.max(np.abs(S_ref[sl])), j
        P = int(f.cluster_P[j])
        Sj = S[sl].reshape(P, P, order="F")
        assert np.array_equal(Sj, Sj.T)
    assert np.max(np.abs(AY - AY_ref)) <= 1e-11 * max(1.0, np.max(np.abs(AY_ref)))
    # the general kernels on the same problem: the same numbers to rounding
    ctx0 = SchurContext(f, wave2=True, wave4=False)
    assert ctx0.wave4_clusters() == 0
    S0, AY0 = ctx0.compute_S_integrated(Xc, Y)
    ctx0.close()
    assert np.max(np.abs(S - S0)) <= 1e-11 * np.max(np.abs(S_ref))


@pytest.mark.parametrize("name,copies", [("polyopt40", 1), ("polyopt40", 1500)])
def test_wave4_many_clusters_and_rare_paths(name, copies, oracle_built):
    """k_cluster_assemble_w4 with several clusters per wave (more clusters than resident waves) and its rarely taken paths: the constraints of a
    cluster in a permuted order (S_j stored through the vector -> constraint table), a second low-rank block with negative lambdas in the cluster;
    every cluster against the oracle, and the device-pointer entry on other iterates."""
    import torch
    torch.cuda.set_device(0)
    from clrs_amd.sdp import replicate_clusters
    from clrs_amd.solver import SchurContext
    from clrs_amd.sharded import _DevArray
    from oracle.oracle import Oracle
    from tests.util import duplicate_block, permute_cluster_constraints
    f = flat(name)
    g = duplicate_block(f, 0, -0.75)
    g = permute_cluster_constraints(g, seed=5)
    big = replicate_clusters(g, copies) if copies > 1 else g
    X, Y = spd_iterates(big, seed=21)
    Xc = chol_blocks_np(big, X)
    ctx = SchurContext(big)
    assert ctx.wave4_clusters() == big.n_clusters
    S, AY = ctx.compute_S_integrated(Xc, Y)
    X2, Y2 = spd_iterates(big, seed=22)
    Xc2 = chol_blocks_np(big, X2)
    tX, tY = torch.from_numpy(Xc2).to("cuda:0"), torch.from_numpy(Y2).to("cuda:0")
    torch.cuda.synchronize()
    ctx.assemble_dev(tX.data_ptr(), tY.data_ptr())
    torch.cuda.synchronize()
    S2 = torch.as_tensor(_DevArray(ctx.S_buffer(), big.S_len), device="cuda:0").cpu().numpy()
    ctx.close()
    o = Oracle(g, quad=False)
    nxy, nS, nT = g.xy_len, g.S_len, g.n_terms
    for k in sorted(set((0, copies // 2, copies - 1))):
        Sk, _ = o.schur_assemble(Xc2[k * nxy:(k + 1) * nxy], Y2[k * nxy:(k + 1) * nxy])
        assert np.max(np.abs(S2[k * nS:(k + 1) * nS] - Sk)) <= 1e-11 * np.max(np.abs(Sk))
    step = max(1, copies // 40)
    for k in list(range(0, copies, step)) + [copies - 1]:
        Sk, AYk = o.schur_assemble(Xc[k * nxy:(k + 1) * nxy], Y[k * nxy:(k + 1) * nxy])
        assert np.max(np.abs(S[k * nS:(k + 1) * nS] - Sk)) <= 1e-11 * np.max(np.abs(Sk)), k
        assert np.max(np.abs(AY[k * nT:(k + 1) * nT] - AYk)) <= 1e-11 * max(1.0, np.max(np.abs(AYk))), k


@pytest.mark.parametrize("copies,extra_block", [(1, False), (1, True), (600, True)])
def test_wave5_matrix_valued_blocks(copies, extra_block, oracle_built):
    """k_cluster_assemble_w5 on the 2 x 2 blocks of 16 x 16 sub-blocks of Nsphere_packing(8, 15, [1/2, 1/2]) (3 x 32 constraints on 32 shared sample vectors,
    two such blocks per cluster, the second with lambda_u != 1): S_j accumulates in memory across the cluster's blocks -- alone, with a third block of
    negative lambdas in the cluster, and with several clusters per wave; every checked cluster against the oracle (S_j and A_Y), S_j exactly symmetric,
    and the general kernels on the same problem to rounding.  A cluster whose constraints are not in pair-major vector order stays on the general kernel."""
    from clrs_amd.sdp import replicate_clusters
    from clrs_amd.solver import SchurContext
    from oracle.oracle import Oracle
    from tests.util import duplicate_block, permute_cluster_constraints
    f = flat("ns_8_15_2")
    big_block = [b for b in range(f.n_blocks) if int(f.block_m[b]) == 2 and int(f.block_delta[b]) == 16][0]
    g = duplicate_block(f, big_block, -0.6) if extra_block else f
    big = replicate_clusters(g, copies) if copies > 1 else g
    X, Y = spd_iterates(big, seed=31)
    Xc = chol_blocks_np(big, X)
    ctx = SchurContext(big)
    assert ctx.wave5_clusters() == copies
    S, AY = ctx.compute_S_integrated(Xc, Y)
    ctx.close()
    ctx0 = SchurContext(big, wave5=False)
    assert ctx0.wave5_clusters() == 0
    S0, AY0 = ctx0.compute_S_integrated(Xc, Y)
    ctx0.close()
    assert np.max(np.abs(S - S0)) <= 1e-11 * np.max(np.abs(S0))
    assert np.max(np.abs(AY - AY0)) <= 1e-11 * max(1.0, np.max(np.abs(AY0)))
    o = Oracle(g, quad=False)
    nxy, nS, nT = g.xy_len, g.S_len, g.n_terms
    j5 = int(g.block_cluster[big_block]); P5 = int(g.cluster_P[j5])
    for k in sorted(set(list(range(0, copies, max(1, copies // 12))) + [copies - 1])):
        Sk, AYk = o.schur_assemble(Xc[k * nxy:(k + 1) * nxy], Y[k * nxy:(k + 1) * nxy])
        for j in range(g.n_clusters):
            sl = slice(int(g.S_off[j]), int(g.S_off[j + 1]))
            assert np.max(np.abs(S[k * nS:(k + 1) * nS][sl] - Sk[sl])) <= 1e-11 * np.max(np.abs(Sk[sl])), (k, j)
        assert np.max(np.abs(AY[k * nT:(k + 1) * nT] - AYk)) <= 1e-11 * max(1.0, np.max(np.abs(AYk))), k
        Sj = S[k * nS:(k + 1) * nS][int(g.S_off[j5]):int(g.S_off[j5 + 1])].reshape(P5, P5, order="F")
        assert np.array_equal(Sj, Sj.T)
    if copies == 1:
        h = permute_cluster_constraints(g, seed=3)
        ctxp = SchurContext(h)
        assert ctxp.wave5_clusters() == 0
        Sp, AYp = ctxp.compute_S_integrated(Xc, Y)
        ctxp.close()
        Sr, AYr = Oracle(h, quad=False).schur_assemble(Xc, Y)
        assert np.max(np.abs(Sp - Sr)) <= 1e-11 * np.max(np.abs(Sr))
