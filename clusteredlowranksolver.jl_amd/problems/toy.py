"""The tiny dense SDPs whose answers the reference's own tests pin (all take the dense, "high rank" branch of the hot path,
reference src/solver.jl:1089-1104 -- every MOI/JuMP problem does, ext/MOIExt.jl produces dense matrices only):

    theta_c5()        Lovasz number of the 5-cycle = sqrt(5)                  examples/jump.jl:4-35,  test/moi_tests.jl:6-8   (1e-30)
    povm_2x2()        minimum-error discrimination of two qubit states
                      = sqrt(2)/4 + 1/2                                        examples/jump.jl:37-55, test/moi_tests.jl:9-10 (1e-30)
    toy_z()           max z s.t. z + z2 = 1  = 1                               test/runtests_solver.jl:30-51
    toy_z_as_free()   the same with z modelled as a free variable = 1          test/runtests_solver.jl:31-38, src/interface.jl:652-752

Written down directly in the solver's standard form  sum_l <A_p^l, Y_l> + (B y)_p = c_p  (src/interface.jl:478-483); the instances of
the linear-dependency suite (test/runtests_solver.jl:249-314) are not reproduced: every one of them has a singular Schur complement as
written (two constraints with the same 1 x 1 matrix: S = a [1 1; 1 1]) and is only solvable after `preprocess!` has eliminated free
variables and constraints (src/pre_postprocessing.jl), which is outside the path.
"""
from __future__ import annotations

import numpy as np

from ..sdp import Block, ClusteredLowRankSDP, HiLo


def dense_sdp(block_sizes, constraints, objective, maximize, free=None, b=None, constant=0.0, clusters=None, names=None) -> ClusteredLowRankSDP:
    """`constraints`: list of (c_p, {block index: n x n matrix}[, row of B]); `objective`: {block index: n x n matrix};
    `clusters`: list of lists of constraint indices (default: one cluster); a block belongs to the cluster of the constraints that use it
    (reference src/interface.jl:850-885: constraints are clustered by the positive semidefinite variables they share)."""
    nb = len(block_sizes)
    N = 0 if b is None else len(b)
    if clusters is None:
        clusters = [list(range(len(constraints)))]
    owner = {}
    for j, cons in enumerate(clusters):
        for p in cons:
            for l in constraints[p][1]:
                if owner.setdefault(l, j) != j:
                    raise ValueError("a block is used by two clusters")
    blocks, Cs, Bs, cs = [], [], [], []
    for j, cons in enumerate(clusters):
        mine = [l for l in range(nb) if owner.get(l) == j]
        bl, Cl = [], []
        for l in mine:
            n = block_sizes[l]
            ent = {}
            for q, p in enumerate(cons):
                A = constraints[p][1].get(l)
                if A is not None:
                    A = np.asarray(A, dtype=np.float64).reshape(n, n)
                    ent[q] = HiLo.of((A + A.T) / 2)
            bl.append(Block(m=1, delta=n, entries={(0, 0): ent}, name=(names[l] if names else l)))
            Cl.append(np.asarray(objective.get(l, np.zeros((n, n))), dtype=np.float64).reshape(n, n))
        blocks.append(bl)
        Cs.append(Cl)
        Bj = np.zeros((len(cons), N))
        for q, p in enumerate(cons):
            if len(constraints[p]) > 2 and N:
                Bj[q, :] = constraints[p][2]
        Bs.append(Bj)
        cs.append(np.array([float(constraints[p][0]) for p in cons]))
    return ClusteredLowRankSDP(maximize=bool(maximize), constant=float(constant), blocks=blocks, B=Bs, c=cs, C=Cs,
                               b=np.zeros(0) if b is None else np.asarray(b, dtype=np.float64),
                               names={"free": list(free or []), "blocks": [[b_.name for b_ in cl] for cl in blocks]})


def theta_c5() -> ClusteredLowRankSDP:
    """max <J, X> s.t. X_ij = 0 for the non-edges of the 5-cycle, tr X = 1, X psd 5 x 5: sqrt(5)."""
    E = {(0, 1), (1, 2), (2, 3), (3, 4), (4, 0)}
    cons = []
    for i in range(5):
        for j in range(i + 1, 5):
            if (i, j) not in E and (j, i) not in E:
                A = np.zeros((5, 5)); A[i, j] = A[j, i] = 1.0
                cons.append((0.0, {0: A}))
    cons.append((1.0, {0: np.eye(5)}))
    return dense_sdp([5], cons, {0: np.ones((5, 5))}, maximize=True, names=["X"])


def _herm_embed(H):
    """Hermitian H = R + iS  ->  real symmetric [[R, -S], [S, R]] (the embedding of HermitianPSDCone into a real PSD cone)."""
    R, S = np.real(H), np.imag(H)
    return np.block([[R, -S], [S, R]])


def povm_2x2() -> ClusteredLowRankSDP:
    """max (<rho_1, E_1> + <rho_2, E_2>)/2 s.t. E_1 + E_2 = I, E_i Hermitian psd 2 x 2, rho_1 = |-><-|, rho_2 = |-i><-i|
    (examples/jump.jl:37-55): sqrt(2)/4 + 1/2.  Each E_i is a real 4 x 4 psd variable Z_i = [[R, -S], [S, R]], the structure imposed by
    six equality constraints per variable; <rho, E> = <embed(rho), Z>/2."""
    d = 2
    states = [0.5 * np.outer([1, -1], np.conj([1, -1])), 0.5 * np.outer([1, -1j], np.conj([1, -1j]))]
    cons = []

    def unit(i, j, v=1.0):
        A = np.zeros((2 * d, 2 * d)); A[i, j] += v / 2; A[j, i] += v / 2
        return A

    for l in range(2):
        for i in range(d):
            for j in range(i, d):                    # Z[i,j] = Z[d+i, d+j]
                cons.append((0.0, {l: unit(i, j) - unit(d + i, d + j)}))
        for i in range(d):
            cons.append((0.0, {l: unit(d + i, i)}))  # diagonal of S vanishes
        for i in range(d):
            for j in range(i + 1, d):                # S antisymmetric: Z[d+i, j] = -Z[d+j, i]
                cons.append((0.0, {l: unit(d + i, j) + unit(d + j, i)}))
    for i in range(d):
        for j in range(i, d):                        # real part of E_1 + E_2 = I
            cons.append((1.0 if i == j else 0.0, {0: unit(i, j), 1: unit(i, j)}))
    for i in range(d):
        for j in range(i + 1, d):                    # imaginary part
            cons.append((0.0, {0: unit(d + i, j), 1: unit(d + i, j)}))
    obj = {l: _herm_embed(states[l]) / 2 / 2 for l in range(2)}
    return dense_sdp([2 * d, 2 * d], cons, obj, maximize=True, names=["E1", "E2"])


def toy_z() -> ClusteredLowRankSDP:
    one = np.ones((1, 1))
    return dense_sdp([1, 1], [(1.0, {0: one, 1: one})], {0: one}, maximize=True, names=["z", "z2"])


def toy_z_as_free() -> ClusteredLowRankSDP:
    """`model_psd_variables_as_free_variables(problem, [:z])`: z becomes the free variable (z,1,1), a new 1 x 1 block Block(z,1,1) is set
    equal to it by the constraint <1, Z> - y = 0, and carries the objective: two clusters ({z2}, {z}), N = 1."""
    one = np.ones((1, 1))
    cons = [(1.0, {0: one}, [1.0]), (0.0, {1: one}, [-1.0])]
    return dense_sdp([1, 1], cons, {1: one}, maximize=True, free=["z"], b=[0.0], clusters=[[0], [1]], names=["z2", ("z", 1, 1)])
