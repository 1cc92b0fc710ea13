#!/usr/bin/env python3
"""Generates tests/golden/*.npz: 256-bit known answers for the hot path on small instances.

The reference (Julia + Arb) cannot run in the build container, so these vectors are an
INDEPENDENT multi-precision restatement of the mathematics the reference computes at its default
precision of 256 bits (src/solver.jl:73,103), written with mpmath and deliberately NOT sharing the
algorithm of the HIP path or of oracle/clrs_oracle.c:

  * S[j][p,q] = sum_l <A_p, X^-1 A_q Y> with DENSE A_p = Matrix(::LowRankMat) (src/interface.jl:798-800),
    X^-1 from an LU inverse -- the definition (src/solver.jl:1062-1226 computes the same numbers through
    bilinear pairings);
  * (dx, dy) from one LU solve of the full KKT matrix [S -B; B^T 0] (src/solver.jl:1527) -- no Schur
    complement, no Cholesky.

Inputs are the fp64-rounded problem data (what the C ABI receives), seeded fp64 iterates and right-hand
sides; outputs are rounded to fp64 once at the end.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import mpmath as mp
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.util import chol_blocks_np, flat, spd_iterates  # noqa: E402

PREC = 256
CASES = ["x2p1", "polyopt8", "delsarte_8_3", "ce_8_3", "ns_8_3_2", "sdpa_small", "polyopt40", "delsarte_3_10", "threepoint_4", "ce_8_15", "ns_8_15_2"]
# 2d = 30 sphere packing: S is numerically singular in fp64 (cond > 1/eps), so only the assembly is pinned there
S_ONLY = {"ce_8_15", "ns_8_15_2"}


def fingerprint(f):
    """Cheap content hash of the generated problem, so that a drifting generator is noticed."""
    parts = [f.cluster_P, f.block_m, f.block_delta, f.block_kind, f.term_p, f.term_r, f.term_s]
    h = sum(int(np.sum(np.asarray(a, dtype=np.int64) * (np.arange(len(a)) + 1))) for a in parts)
    v = float(np.sum(f.term_vs)) + float(np.sum(f.term_lambda)) + float(np.sum(f.dense_A)) + float(np.sum(f.B))
    return np.array([h, v], dtype=np.float64)


def dense_constraint_matrices(f, b):
    """{p: n x n mp.matrix} for block b."""
    n, m, dl = int(f.block_n[b]), int(f.block_m[b]), int(f.block_delta[b])
    out = {}
    if f.block_kind[b] == 0:
        for t in range(int(f.term_ptr[b]), int(f.term_ptr[b + 1])):
            p, r, s, lam = int(f.term_p[t]), int(f.term_r[t]), int(f.term_s[t]), mp.mpf(float(f.term_lambda[t]))
            v0 = int(f.term_vec_ptr[t])
            A = out.setdefault(p, mp.zeros(n, n))
            for i in range(dl):
                vi = mp.mpf(float(f.term_vs[v0 + i]))
                for k in range(dl):
                    A[r * dl + i, s * dl + k] += lam * vi * mp.mpf(float(f.term_ws[v0 + k]))
    else:
        for e in range(int(f.dense_ptr[b]), int(f.dense_ptr[b + 1])):
            a0 = int(f.dense_A_ptr[e])
            A = mp.zeros(n, n)
            for col in range(n):
                for row in range(n):
                    A[row, col] = mp.mpf(float(f.dense_A[a0 + row + col * n]))
            out[int(f.dense_p[e])] = A
    return out


def golden(name):
    f = flat(name)
    X, Y = spd_iterates(f, seed=11)
    Lc = chol_blocks_np(f, X)
    rng = np.random.default_rng(12)
    rhs_x, rhs_y = rng.standard_normal(f.x_len), rng.standard_normal(f.n_free)
    mp.mp.prec = PREC
    J, N = f.n_clusters, f.n_free
    S = [mp.zeros(int(P), int(P)) for P in f.cluster_P]
    for b in range(f.n_blocks):
        n, j = int(f.block_n[b]), int(f.block_cluster[b])
        o = int(f.block_off[b])
        Lm = mp.matrix(n, n)
        Ym = mp.matrix(n, n)
        for col in range(n):
            for row in range(n):
                Lm[row, col] = mp.mpf(float(Lc[o + row + col * n])) if row >= col else mp.mpf(0)
                Ym[row, col] = mp.mpf(float(Y[o + row + col * n]))
        Xinv = mp.inverse(Lm * Lm.T)
        As = dense_constraint_matrices(f, b)
        T = {q: Xinv * A * Ym for q, A in As.items()}
        for p, Ap in As.items():
            for q, Tq in T.items():
                acc = mp.mpf(0)
                for i in range(n):
                    for k in range(n):
                        acc += Ap[i, k] * Tq[i, k]
                S[j][p, q] += acc
    nx = f.x_len
    S_flat = np.concatenate([np.array([[float(S[j][p, q]) for p in range(int(f.cluster_P[j]))] for q in range(int(f.cluster_P[j]))]).reshape(-1)
                             for j in range(J)])   # column-major: element (p,q) at p + q*P
    if name in S_ONLY:
        return dict(fingerprint=fingerprint(f), Xchol=Lc, Y=Y, S=S_flat, prec=np.array([PREC]))
    K = mp.zeros(nx + N, nx + N)
    for j in range(J):
        P, o = int(f.cluster_P[j]), int(f.cluster_off[j])
        for p in range(P):
            for q in range(P):
                K[o + p, o + q] = (S[j][p, q] + S[j][q, p]) / 2
            for k in range(N):
                bv = mp.mpf(float(f.B[o * N + p + k * P]))
                K[o + p, nx + k] = -bv
                K[nx + k, o + p] = bv
    rhs = mp.matrix([mp.mpf(float(v)) for v in np.concatenate([rhs_x, rhs_y])])
    sol = mp.lu_solve(K, rhs)
    # conditioning info for the tolerance of the solve comparison
    conds = [float(mp.norm(S[j], 2)) for j in range(J)]
    return dict(fingerprint=fingerprint(f), Xchol=Lc, Y=Y, rhs_x=rhs_x, rhs_y=rhs_y, S=S_flat,
                dx=np.array([float(sol[i]) for i in range(nx)]), dy=np.array([float(sol[nx + i]) for i in range(N)]),
                S_norms=np.array(conds), prec=np.array([PREC]))


if __name__ == "__main__":
    out = os.path.dirname(os.path.abspath(__file__))
    for name in (sys.argv[1:] or CASES):
        g = golden(name)
        np.savez_compressed(os.path.join(out, name + ".npz"), **g)
        print(name, {k: v.shape for k, v in g.items()})
