"""Polynomial bases, sample sets and approximate-Fekete orthogonalisation for the problem generators.

Written from the mathematics the reference documents, evaluated directly at sample points with
mpmath (no symbolic polynomial ring):
  bases        -- reference src/basesandsamples.jl:6-99   (three-term recurrences)
  sample sets  -- reference src/basesandsamples.jl:106-183
  Fekete       -- reference src/approximate_fekete.jl:51-80 (repeated fp64 QR, basis change in
                  high precision, column-pivoted QR to pick the points)
Everything returns mpmath numbers inside numpy object arrays.
"""
from __future__ import annotations

import numpy as np
import mpmath as mp
import scipy.linalg as sla

DEFAULT_PREC = 256  # bits, = precision(BigFloat) default of the reference (src/solver.jl:103)


def mpf_array(x):
    a = np.empty(np.shape(x), dtype=object)
    flat = a.reshape(-1)
    for i, v in enumerate(np.asarray(x, dtype=object).reshape(-1)):
        flat[i] = mp.mpf(v)
    return a


def to_float(a) -> np.ndarray:
    return np.array([[float(v) for v in row] for row in np.atleast_2d(a)], dtype=np.float64).reshape(np.shape(a))


def mp_matmul(A, B):
    """Object-array matrix product in the current mp precision."""
    A = np.asarray(A, dtype=object)
    B = np.asarray(B, dtype=object)
    n, k = A.shape
    k2, m = B.shape
    assert k == k2
    C = np.empty((n, m), dtype=object)
    for i in range(n):
        Ai = A[i]
        for j in range(m):
            C[i, j] = mp.fdot(Ai, B[:, j])
    return C


# ------------------------------------------------------------------------------------------
# sample sets
# ------------------------------------------------------------------------------------------

def sample_points_chebyshev(d, a=-1, b=1):
    """d+1 Chebyshev points in [a, b] (reference src/basesandsamples.jl:161-168)."""
    a, b = mp.mpf(a), mp.mpf(b)
    return [(a + b) / 2 + (b - a) / 2 * mp.cospi(mp.mpf(2 * k - 1) / (2 * (d + 1))) for k in range(1, d + 2)]


def sample_points_rescaled_laguerre(d):
    """'rescaled Laguerre' points of SDPB (reference src/basesandsamples.jl:146-155)."""
    const = -mp.sqrt(mp.pi) / (64 * (d + 1) * mp.log(3 - 2 * mp.sqrt(2)))
    return [const * (-1 + 4 * k) ** 2 for k in range(d + 1)]


# ------------------------------------------------------------------------------------------
# bases evaluated at points: value matrices V[i, k] = basis_k(x_i)
# ------------------------------------------------------------------------------------------

def chebyshev_values(d, xs):
    """T_0..T_d at xs (reference src/basesandsamples.jl:66-76)."""
    V = np.empty((len(xs), d + 1), dtype=object)
    for i, x in enumerate(xs):
        V[i, 0] = mp.mpf(1)
        if d >= 1:
            V[i, 1] = mp.mpf(x)
        for l in range(2, d + 1):
            V[i, l] = 2 * x * V[i, l - 1] - V[i, l - 2]
    return V


def gegenbauer_values(d, n, xs):
    """Gegenbauer polynomials for S^{n-1}, normalised to 1 at 1 (reference src/basesandsamples.jl:88-99)."""
    V = np.empty((len(xs), d + 1), dtype=object)
    for i, x in enumerate(xs):
        V[i, 0] = mp.mpf(1)
        if d >= 1:
            V[i, 1] = mp.mpf(x)
        for l in range(2, d + 1):
            V[i, l] = mp.mpf(2 * l + n - 4) / (l + n - 3) * x * V[i, l - 1] - mp.mpf(l - 1) / (l + n - 3) * V[i, l - 2]
    return V


def laguerre_coefficients(d, alpha, scale=1):
    """Coefficient lists (ascending powers of x) of L_k^{alpha}(scale * x), k = 0..d
    (recurrence of reference src/basesandsamples.jl:33-43)."""
    alpha, scale = mp.mpf(alpha), mp.mpf(scale)

    def padd(p, q):
        n = max(len(p), len(q))
        return [(p[i] if i < len(p) else 0) + (q[i] if i < len(q) else 0) for i in range(n)]

    def pscale(p, s):
        return [s * c for c in p]

    def pmulx(p, s):  # p * (s x)
        return [mp.mpf(0)] + [s * c for c in p]

    polys = [[mp.mpf(1)]]
    if d >= 1:
        polys.append([1 + alpha, -scale])
    for l in range(2, d + 1):
        a = padd(pscale(polys[l - 1], 2 * l - 1 + alpha), pscale(pmulx(polys[l - 1], scale), -1))
        b = pscale(polys[l - 2], -(l + alpha - 1))
        polys.append(pscale(padd(a, b), mp.mpf(1) / l))
    return polys


def polyval(coeffs, x):
    acc = mp.mpf(0)
    for c in reversed(coeffs):
        acc = acc * x + c
    return acc


def laguerre_value(k, alpha, x):
    """L_k^{alpha}(x) by the three-term recurrence."""
    alpha, x = mp.mpf(alpha), mp.mpf(x)
    if k == 0:
        return mp.mpf(1)
    prev, cur = mp.mpf(1), 1 + alpha - x
    for l in range(2, k + 1):
        prev, cur = cur, ((2 * l - 1 + alpha - x) * cur - (l + alpha - 1) * prev) / l
    return cur


# ------------------------------------------------------------------------------------------
# approximate Fekete points / orthogonalised basis
# ------------------------------------------------------------------------------------------

def approximate_fekete(V, samples, s=3):
    """Given V[i, k] = basis_k(sample_i), return (V', samples') where V' holds the values of a new
    basis (an upper-triangular change of the old one, so degree ordering is preserved) that is
    numerically orthonormal on the selected samples; as many samples as basis elements are kept,
    sorted increasingly.  Method of reference src/approximate_fekete.jl:76-106."""
    V = np.array(V, dtype=object)
    nb = V.shape[1]
    for _ in range(s):
        R = np.linalg.qr(to_float(V), mode="r")
        U = sla.solve_triangular(R, np.eye(nb), lower=False)
        V = mp_matmul(V, mpf_array(np.triu(U)))
    if V.shape[0] > nb:
        _, _, piv = sla.qr(to_float(V).T, pivoting=True, mode="economic")
        idx = list(piv[:nb])
    else:
        idx = list(range(V.shape[0]))
    V = V[idx, :]
    R = np.linalg.qr(to_float(V), mode="r")
    U = sla.solve_triangular(R, np.eye(nb), lower=False)
    V = mp_matmul(V, mpf_array(np.triu(U)))
    sel = [samples[i] for i in idx]
    order = sorted(range(len(sel)), key=lambda i: sel[i])
    return V[order, :], [sel[i] for i in order]


def orthonormalize_free_basis(Bs, b):
    """Change of free variables y = R^-1 y' that makes the stacked B = [B_1; B_2; ...] have orthonormal
    columns: returns (Bs', b') with B_j' = B_j R^-1, b' = R^-T b (B = Q R by modified Gram-Schmidt with
    re-orthogonalisation, in the current mp precision).  The SDP is mathematically unchanged
    (<b, y> = <b', y'>), but the normal-equation matrix Q = B^T S^-1 B becomes well conditioned enough
    for fp64 -- the monomial / Laguerre free-variable bases of the sphere-packing examples give
    cond(B) ~ 1e30, which is why the reference needs 256-bit Arb there."""
    Bst = np.vstack([np.asarray(B, dtype=object) for B in Bs])
    rows, N = Bst.shape
    Q = Bst.copy()
    R = np.empty((N, N), dtype=object)
    R[:, :] = mp.mpf(0)
    for k in range(N):
        for _ in range(2):
            for i in range(k):
                rik = mp.fdot(Q[:, i], Q[:, k])
                R[i, k] += rik
                Q[:, k] = Q[:, k] - rik * Q[:, i]
        nrm = mp.sqrt(mp.fdot(Q[:, k], Q[:, k]))
        R[k, k] = nrm
        Q[:, k] = Q[:, k] / nrm
    # b' = R^-T b  (forward substitution with R^T lower triangular)
    bp = np.empty(N, dtype=object)
    for i in range(N):
        s = mp.mpf(b[i])
        for k in range(i):
            s -= R[k, i] * bp[k]
        bp[i] = s / R[i, i]
    out, r0 = [], 0
    for B in Bs:
        n = np.shape(B)[0]
        out.append(Q[r0:r0 + n, :])
        r0 += n
    return out, bp, R
