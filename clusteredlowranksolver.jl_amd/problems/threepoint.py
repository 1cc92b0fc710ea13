"""Three-point (Bachoc-Vallentin) bound for spherical codes with the S_3 symmetry in (u, v, t)
(reference examples/ThreePointBound.jl:1-159, `three_point_spherical_codes(n, costheta, d2, d3)`).

One cluster (both polynomial constraints share the matrix variables F_k), N = 0 free variables (d2 = -1) or the
rank-1 variables a_k, P = (2 N2 + 1) + #invariant monomials of degree <= 2 N3 constraints, dense blocks F_k of
sides d3+1-k (matrix polynomials sampled entry-wise) and low-rank sum-of-squares blocks of rank 1 or 2.

Everything is evaluated numerically at the sample points (mpmath), which is what the reference's
`sampleevaluate` produces from its symbolic description.  The reference draws its trivariate sample subset with
Julia's `Random.seed!(1935)` stream, which cannot be reproduced here: this generator draws its own seeded subset and
checks that it is unisolvent for the invariant polynomials of degree <= 2 N3.
"""
from __future__ import annotations

import numpy as np
import mpmath as mp

from ..sdp import Block, ClusteredLowRankSDP, HiLo, LowRankMat
from .polytools import DEFAULT_PREC, chebyshev_values, sample_points_chebyshev


def _gegenbauer_coefficients(k, n):
    """Ascending coefficients of the degree-k Gegenbauer polynomial for S^{n-1}, P(1) = 1
    (recurrence of reference src/basesandsamples.jl:88-99)."""
    polys = [[mp.mpf(1)], [mp.mpf(0), mp.mpf(1)]]
    for l in range(2, k + 1):
        a = [mp.mpf(0)] + [mp.mpf(2 * l + n - 4) / (l + n - 3) * c for c in polys[l - 1]]
        b = [mp.mpf(l - 1) / (l + n - 3) * c for c in polys[l - 2]] + [mp.mpf(0), mp.mpf(0)]
        polys.append([x - y for x, y in zip(a, b)])
    return polys[k]


def _Q(coef, k, u, v, t):
    """Q_k^n(u, v, t) = sum_i c_i ((1-u^2)(1-v^2))^((k-i)/2) (t - uv)^i   (examples/ThreePointBound.jl:7-11)."""
    base = (1 - u * u) * (1 - v * v)
    return sum(coef[i] * base ** ((k - i) // 2) * (t - u * v) ** i for i in range(len(coef)) if coef[i] != 0)


def _Smat(n, k, d, u, v, t, coef):
    """S_k(u,v,t), side d-k+1 (examples/ThreePointBound.jl:13-18)."""
    def m(w):
        return np.array([w ** e for e in range(d - k + 1)], dtype=object)

    def sym(a, b):
        return np.outer(a, b) + np.outer(b, a)
    mu, mv, mt = m(u), m(v), m(t)
    mat = _Q(coef, k, u, v, t) * sym(mv, mu) + _Q(coef, k, t, u, v) * sym(mt, mu) + _Q(coef, k, t, v, u) * sym(mt, mv)
    return mat / mp.mpf(6)


def _floor4(x):
    return mp.floor(mp.mpf(10) ** 4 * x) / mp.mpf(10) ** 4


def three_point_spherical_codes(n, costheta, d2, d3, seed=1935, prec=DEFAULT_PREC) -> ClusteredLowRankSDP:
    with mp.workprec(prec):
        ct = mp.mpf(costheta)
        N2, N3 = max(d2, d3), d3
        gco = [_gegenbauer_coefficients(k, n - 1) for k in range(d3 + 1)]

        def pw(w):      # p(u, costheta) = (u + 1)(costheta - u)
            return (w + 1) * (ct - w)

        # ---- univariate constraint: sum_k <F_k, 3 S_k(w,w,1)> [+ a_k terms] + SOS = -1 ----
        s1 = [_floor4(x) for x in sample_points_chebyshev(2 * N2, -1, 1)]
        n1 = len(s1)
        T1 = chebyshev_values(2 * N2, s1)
        # ---- trivariate samples ----
        inv = [(deg, k, j) for deg in range(2 * N3 + 1) for k in range(deg // 3 + 1) for j in range((deg - 3 * k) // 2 + 1)]
        n3 = len(inv)
        cheb = [sample_points_chebyshev(2 * N3 + q, -1, 1) for q in range(3)]
        grid = [(cheb[0][i], cheb[1][j], cheb[2][k]) for i in range(2 * N3 + 1) for j in range(2 * N3 + 2) for k in range(2 * N3 + 3)]
        rng = np.random.default_rng(seed)
        for attempt in range(50):
            pick = sorted(rng.permutation(len(grid))[:n3])
            s3 = sorted([tuple(_floor4(x) for x in grid[i]) for i in pick])
            Em = [[(u + v + t) ** (deg - 3 * k - 2 * j) * (u * v + v * t + u * t) ** j * (u * v * t) ** k for (deg, k, j) in inv] for (u, v, t) in s3]
            E = np.array([[float(x) for x in row] for row in Em])
            if np.linalg.cond(E) < 1e13:      # unisolvent for the invariant polynomials of degree <= 2 N3
                break
            if n3 > 60:
                # the invariant monomials are too ill-conditioned for an fp64 test at this degree: unisolvence is decided by an LU
                # decomposition at the working precision (the reference takes the shuffled subset unchecked, examples/ThreePointBound.jl:101-105)
                try:
                    LU, _ = mp.mp.LU_decomp(mp.matrix(Em))
                    piv = [abs(LU[i, i]) for i in range(n3)]
                    if min(piv) > mp.mpf(2) ** (-prec // 2) * max(piv):
                        break
                except ZeroDivisionError:
                    pass
        else:
            raise RuntimeError("no unisolvent sample subset found")
        P = n1 + n3
        blocks, Cs = [], []

        # dense F_k blocks: present in both constraints
        for k in range(d3 + 1):
            ent = {}
            for p, w in enumerate(s1):
                ent[p] = HiLo.of(3 * _Smat(n, k, d3, w, w, mp.mpf(1), gco[k]))
            for q, (u, v, t) in enumerate(s3):
                ent[n1 + q] = HiLo.of(_Smat(n, k, d3, u, v, t, gco[k]))
            blocks.append(Block(m=1, delta=d3 - k + 1, entries={(0, 0): ent}, name=("F", k)))
            Cs.append(np.ones((d3 + 1, d3 + 1)) if k == 0 else np.zeros((d3 - k + 1,) * 2))
        # rank-1 variables a_k (only when d2 >= 0): Gegenbauer P_k^n(w) * [1][1]^T in the univariate constraint
        if d2 >= 0:
            from .polytools import gegenbauer_values
            G = gegenbauer_values(2 * d2, n, s1)
            for k in range(2 * d2 + 1):
                ent = {p: LowRankMat(HiLo.of(np.array([G[p, k]], dtype=object)), np.ones((1, 1)), np.ones((1, 1))) for p in range(n1)}
                blocks.append(Block(m=1, delta=1, entries={(0, 0): ent}, name=("a", k)))
                Cs.append(np.ones((1, 1)))
        # univariate SOS certificates
        if N2 >= 0:
            ent = {p: LowRankMat(np.array([1.0]), HiLo.of(T1[p:p + 1, :N2 + 1]), HiLo.of(T1[p:p + 1, :N2 + 1])) for p in range(n1)}
            blocks.append(Block(m=1, delta=N2 + 1, entries={(0, 0): ent}, name=("univariatesos", 1)))
            Cs.append(np.zeros((N2 + 1,) * 2))
        if N2 >= 1:
            ent = {p: LowRankMat(HiLo.of(np.array([pw(s1[p])], dtype=object)), HiLo.of(T1[p:p + 1, :N2]), HiLo.of(T1[p:p + 1, :N2])) for p in range(n1)}
            blocks.append(Block(m=1, delta=N2, entries={(0, 0): ent}, name=("univariatesos", 2)))
            Cs.append(np.zeros((N2,) * 2))

        # trivariate invariant SOS certificates (examples/ThreePointBound.jl:96-141)
        basis3 = [(deg, k, j) for deg in range(N3 + 1) for k in range(deg // 3 + 1) for j in range((deg - 3 * k) // 2 + 1)]

        def b3(idx, u, v, t):
            deg, k, j = basis3[idx]
            return (u + v + t) ** (deg - 3 * k - 2 * j) * (u * v + v * t + u * t) ** j * (u * v * t) ** k
        equivariants = [
            [[(0, lambda u, v, t: mp.mpf(1))]],
            [[(3, lambda u, v, t: (u - v) * (v - t) * (t - u))]],
            [[(1, lambda u, v, t: 2 * u - v - t), (2, lambda u, v, t: 2 * v * t - u * t - u * v)],
             [(1, lambda u, v, t: v - t), (2, lambda u, v, t: u * t - u * v)]],
        ]
        factors = [[mp.mpf(1)], [mp.mpf(1)], [mp.mpf(1) / 2, mp.mpf(3) / 2]]
        weights = [
            (0, lambda u, v, t: mp.mpf(1)),
            (2, lambda u, v, t: pw(u) + pw(v) + pw(t)),
            (4, lambda u, v, t: pw(u) * pw(v) + pw(v) * pw(t) + pw(t) * pw(u)),
            (6, lambda u, v, t: pw(u) * pw(v) * pw(t)),
            (3, lambda u, v, t: 2 * u * v * t + 1 - u * u - v * v - t * t),
        ]
        for wi, (wdeg, wf) in enumerate(weights):
            if wdeg > 2 * N3:
                continue
            for eqi, rows in enumerate(equivariants):
                sel = []      # per row r: list of (eq function, basis index)
                for row in rows:
                    items = [(ef, bi) for (edeg, ef) in row for bi, (qdeg, _, _) in enumerate(basis3) if wdeg + 2 * edeg + 2 * qdeg <= 2 * N3]
                    if items:
                        sel.append(items)
                if not sel:
                    continue
                side = len(sel[0])
                assert all(len(r_) == side for r_ in sel)
                ent = {}
                for q, (u, v, t) in enumerate(s3):
                    vs = np.array([[ef(u, v, t) * b3(bi, u, v, t) for (ef, bi) in row] for row in sel], dtype=object)
                    lam = np.array([wf(u, v, t) * factors[eqi][r_] for r_ in range(len(sel))], dtype=object)
                    ent[n1 + q] = LowRankMat(HiLo.of(lam), HiLo.of(vs), HiLo.of(vs))
                blocks.append(Block(m=1, delta=side, entries={(0, 0): ent}, name=("trivariatesos", wi + 1, eqi + 1)))
                Cs.append(np.zeros((side, side)))

        c = np.concatenate([-np.ones(n1), np.zeros(n3)])
        return ClusteredLowRankSDP(maximize=False, constant=1.0, blocks=[blocks], B=[np.zeros((P, 0))], c=[c], C=[Cs],
                                   b=np.zeros(0), names={"free": [], "blocks": [[b.name for b in blocks]],
                                                         "samples1d": s1, "samples3d": s3})
