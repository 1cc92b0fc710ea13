"""Cohn-Elkies linear-programming bound for sphere packing and its N-radii generalisation
(reference examples/SpherePacking.jl:13-115 `Nsphere_packing`, :117-185 `cohnelkies`).

f(x) = sum_k y_k k!/pi^k L_k^{n/2-1}(pi |x|^2),  f^(t) = sum_k y_k t^k (t = |xi|^2):
    f^(t)  = <S21, b b^T> + t <S22, b b^T>                     t >= 0
    -f(w)  = <S31, .> + (w - r^2) <S32, b b^T>                 w >= r^2
    minimise vol(B(r/2)) f(0),  f^(0) = y_0 = 1.
b = the first d+1 elements of a Laguerre basis orthogonalised on 2d+2 rescaled-Laguerre samples.
"""
from __future__ import annotations

import numpy as np
import mpmath as mp

from ..sdp import Block, ClusteredLowRankSDP, HiLo, LowRankMat
from .polytools import (DEFAULT_PREC, approximate_fekete, laguerre_coefficients, laguerre_value,
                        polyval, sample_points_rescaled_laguerre, orthonormalize_free_basis)


def spherevolume(n, r):
    return mp.sqrt(mp.pi) ** n / mp.gamma(mp.mpf(n) / 2 + 1) * mp.mpf(r) ** n


def _normalised_laguerre_values(n, d, xs):
    """Values at xs of q_k / max(coefficients(q_k)), q_k = L_k^{n/2-1}(2 pi x), k = 0..2d+1
    (reference examples/SpherePacking.jl:124-126)."""
    polys = laguerre_coefficients(2 * d + 1, mp.mpf(n) / 2 - 1, 2 * mp.pi)
    V = np.empty((len(xs), len(polys)), dtype=object)
    for k, pc in enumerate(polys):
        mc = max(pc)
        for i, x in enumerate(xs):
            V[i, k] = polyval(pc, x) / mc
    return V


def _rank1(lam, vec):
    return LowRankMat(HiLo.of(np.array([lam], dtype=object)), HiLo.of(vec.reshape(1, -1)), HiLo.of(vec.reshape(1, -1)))


def _fhat_cluster(d, V, xs, names):
    """f^(x) = <S21, bb^T> + x <S22, bb^T> sampled on xs: two rank-1 blocks of side d+1."""
    ns = len(xs)
    e21 = {p: _rank1(mp.mpf(1), V[p, :d + 1]) for p in range(ns)}
    e22 = {p: _rank1(xs[p], V[p, :d + 1]) for p in range(ns)}
    return [Block(1, d + 1, {(0, 0): e21}, names[0]), Block(1, d + 1, {(0, 0): e22}, names[1])]


def cohnelkies_multi(n, d, radii, prec=DEFAULT_PREC, orth_free=False) -> ClusteredLowRankSDP:
    """Cohn-Elkies bound with one sign-constraint cluster per radius in `radii` (the bound is that of
    min(radii); the extra clusters are valid, redundant constraints).  `radii=[1]` is exactly
    `cohnelkies(n, d)`.  1 + len(radii) clusters of P = 2d+2 constraints, N = 2d+1 free variables."""
    with mp.workprec(prec):
        alpha = mp.mpf(n) / 2 - 1
        K = 2 * d + 1
        base = sample_points_rescaled_laguerre(K)
        V1, xs1 = approximate_fekete(_normalised_laguerre_values(n, d, base), base)
        ns = len(xs1)
        blocks = [_fhat_cluster(d, V1, xs1, ("SOS21", "SOS22"))]
        B1 = np.empty((ns, K), dtype=object)
        for p in range(ns):
            for k in range(1, K + 1):
                B1[p, k - 1] = -xs1[p] ** k
        Bs, cs = [B1], [HiLo.of(np.ones(ns))]
        Cs = [[np.zeros((d + 1, d + 1)), np.zeros((d + 1, d + 1))]]
        fact = [mp.factorial(k) / mp.pi ** k for k in range(K + 1)]
        for ri, r in enumerate(radii):
            r = mp.mpf(r)
            sh = [x + r ** 2 for x in base]
            V2, xs2 = approximate_fekete(_normalised_laguerre_values(n, d, sh), sh)
            e31 = {p: HiLo.of(np.array([[V2[p, 0] ** 2]], dtype=object)) for p in range(ns)}
            e32 = {p: _rank1(xs2[p] - r ** 2, V2[p, :d + 1]) for p in range(ns)}
            blocks.append([Block(1, 1, {(0, 0): e31}, ("SOS31", ri)), Block(1, d + 1, {(0, 0): e32}, ("SOS32", ri))])
            B2 = np.empty((ns, K), dtype=object)
            for p in range(ns):
                for k in range(1, K + 1):
                    B2[p, k - 1] = fact[k] * laguerre_value(k, alpha, mp.pi * xs2[p])
            Bs.append(B2)
            cs.append(HiLo.of(-np.ones(ns)))
            Cs.append([np.zeros((1, 1)), np.zeros((d + 1, d + 1))])
        rmin = min(mp.mpf(r) for r in radii)
        vol = spherevolume(n, rmin / 2)
        b = np.array([vol * fact[k] * laguerre_value(k, alpha, 0) for k in range(1, K + 1)], dtype=object)
        if orth_free:
            Bs, b, _ = orthonormalize_free_basis(Bs, b)
        return ClusteredLowRankSDP(maximize=False, constant=float(vol), blocks=blocks, B=[HiLo.of(x) for x in Bs],
                                   c=cs, C=Cs, b=HiLo.of(b),
                                   names={"free": list(range(1, K + 1)), "constant_mp": vol, "orth_free": orth_free})


def cohnelkies(n, d, r=1, prec=DEFAULT_PREC, orth_free=False) -> ClusteredLowRankSDP:
    """BASELINE config 3 (reference examples/SpherePacking.jl:117-185): 2 clusters, P = 32 each for d = 15.
    `orth_free=True` applies the (mathematically neutral) orthonormalising change of free variables of
    `polytools.orthonormalize_free_basis`, which makes the instance solvable in fp64."""
    return cohnelkies_multi(n, d, [r], prec=prec, orth_free=orth_free)


def nsphere_packing(n, d, radii, prec=DEFAULT_PREC) -> ClusteredLowRankSDP:
    """N-radii sphere packing bound (reference examples/SpherePacking.jl:13-115).
    Clusters: 1 (PSD1, m = N, delta = 1) + 1 (SOS21/SOS22, m = N, delta = d+1, P = (2d+2) N(N+1)/2)
    + N(N+1)/2 (SOS31 1x1 + SOS32) + N (slack).  N_free = (2d+2) N(N+1)/2 + 1."""
    with mp.workprec(prec):
        N = len(radii)
        rad = [mp.mpf(r) for r in radii]
        alpha = mp.mpf(n) / 2 - 1
        K = 2 * d + 1
        pairs = [(i, j) for i in range(N) for j in range(i + 1)]
        fidx = {}
        for (i, j) in pairs:
            for k in range(K + 1):
                fidx[(k, i, j)] = len(fidx)
        fidx["M"] = len(fidx)
        NF = len(fidx)
        base = sample_points_rescaled_laguerre(K)
        V, xs = approximate_fekete(_normalised_laguerre_values(n, d, base), base)
        ns = len(xs)
        fact = [mp.factorial(k) / mp.pi ** k for k in range(K + 1)]
        blocks, Bs, cs, Cs = [], [], [], []
        zero = mp.mpf(0)

        def zeros(r_, c_):
            a = np.empty((r_, c_), dtype=object); a[:, :] = zero
            return a

        # constraint 1: PSD1_ij - y_(0,i,j) = -sqrt(vol_i vol_j)
        one_v = np.array([mp.mpf(1)], dtype=object)
        ent = {}
        B = zeros(len(pairs), NF); c = np.empty(len(pairs), dtype=object)
        for p, (i, j) in enumerate(pairs):
            if i != j:
                ent.setdefault((i, j), {})[p] = _rank1(mp.mpf(1) / 2, one_v)
                ent.setdefault((j, i), {})[p] = _rank1(mp.mpf(1) / 2, one_v)
            else:
                ent.setdefault((i, i), {})[p] = _rank1(mp.mpf(1), one_v)
            B[p, fidx[(0, i, j)]] = mp.mpf(-1)
            c[p] = -mp.sqrt(spherevolume(n, rad[i]) * spherevolume(n, rad[j]))
        blocks.append([Block(N, 1, ent, "PSD1")]); Bs.append(HiLo.of(B)); cs.append(HiLo.of(c))
        Cs.append([np.zeros((N, N))])

        # constraint 2: sum_k y_(k,i,j) x^k = <S21_ij, bb^T> + x <S22_ij, bb^T>
        e21, e22 = {}, {}
        P2 = ns * len(pairs)
        B = zeros(P2, NF)
        p = 0
        for (i, j) in pairs:
            for t in range(ns):
                b = V[t, :d + 1]
                for (r_, s_) in ({(i, j), (j, i)}):
                    e21.setdefault((r_, s_), {})[p] = _rank1(mp.mpf(1), b)
                    e22.setdefault((r_, s_), {})[p] = _rank1(xs[t], b)
                for k in range(K + 1):
                    B[p, fidx[(k, i, j)]] = (-2 if i != j else -1) * xs[t] ** k
                p += 1
        blocks.append([Block(N, d + 1, e21, "SOS21"), Block(N, d + 1, e22, "SOS22")])
        Bs.append(HiLo.of(B)); cs.append(HiLo.of(np.zeros(P2)))
        Cs.append([np.zeros((N * (d + 1),) * 2), np.zeros((N * (d + 1),) * 2)])

        # constraint 3: S31 + (x - (r_i + r_j)^2) <S32, bb^T> + sum_k y_(k,i,j) k!/pi^k L_k(pi x) = 0
        for (i, j) in pairs:
            e31 = {t: _rank1(mp.mpf(1), V[t, :1]) for t in range(ns)}
            e32 = {t: _rank1(xs[t] - (rad[i] + rad[j]) ** 2, V[t, :d + 1]) for t in range(ns)}
            B = zeros(ns, NF)
            for t in range(ns):
                for k in range(K + 1):
                    B[t, fidx[(k, i, j)]] = fact[k] * laguerre_value(k, alpha, mp.pi * xs[t])
            blocks.append([Block(1, 1, {(0, 0): e31}, ("SOS31", i, j)), Block(1, d + 1, {(0, 0): e32}, ("SOS32", i, j))])
            Bs.append(HiLo.of(B)); cs.append(HiLo.of(np.zeros(ns)))
            Cs.append([np.zeros((1, 1)), np.zeros((d + 1, d + 1))])

        # constraint 4: M - sum_k y_(k,i,i) k!/pi^k L_k(0) = slack_i
        for i in range(N):
            B = zeros(1, NF)
            for k in range(K + 1):
                B[0, fidx[(k, i, i)]] = fact[k] * laguerre_value(k, alpha, 0)
            B[0, fidx["M"]] = mp.mpf(-1)
            blocks.append([Block(1, 1, {(0, 0): {0: HiLo.of(np.ones((1, 1)))}}, ("slack4", i))])
            Bs.append(HiLo.of(B)); cs.append(HiLo.of(np.zeros(1))); Cs.append([np.zeros((1, 1))])

        b = np.zeros(NF); b[fidx["M"]] = 1.0
        return ClusteredLowRankSDP(maximize=False, constant=0.0, blocks=blocks, B=Bs, c=cs, C=Cs, b=b,
                                   names={"free": fidx})
