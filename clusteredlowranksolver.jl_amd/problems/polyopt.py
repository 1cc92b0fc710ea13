"""Univariate polynomial optimisation as an SOS problem (reference examples/PolyOpt.jl:7-30).

    maximise lambda  s.t.  f(x) - lambda = <Y, b(x) b(x)^T>   sampled at 2d+1 Chebyshev points,
with b = (T_0..T_d).  One cluster, P = 2d+1, one (d+1)x(d+1) rank-1 block, N = 1.
"""
from __future__ import annotations

import numpy as np
import mpmath as mp

from ..sdp import Block, ClusteredLowRankSDP, HiLo, LowRankMat
from .polytools import DEFAULT_PREC, chebyshev_values, sample_points_chebyshev


def polyopt(f, d, prec=DEFAULT_PREC) -> ClusteredLowRankSDP:
    """`f`: callable mp.mpf -> mp.mpf of degree <= 2d."""
    with mp.workprec(prec):
        xs = sample_points_chebyshev(2 * d, -1, 1)
        V = chebyshev_values(d, xs)
        P = len(xs)
        entries = {(0, 0): {p: LowRankMat(np.array([1.0]), HiLo.of(V[p:p + 1, :]), HiLo.of(V[p:p + 1, :])) for p in range(P)}}
        c = np.array([f(x) for x in xs], dtype=object)
        blk = Block(m=1, delta=d + 1, entries=entries, name="sos")
        return ClusteredLowRankSDP(
            maximize=True, constant=0.0, blocks=[[blk]],
            B=[HiLo.of(np.ones((P, 1)))], c=[HiLo.of(c)],
            C=[[HiLo.of(np.zeros((d + 1, d + 1)))]], b=HiLo.of(np.ones(1)),
            names={"free": ["lambda"], "blocks": [["sos"]]})


def chebyshev_series(coeffs):
    """f(x) = sum_k coeffs[k] T_k(x) (Clenshaw)."""
    cs = [mp.mpf(c) for c in coeffs]

    def f(x):
        b1 = b2 = mp.mpf(0)
        for c in reversed(cs[1:]):
            b1, b2 = 2 * x * b1 - b2 + c, b1
        return x * b1 - b2 + cs[0]
    return f


def polyopt_random(d=20, seed=0, prec=DEFAULT_PREC):
    """BASELINE config 2: random degree-2d polynomial with U(-1,1) Chebyshev coefficients and a
    positive leading coefficient (so that it is bounded below).  Returns (sdp, coeffs)."""
    rng = np.random.default_rng(seed)
    coeffs = rng.uniform(-1.0, 1.0, size=2 * d + 1)
    coeffs[-1] = abs(coeffs[-1]) + 0.5
    return polyopt(chebyshev_series(coeffs), d, prec=prec), coeffs


def polyopt_scaled(d, seed=0) -> ClusteredLowRankSDP:
    """Roofline instance 'R' (SURVEY section 8d): config-2 structure at large d, built in fp64 only
    (n = d+1, P = 2d+1).  Not meant to be solved, only to drive the hot path at scale."""
    k = np.arange(1, 2 * d + 2)
    xs = np.cos(np.pi * (2 * k - 1) / (2 * (2 * d + 1)))
    V = np.cos(np.outer(np.arccos(xs), np.arange(d + 1)))  # T_k(x) = cos(k arccos x)
    P = xs.shape[0]
    rng = np.random.default_rng(seed)
    entries = {(0, 0): {p: LowRankMat(np.array([1.0]), V[p:p + 1, :], V[p:p + 1, :]) for p in range(P)}}
    blk = Block(m=1, delta=d + 1, entries=entries, name="sos")
    return ClusteredLowRankSDP(
        maximize=True, constant=0.0, blocks=[[blk]], B=[np.ones((P, 1))], c=[rng.uniform(-1, 1, P)],
        C=[[np.zeros((d + 1, d + 1))]], b=np.ones(1), names={"free": ["lambda"], "blocks": [["sos"]]})


def invariant_basis(d):
    """Exponents (a, i, j) of the S_3-invariant monomials (x+y+z)^a (xy+yz+zx)^i (xyz)^j of degree <= d, ordered by degree
    (reference examples/PolyOpt.jl:33-37: `for deg = 0:d for j = 0:div(deg,3) for i = 0:div(deg-3j,2)`)."""
    return [(deg - 2 * i - 3 * j, i, j) for deg in range(d + 1) for j in range(deg // 3 + 1) for i in range((deg - 3 * j) // 2 + 1)]


def min_f(d=2, prec=DEFAULT_PREC) -> ClusteredLowRankSDP:
    """S_3-invariant polynomial optimisation (reference examples/PolyOpt.jl:40-86, docs/src/examples/poly_opt.md):

        maximise M  s.t.  f - M = <Y_1, w w^T> + <Y_2, Pi_2 (x) w w^T> + <Y_3, Pi_3 (x) w w^T>,
        f = x^4 + y^4 + z^4 - 4xyz + x + y + z,

    sampled at the approximate Fekete points of the (2d+1)(2d+2)(2d+3) Chebyshev grid for the invariant polynomials of degree
    <= 2d.  `min_f(2)`: one cluster, P = 11, one 4x4 rank-1 block and one 3x3 rank-2 block (Pi_3 = v_1 v_1^T/2 + 3 v_2 v_2^T/2),
    N = 1.  The block of Pi_2 = ((x-y)(y-z)(z-x))^2 appears from d = 3 on."""
    from .polytools import approximate_fekete
    with mp.workprec(prec):
        inv = invariant_basis(2 * d)
        degrees = [a + 2 * i + 3 * j for (a, i, j) in inv]

        def ev(e, x, y, z):
            a, i, j = e
            return (x + y + z) ** a * (x * y + y * z + z * x) ** i * (x * y * z) ** j

        cheb = [sample_points_chebyshev(2 * d + k) for k in range(3)]
        grid = [(cheb[0][i], cheb[1][j], cheb[2][k]) for i in range(2 * d + 1) for j in range(2 * d + 2) for k in range(2 * d + 3)]
        V0 = np.array([[ev(e, *pt) for e in inv] for pt in grid], dtype=object)
        V, samples = approximate_fekete(V0, grid)       # V[p, k] = (new basis element k)(sample p); degree order preserved
        P = len(samples)
        equivariants = [
            [[(0, lambda x, y, z: mp.mpf(1))]],
            [[(3, lambda x, y, z: (x - y) * (y - z) * (z - x))]],
            [[(1, lambda x, y, z: 2 * x - y - z), (2, lambda x, y, z: 2 * y * z - x * z - x * y)],
             [(1, lambda x, y, z: y - z), (2, lambda x, y, z: x * z - x * y)]],
        ]
        factors = [[mp.mpf(1)], [mp.mpf(1)], [mp.mpf(1) / 2, mp.mpf(3) / 2]]
        blocks, Cs, names = [], [], []
        for eqi, rows in enumerate(equivariants):
            sel = []        # per rank-one term: list of (equivariant, basis index)
            for row in rows:
                items = [(ef, k) for (edeg, ef) in row for k, qdeg in enumerate(degrees) if 2 * edeg + 2 * qdeg <= 2 * d]
                if items:
                    sel.append(items)
            if not sel:
                continue
            side = len(sel[0])
            ent = {}
            for p, pt in enumerate(samples):
                vs = np.array([[ef(*pt) * V[p, k] for (ef, k) in row] for row in sel], dtype=object)
                lam = np.array(factors[eqi][:len(sel)], dtype=object)
                ent[p] = LowRankMat(HiLo.of(lam), HiLo.of(vs), HiLo.of(vs))
            blocks.append(Block(m=1, delta=side, entries={(0, 0): ent}, name=("trivariatesos", eqi + 1)))
            Cs.append(HiLo.of(np.zeros((side, side))))
            names.append(("trivariatesos", eqi + 1))
        c = np.array([x ** 4 + y ** 4 + z ** 4 - 4 * x * y * z + x + y + z for (x, y, z) in samples], dtype=object)
        return ClusteredLowRankSDP(
            maximize=True, constant=0.0, blocks=[blocks], B=[HiLo.of(np.ones((P, 1)))], c=[HiLo.of(c)], C=[Cs], b=HiLo.of(np.ones(1)),
            names={"free": ["M"], "blocks": [names], "samples": samples})
