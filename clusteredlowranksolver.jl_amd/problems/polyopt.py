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
