"""Delsarte LP bound for spherical codes (reference examples/Delsarte.jl:7-49).

    minimise M  s.t.  sum_k a_k P_k^n(x) + <S1, b b^T> + (1+x)(cos t - x) <S2, b' b'^T> = -1   on samples
                      sum_k a_k + slack - M = -1,         a_k >= 0, slack >= 0
One cluster (all constraints share the a_k), P = 2d+2, N = 1; blocks: 2d dense 1x1 (a_k),
(d+1)x(d+1) and d x d rank-1 SOS blocks, one dense 1x1 slack.
"""
from __future__ import annotations

import numpy as np
import mpmath as mp

from ..sdp import Block, ClusteredLowRankSDP, HiLo, LowRankMat
from .polytools import (DEFAULT_PREC, approximate_fekete, chebyshev_values, gegenbauer_values,
                        sample_points_chebyshev)


def delsarte(n, d, costheta, prec=DEFAULT_PREC) -> ClusteredLowRankSDP:
    with mp.workprec(prec):
        ct = mp.mpf(costheta) if not isinstance(costheta, str) else mp.mpf(eval(costheta))
        xs = sample_points_chebyshev(2 * d, -1, ct)
        V, xs = approximate_fekete(chebyshev_values(2 * d, xs), xs)
        ns = len(xs)                      # 2d+1 samples
        P = ns + 1                        # + the scalar constraint
        G = gegenbauer_values(2 * d, n, xs)
        blocks, Cs = [], []
        one = np.ones((1, 1))
        for k in range(1, 2 * d + 1):     # a_k : dense 1x1, present in every constraint
            ent = {p: HiLo.of(np.array([[G[p, k]]], dtype=object)) for p in range(ns)}
            ent[ns] = HiLo.of(one)
            blocks.append(Block(m=1, delta=1, entries={(0, 0): ent}, name=("a", k)))
            Cs.append(np.zeros((1, 1)))
        e1 = {p: LowRankMat(np.array([1.0]), HiLo.of(V[p:p + 1, :d + 1]), HiLo.of(V[p:p + 1, :d + 1])) for p in range(ns)}
        blocks.append(Block(m=1, delta=d + 1, entries={(0, 0): e1}, name=("SOS", 1)))
        Cs.append(np.zeros((d + 1, d + 1)))
        e2 = {}
        for p in range(ns):
            lam = np.array([(1 + xs[p]) * (ct - xs[p])], dtype=object)
            e2[p] = LowRankMat(HiLo.of(lam), HiLo.of(V[p:p + 1, :d]), HiLo.of(V[p:p + 1, :d]))
        blocks.append(Block(m=1, delta=d, entries={(0, 0): e2}, name=("SOS", 2)))
        Cs.append(np.zeros((d, d)))
        blocks.append(Block(m=1, delta=1, entries={(0, 0): {ns: HiLo.of(one)}}, name="slack"))
        Cs.append(np.zeros((1, 1)))
        B = np.zeros((P, 1)); B[ns, 0] = -1.0
        c = -np.ones(P)
        return ClusteredLowRankSDP(maximize=False, constant=0.0, blocks=[blocks], B=[B], c=[c], C=[Cs],
                                   b=np.ones(1), names={"free": ["M"], "blocks": [[b.name for b in blocks]]})
