"""Shared helpers for the tests: seeded iterates, cached problem instances."""
import functools

import numpy as np

import clrs_amd
from clrs_amd import sdp as sdpmod


def spd_iterates(flat, seed=1, scale=1.0, shift=1.0):
    """X, Y = shift*I + G G^T / n per block (SURVEY section 8d 'synthetic X, Y'), xy layout."""
    rng = np.random.default_rng(seed)
    X, Y = np.zeros(flat.xy_len), np.zeros(flat.xy_len)
    for b in range(flat.n_blocks):
        n = int(flat.block_n[b])
        for M in (X, Y):
            G = rng.standard_normal((n, n))
            A = scale * (shift * np.eye(n) + G @ G.T / n)
            M[flat.block_off[b]:flat.block_off[b + 1]] = A.reshape(-1, order="F")
    return X, Y


def chol_blocks_np(flat, X):
    out = np.zeros(flat.xy_len)
    for b in range(flat.n_blocks):
        n = int(flat.block_n[b])
        sl = slice(int(flat.block_off[b]), int(flat.block_off[b + 1]))
        out[sl] = np.linalg.cholesky(X[sl].reshape(n, n, order="F")).reshape(-1, order="F")
    return out


@functools.lru_cache(maxsize=None)
def instance(name):
    from clrs_amd import problems as P
    if name == "x2p1":
        return P.polyopt(lambda x: x * x + 1, 1)
    if name == "polyopt40":
        return P.polyopt_random(20, seed=0)[0]
    if name == "polyopt8":
        return P.polyopt_random(4, seed=3)[0]
    if name == "delsarte_3_10":
        return P.delsarte(3, 10, 0.5)
    if name == "delsarte_8_3":
        return P.delsarte(8, 3, 0.5)
    if name == "ce_8_15":
        return P.cohnelkies(8, 15)
    if name == "ce_8_15_orth":
        return P.cohnelkies(8, 15, orth_free=True)
    if name == "ce_8_3":
        return P.cohnelkies(8, 3, orth_free=True)
    if name == "ns_8_15_2":
        return P.nsphere_packing(8, 15, [0.5, 0.5])
    if name == "ns_8_3_2":
        return P.nsphere_packing(8, 3, [0.5, 0.5])
    if name == "sdpa_small":
        return P.sdpa_to_sdp(P.sdpa_scaled(nb=4, bs=8, m=12, seed=5))
    if name == "sdpa_mid":
        return P.sdpa_to_sdp(P.sdpa_scaled(nb=8, bs=32, m=40, seed=6))
    if name == "sdpa_example":
        import os
        return P.sdpa_to_sdp(P.read_sdpa(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "example.dat-s")))
    if name == "threepoint_4":
        import mpmath as mp
        return P.three_point_spherical_codes(4, mp.mpf(1) / 6, -1, 4)
    if name == "polyopt_scaled_100":
        return P.polyopt_scaled(100)
    raise KeyError(name)


@functools.lru_cache(maxsize=None)
def flat(name):
    return clrs_amd.flatten(instance(name))
