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
    if name == "polyopt80":     # one 41 x 41 block with 81 unique vectors: beyond the shapes of k_mws_pair, two chunks of k in k_mwx_gram
        return P.polyopt_random(40, seed=1)[0]
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
    if name == "ns_8_15_3":     # Nsphere_packing(8, 15, [1/2, 1/2, 1/2]): 11 clusters, P = 192, 193 free variables (BASELINE config 3's many-cluster instance)
        return P.nsphere_packing(8, 15, [0.5, 0.5, 0.5])
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
    if name == "threepoint_3_8_8":      # BASELINE config 4 as named: "ThreePointBound n=3, 2d=16" (examples/ThreePointBound.jl:45-160 with d2 = d3 = 8)
        import mpmath as mp
        return P.three_point_spherical_codes(3, mp.mpf(1) / 2, 8, 8)
    if name == "sdpa_x64":              # BASELINE config 5 as named: SDPA dense-constraint import scaled to 64 blocks
        return P.sdpa_to_sdp(P.sdpa_scaled(nb=64, bs=32, m=256, seed=64))
    if name == "sdpa_x64_full":         # the same with every constraint matrix full block diagonal (as in test/example.dat-s): 7.5 GFLOP per assembly
        return P.sdpa_to_sdp(P.sdpa_scaled(nb=64, bs=32, m=256, seed=64, blocks_per_constraint=64))
    if name == "polyopt_scaled_300":      # n = 301, P = 601: beyond one outer block (256) of the staged TRSM / Cholesky plans
        return P.polyopt_scaled(300)
    if name == "polyopt_scaled_100":
        return P.polyopt_scaled(100)
    raise KeyError(name)


@functools.lru_cache(maxsize=None)
def flat(name):
    return clrs_amd.flatten(instance(name))


def permute_cluster_constraints(f, seed=0):
    """The same SDP with the constraints of every cluster relabelled by a random permutation (rows of B_j and c_j moved along):
    the vector -> constraint map of the cluster-per-wave assembly kernels is then not the identity."""
    import copy
    rng = np.random.default_rng(seed)
    g = copy.copy(f)
    g.term_p, g.dense_p = f.term_p.copy(), f.dense_p.copy()
    g.B, g.B_lo, g.c, g.c_lo = f.B.copy(), f.B_lo.copy(), f.c.copy(), f.c_lo.copy()
    N = f.n_free
    for j in range(f.n_clusters):
        P = int(f.cluster_P[j])
        perm = rng.permutation(P)                    # new index of old constraint p
        for b in range(f.n_blocks):
            if int(f.block_cluster[b]) != j:
                continue
            t0, t1 = int(f.term_ptr[b]), int(f.term_ptr[b + 1])
            g.term_p[t0:t1] = perm[f.term_p[t0:t1]]
            d0, d1 = int(f.dense_ptr[b]), int(f.dense_ptr[b + 1])
            g.dense_p[d0:d1] = perm[f.dense_p[d0:d1]]
        o = int(f.cluster_off[j])
        for arr_new, arr_old in ((g.c, f.c), (g.c_lo, f.c_lo)):
            arr_new[o + perm] = arr_old[o:o + P]
        for arr_new, arr_old in ((g.B, f.B), (g.B_lo, f.B_lo)):
            Bo = arr_old[o * N:(o + P) * N].reshape(P, N, order="F")
            Bn = np.empty_like(Bo)
            Bn[perm, :] = Bo
            arr_new[o * N:(o + P) * N] = Bn.reshape(-1, order="F")
    return g


def duplicate_block(f, b, scale=0.5):
    """The same SDP with a copy of PSD block b (its constraint matrices scaled) appended right after it in the same cluster:
    gives a cluster a second 1 x 1 dense block, or one more low-rank block."""
    import copy
    g = copy.copy(f)
    ins = b + 1

    def ins1(a, val):
        return np.insert(a, ins, val)

    g.n_blocks = f.n_blocks + 1
    for name in ("block_cluster", "block_m", "block_delta", "block_kind", "block_n"):
        setattr(g, name, ins1(getattr(f, name), getattr(f, name)[b]))
    t0, t1 = int(f.term_ptr[b]), int(f.term_ptr[b + 1])
    d0, d1 = int(f.dense_ptr[b]), int(f.dense_ptr[b + 1])
    nt, nd = t1 - t0, d1 - d0
    g.term_ptr = np.concatenate([f.term_ptr[:ins + 1], f.term_ptr[ins:] + nt]).astype(np.int64)
    g.dense_ptr = np.concatenate([f.dense_ptr[:ins + 1], f.dense_ptr[ins:] + nd]).astype(np.int64)
    for name in ("term_p", "term_r", "term_s", "term_rank"):
        a = getattr(f, name)
        setattr(g, name, np.concatenate([a[:t1], a[t0:t1], a[t1:]]))
    for name, s_ in (("term_lambda", scale), ("term_lambda_lo", scale)):
        a = getattr(f, name)
        setattr(g, name, np.concatenate([a[:t1], s_ * a[t0:t1], a[t1:]]))
    v0, v1 = int(f.term_vec_ptr[t0]), int(f.term_vec_ptr[t1])
    lens = np.diff(f.term_vec_ptr)
    g.term_vec_ptr = np.concatenate([[0], np.cumsum(np.concatenate([lens[:t1], lens[t0:t1], lens[t1:]]))]).astype(np.int64)
    for name in ("term_vs", "term_vs_lo", "term_ws", "term_ws_lo"):
        a = getattr(f, name)
        setattr(g, name, np.concatenate([a[:v1], a[v0:v1], a[v1:]]))
    g.dense_p = np.concatenate([f.dense_p[:d1], f.dense_p[d0:d1], f.dense_p[d1:]])
    a0, a1 = int(f.dense_A_ptr[d0]), int(f.dense_A_ptr[d1])
    dl = np.diff(f.dense_A_ptr)
    g.dense_A_ptr = np.concatenate([[0], np.cumsum(np.concatenate([dl[:d1], dl[d0:d1], dl[d1:]]))]).astype(np.int64)
    for name in ("dense_A", "dense_A_lo"):
        a = getattr(f, name)
        setattr(g, name, np.concatenate([a[:a1], scale * a[a0:a1], a[a1:]]))
    x0, x1 = int(f.block_off[b]), int(f.block_off[b + 1])
    for name in ("C", "C_lo"):
        a = getattr(f, name)
        setattr(g, name, np.concatenate([a[:x1], a[x0:x1], a[x1:]]))
    g.block_off = np.concatenate([[0], np.cumsum(g.block_n.astype(np.int64) ** 2)]).astype(np.int64)
    return g


def random_simple_sdp(seed, J=5, n_free=3, max_P=32, max_n=16, definite=False, fixed_P=None, lr_blocks=None):
    """A random SDP of the shapes the cluster-per-wave assembly takes: per cluster P_j constraints, 1-3 low-rank blocks of side
    n <= 16 with ONE rank-1 symmetric term per constraint (distinct vectors, the same constraint order in every block) and 0-2 dense
    1 x 1 blocks that touch a random subset of the constraints; mixed sizes within one context.  `definite`: the first block of every
    cluster has side >= 9 (45 independent pairings >= P), so that S_j is positive definite at generic iterates."""
    from clrs_amd.sdp import Block, ClusteredLowRankSDP, HiLo, LowRankMat
    rng = np.random.default_rng(seed)
    blocks, B, c, C = [], [], [], []
    for j in range(J):
        P = int(rng.integers(1, max_P + 1)) if fixed_P is None else int(fixed_P)      # fixed_P / lr_blocks: every cluster P constraints and
        cl, Cl = [], []                                                                # lr_blocks low-rank blocks of side max_n (size-limit tests)
        for bi in range(int(rng.integers(1, 4)) if lr_blocks is None else int(lr_blocks)):
            n = int(rng.integers(1, max_n + 1)) if lr_blocks is None else int(max_n)
            if definite and bi == 0:
                n = max(n, 9)
            V = rng.standard_normal((P, n))
            lam = rng.uniform(0.5, 1.5, P) * rng.choice([-1.0, 1.0], P)
            ent = {(0, 0): {p: LowRankMat(np.array([lam[p]]), V[p:p + 1, :], V[p:p + 1, :]) for p in range(P)}}
            cl.append(Block(m=1, delta=n, entries=ent, name="lr"))
            Cl.append(np.zeros((n, n)))
        for _ in range(int(rng.integers(0, 3))):
            ps = [p for p in range(P) if rng.random() < 0.6] or [0]
            ent = {(0, 0): {p: HiLo.of(rng.standard_normal((1, 1))) for p in ps}}
            cl.append(Block(m=1, delta=1, entries=ent, name="dense"))
            Cl.append(np.zeros((1, 1)))
        order = rng.permutation(len(cl))          # dense blocks anywhere among the low-rank ones
        blocks.append([cl[i] for i in order])
        C.append([Cl[i] for i in order])
        B.append(rng.standard_normal((P, n_free)))
        c.append(rng.standard_normal(P))
    sdp = ClusteredLowRankSDP(maximize=True, constant=0.0, blocks=blocks, B=B, c=c, C=C, b=rng.standard_normal(n_free), names={})
    sdp.check()
    return sdp


# ---- multi-word (planar limbs) helpers --------------------------------------------------------------------------------
def mw_from_double(a, K):
    """fp64 array -> planar limbs (K, len) with zero tails."""
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
    out = np.zeros((K, a.size))
    out[0] = a
    return out


def mw_with_tails(a, K, seed=0):
    """fp64 array -> planar limbs whose lower limbs are random but properly nested (|limb l+1| <= ulp(limb l) / 2): numbers that
    genuinely carry 53 K bits."""
    rng = np.random.default_rng(seed)
    out = mw_from_double(a, K)
    for l in range(1, K):
        prev = out[l - 1]
        t = 0.49 * np.spacing(np.abs(prev)) * rng.uniform(-1, 1, prev.shape)
        out[l] = np.where(prev != 0, t, 0.0)
    return out


def mw_diff(a, b):
    """Exact elementwise difference of two planar-limb arrays (possibly of different limb counts), rounded to fp64."""
    import math
    a, b = np.atleast_2d(a), np.atleast_2d(b)
    assert a.shape[1] == b.shape[1]
    out = np.empty(a.shape[1])
    for i in range(a.shape[1]):
        out[i] = math.fsum(list(a[:, i]) + list(-b[:, i]))
    return out


def mw_relerr(a, b, scale=None):
    """max |a - b| / max |b| (or / scale), exact differences."""
    d = np.abs(mw_diff(a, b))
    s = np.max(np.abs(np.atleast_2d(b)[0])) if scale is None else scale
    return float(np.max(d) / s) if d.size else 0.0


# ---- FlatSDP <-> .npz (fixtures of generated instances) ------------------------------------------------------------------
def save_flat(path, f, **extra):
    import dataclasses
    arrs = {k: np.asarray(getattr(f, k)) for k in (fl.name for fl in dataclasses.fields(f)) if k != "tails"}      # (fixtures carry two limb planes)
    np.savez_compressed(path, **arrs, **extra)


def load_flat(path):
    import dataclasses
    z = np.load(path, allow_pickle=False)
    kw = {}
    for fl in dataclasses.fields(sdpmod.FlatSDP):
        if fl.name not in z.files and fl.default_factory is not dataclasses.MISSING:      # (`tails`: fixtures carry two limb planes)
            continue
        v = z[fl.name]
        kw[fl.name] = v if v.ndim else v.item()
    return sdpmod.FlatSDP(**kw), {k: z[k] for k in z.files if k not in kw}
