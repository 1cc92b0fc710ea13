"""Numeric SDP container: the host-side mirror of the reference's input type for the hot path.

Mirrors `ClusteredLowRankSDP` (reference src/interface.jl:807-819) and `LowRankMat`
(src/interface.jl:759-800): per cluster j a list of PSD blocks l, each an m x m grid of
sub-blocks (r, s) holding, per constraint index p, either a low-rank matrix
sum_k lambda_k vs_k ws_k^T or a dense matrix; plus B[j] (P_j x N), c[j], C[j][l], b.

Numbers are kept as (hi, lo) pairs of float64 arrays: `hi` is the value rounded to fp64 -- what
the HIP path and the fp64 oracle consume -- and `hi + lo` carries ~106 bits for the quad-precision
oracle.  `lo` may be None (treated as zero).  Indices are 0-based everywhere in this package.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple, Union

import numpy as np

try:  # mpmath is only needed by the generators / conversions
    import mpmath as mp
except Exception:  # pragma: no cover
    mp = None


# Limb planes a HiLo keeps of mpmath input: 2 = (hi, lo), ~106 bits -- what every committed fixture carries; more (`with data_planes(10): ...` around
# a generator) = the sampled problem at the working precision of a K-limb solve, as the reference holds it (convert_to_prec, src/interface.jl:1078-1112):
# planes 3.. go to `HiLo.tail` and, through `flatten`, to `FlatSDP.tails`.
_DATA_PLANES = 2


class data_planes:
    """Context manager: mpmath numbers converted by `HiLo.of` inside it keep `n` fp64 limb planes (2 by default)."""

    def __init__(self, n: int):
        self.n = max(2, int(n))

    def __enter__(self):
        global _DATA_PLANES
        self.old, _DATA_PLANES = _DATA_PLANES, self.n
        return self

    def __exit__(self, *a):
        global _DATA_PLANES
        _DATA_PLANES = self.old


def split_planes(x, n: int):
    """Array-like of mpmath numbers (or floats) -> n float64 arrays whose sum is x to ~53 n bits: successive roundings to nearest."""
    a = np.asarray(x, dtype=object)
    out = [np.zeros(a.shape, dtype=np.float64) for _ in range(n)]
    it = np.nditer(a, flags=["multi_index", "refs_ok"])
    for _ in it:
        idx = it.multi_index
        v = a[idx]
        if isinstance(v, (float, int, np.floating, np.integer)):
            out[0][idx] = float(v)
            continue
        with mp.workprec(max(mp.mp.prec, 64 * n + 64)):
            r = mp.mpf(v)
            for l in range(n):
                h = float(r)
                out[l][idx] = h
                r = r - mp.mpf(h)
                if r == 0:
                    break
    return out


def split_hi_lo(x):
    """Split an array-like of mpmath numbers (or floats) into fp64 (hi, lo) with hi+lo ~ x."""
    a = np.asarray(x, dtype=object)
    hi = np.empty(a.shape, dtype=np.float64)
    lo = np.zeros(a.shape, dtype=np.float64)
    it = np.nditer(a, flags=["multi_index", "refs_ok"])
    for _ in it:
        idx = it.multi_index
        v = a[idx]
        if isinstance(v, (float, int, np.floating, np.integer)):
            hi[idx] = float(v)
        else:
            h = float(v)
            hi[idx] = h
            lo[idx] = float(v - mp.mpf(h))
    return hi, lo


@dataclass
class HiLo:
    """A float64 array plus an optional low-order correction (value = hi + lo [+ sum of `tail`: further limb planes, see `data_planes`])."""
    hi: np.ndarray
    lo: Optional[np.ndarray] = None
    tail: Optional[List[np.ndarray]] = None

    @staticmethod
    def of(x) -> "HiLo":
        if isinstance(x, HiLo):
            return x
        a = np.asarray(x)
        if a.dtype == object:
            if _DATA_PLANES > 2:
                pl = split_planes(a, _DATA_PLANES)
                return HiLo(pl[0], pl[1] if np.any(pl[1] != 0.0) else None, pl[2:])
            hi, lo = split_hi_lo(a)
            return HiLo(hi, lo if np.any(lo != 0.0) else None)
        return HiLo(np.asarray(a, dtype=np.float64), None)

    def plane(self, t: int) -> np.ndarray:
        """limb plane t (0 = hi, 1 = lo, 2.. = tail), zeros where this number has none"""
        if t == 0:
            return self.hi
        if t == 1:
            return self.lo_or_zero()
        return self.tail[t - 2] if self.tail is not None and t - 2 < len(self.tail) else np.zeros_like(self.hi)

    @property
    def shape(self):
        return self.hi.shape

    def lo_or_zero(self) -> np.ndarray:
        return np.zeros_like(self.hi) if self.lo is None else self.lo


@dataclass
class LowRankMat:
    """sum_k lam[k] * vs[k] ws[k]^T  (reference src/interface.jl:759-763, 798-800).

    `lam` has shape (rank,), `vs` and `ws` shape (rank, delta)."""
    lam: HiLo
    vs: HiLo
    ws: HiLo

    def __post_init__(self):
        self.lam, self.vs, self.ws = HiLo.of(self.lam), HiLo.of(self.vs), HiLo.of(self.ws)
        if self.vs.hi.ndim != 2 or self.vs.shape != self.ws.shape or self.lam.hi.shape[0] != self.vs.hi.shape[0]:
            raise ValueError("LowRankMat should have the same number of values as vectors")

    @property
    def rank(self) -> int:
        return self.lam.hi.shape[0]

    @property
    def delta(self) -> int:
        return self.vs.hi.shape[1]

    def dense(self) -> np.ndarray:
        """Matrix(::LowRankMat), fp64 (reference src/interface.jl:798-800)."""
        return np.einsum("k,ki,kj->ij", self.lam.hi, self.vs.hi, self.ws.hi)

    def transpose(self) -> "LowRankMat":
        return LowRankMat(self.lam, self.ws, self.vs)


Entry = Union[LowRankMat, HiLo]


@dataclass
class Block:
    """One PSD block (j, l): m x m sub-blocks of side delta; entries[(r, s)][p] is the constraint matrix.

    For a low-rank block the convention entries[(s, r)][p] == entries[(r, s)][p]^T must hold
    (reference src/solver.jl:1009).  A dense ("high rank") block always has m == 1
    (reference src/interface.jl:1001-1007: sub-blocks are contracted)."""
    m: int
    delta: int
    entries: Dict[Tuple[int, int], Dict[int, Entry]]
    name: object = None

    @property
    def n(self) -> int:
        return self.m * self.delta

    @property
    def high_rank(self) -> bool:
        """reference src/solver.jl:1000"""
        return any(not isinstance(e, LowRankMat) for d in self.entries.values() for e in d.values())


@dataclass
class ClusteredLowRankSDP:
    """Numeric clustered low-rank SDP (reference src/interface.jl:807-819).

    maximize/constant as in the reference; blocks[j][l]; B[j] is P_j x N; c[j] length P_j;
    C[j][l] is n x n; b length N."""
    maximize: bool
    constant: float
    blocks: List[List[Block]]
    B: List[HiLo]
    c: List[HiLo]
    C: List[List[HiLo]]
    b: HiLo
    names: dict = field(default_factory=dict)

    def __post_init__(self):
        self.B = [HiLo.of(x) for x in self.B]
        self.c = [HiLo.of(x) for x in self.c]
        self.C = [[HiLo.of(x) for x in cl] for cl in self.C]
        self.b = HiLo.of(self.b)

    @property
    def n_clusters(self) -> int:
        return len(self.blocks)

    @property
    def n_free(self) -> int:
        return int(self.b.hi.shape[0])

    def cluster_sizes(self) -> List[int]:
        return [int(c.hi.shape[0]) for c in self.c]

    def block_sizes(self) -> List[List[int]]:
        return [[bl.n for bl in cl] for cl in self.blocks]

    def check(self) -> None:
        """Structural sanity checks (subset of reference src/checks.jl:120-187 relevant to the path)."""
        N = self.n_free
        for j, cl in enumerate(self.blocks):
            P = self.c[j].hi.shape[0]
            if self.B[j].hi.shape != (P, N):
                raise ValueError(f"B[{j}] has shape {self.B[j].hi.shape}, expected {(P, N)}")
            for l, bl in enumerate(cl):
                if self.C[j][l].hi.shape != (bl.n, bl.n):
                    raise ValueError(f"C[{j}][{l}] has the wrong shape")
                for (r, s), d in bl.entries.items():
                    for p, e in d.items():
                        if not (0 <= p < P):
                            raise ValueError(f"constraint index {p} out of range in block ({j},{l})")
                        if isinstance(e, LowRankMat):
                            if e.delta != bl.delta:
                                raise ValueError("The subblocks (j,l,(r,s)) must have the same size for every r,s.")
                            if bl.high_rank:
                                raise ValueError("mixed low-rank / dense entries in one block")
                            t = bl.entries.get((s, r), {}).get(p)
                            if t is None or t.rank != e.rank:
                                raise ValueError(f"block ({j},{l}): entry ({s},{r}) of constraint {p} must be the transpose of ({r},{s})")
                        else:
                            if bl.m != 1 or e.hi.shape != (bl.n, bl.n):
                                raise ValueError("dense entries need m == 1 and an n x n matrix")


# ----------------------------------------------------------------------------------------------
# Flat layout shared by the C ABI (include/clrs_hip.h) and the C oracle (oracle/clrs_oracle.c).
# ----------------------------------------------------------------------------------------------

@dataclass
class FlatSDP:
    """Flattened arrays in the exact order `clrs_sdp_desc` (include/clrs_hip.h) expects."""
    n_clusters: int
    n_free: int
    cluster_P: np.ndarray          # int32 [J]
    B: np.ndarray                  # f64, per cluster col-major P_j x N, concatenated
    B_lo: np.ndarray
    c: np.ndarray                  # f64 [sum P_j]
    c_lo: np.ndarray
    b: np.ndarray                  # f64 [N]
    b_lo: np.ndarray
    C: np.ndarray                  # f64, per block col-major n x n, concatenated (X/Y layout)
    C_lo: np.ndarray
    maximize: int
    constant: float
    n_blocks: int
    block_cluster: np.ndarray      # int32 [NB]
    block_m: np.ndarray            # int32 [NB]
    block_delta: np.ndarray        # int32 [NB]
    block_kind: np.ndarray         # int32 [NB] 0 = low rank, 1 = dense
    term_ptr: np.ndarray           # int64 [NB+1]
    term_p: np.ndarray             # int32 [T]
    term_r: np.ndarray             # int32 [T]
    term_s: np.ndarray             # int32 [T]
    term_rank: np.ndarray          # int32 [T]
    term_lambda: np.ndarray        # f64 [T]
    term_lambda_lo: np.ndarray
    term_vec_ptr: np.ndarray       # int64 [T+1] offsets into term_vs / term_ws
    term_vs: np.ndarray            # f64
    term_vs_lo: np.ndarray
    term_ws: np.ndarray            # f64
    term_ws_lo: np.ndarray
    dense_ptr: np.ndarray          # int64 [NB+1]
    dense_p: np.ndarray            # int32 [D]
    dense_A_ptr: np.ndarray        # int64 [D+1] offsets into dense_A
    dense_A: np.ndarray            # f64, n x n col-major per entry
    dense_A_lo: np.ndarray
    # derived
    block_n: np.ndarray            # int32 [NB]
    block_off: np.ndarray          # int64 [NB+1] offsets of each block in the X/Y layout
    cluster_off: np.ndarray        # int64 [J+1] offsets of each cluster in x (sum P_j)
    S_off: np.ndarray              # int64 [J+1] offsets of S_j (P_j^2) in the concatenated S layout
    # limb planes 3.. of the data arrays (name -> array of shape (planes - 2, len)), when the generator ran under `data_planes(n > 2)`: the sampled problem
    # at the working precision of a solve with more than two limbs.  Empty for every committed fixture; `shard_clusters` / `replicate_clusters` drop it.
    tails: dict = field(default_factory=dict)

    def data_planes_of(self, name: str, planes: int) -> np.ndarray:
        """the data array `name` ('B', 'c', 'b', 'C', 'term_lambda', 'term_vs', 'term_ws', 'dense_A') as (planes, len): hi, lo, tail planes, zero padded"""
        hi = np.ascontiguousarray(getattr(self, name), dtype=np.float64).reshape(-1)
        out = np.zeros((planes, hi.size))
        out[0] = hi
        if planes > 1:
            lo = getattr(self, name + "_lo", None)
            if lo is not None:
                out[1] = np.asarray(lo, dtype=np.float64).reshape(-1)
        t = self.tails.get(name) if self.tails else None
        if t is not None and planes > 2:
            k = min(planes - 2, t.shape[0])
            out[2:2 + k] = t[:k]
        return out

    @property
    def xy_len(self) -> int:
        return int(self.block_off[-1])

    @property
    def x_len(self) -> int:
        return int(self.cluster_off[-1])

    @property
    def S_len(self) -> int:
        return int(self.S_off[-1])

    @property
    def n_terms(self) -> int:
        return int(self.term_ptr[-1])


DATA_ARRAYS = ("B", "c", "b", "C", "term_lambda", "term_vs", "term_ws", "dense_A")


def _map_hilo(sdp: ClusteredLowRankSDP, fn) -> ClusteredLowRankSDP:
    """the same SDP with every HiLo replaced by fn(HiLo) (structure shared)"""
    def ent(e):
        return LowRankMat(fn(e.lam), fn(e.vs), fn(e.ws)) if isinstance(e, LowRankMat) else fn(e)
    blocks = [[Block(bl.m, bl.delta, {rs: {p: ent(e) for p, e in d.items()} for rs, d in bl.entries.items()}, bl.name) for bl in cl] for cl in sdp.blocks]
    return ClusteredLowRankSDP(sdp.maximize, sdp.constant, blocks, [fn(x) for x in sdp.B], [fn(x) for x in sdp.c],
                               [[fn(x) for x in cl] for cl in sdp.C], fn(sdp.b), sdp.names)


def flatten(sdp: ClusteredLowRankSDP) -> FlatSDP:
    """Flatten `sdp` into the C-ABI layout (`_flatten_core`); limb planes beyond (hi, lo) of its numbers (`data_planes`) go to `FlatSDP.tails`: the layout is
    linear in the data, so plane t of every data array is the `hi` array of the flattening of the SDP whose numbers are their plane t."""
    flat = _flatten_core(sdp)
    ntail = 0

    def probe(h):
        nonlocal ntail
        if h.tail is not None:
            ntail = max(ntail, len(h.tail))
        return h
    _map_hilo(sdp, probe)
    if ntail:
        planes = [_flatten_core(_map_hilo(sdp, lambda h, t=t: HiLo(h.plane(t + 2).copy(), None)), check=False) for t in range(ntail)]
        flat.tails = {name: np.stack([np.asarray(getattr(pl, name), dtype=np.float64).reshape(-1) for pl in planes]) for name in DATA_ARRAYS}
    return flat


def _flatten_core(sdp: ClusteredLowRankSDP, check: bool = True) -> FlatSDP:
    """Terms of a block are emitted sorted by (p, r, s, rank), which is also the order of the per-term A_Y output."""
    if check:
        sdp.check()
    J, N = sdp.n_clusters, sdp.n_free
    cluster_P = np.array(sdp.cluster_sizes(), dtype=np.int32)

    def colmajor(a):
        return np.asarray(a, dtype=np.float64).reshape(-1, order="F")

    B = np.concatenate([colmajor(x.hi) for x in sdp.B]) if J else np.zeros(0)
    B_lo = np.concatenate([colmajor(x.lo_or_zero()) for x in sdp.B]) if J else np.zeros(0)
    c = np.concatenate([x.hi.reshape(-1) for x in sdp.c]) if J else np.zeros(0)
    c_lo = np.concatenate([x.lo_or_zero().reshape(-1) for x in sdp.c]) if J else np.zeros(0)

    block_cluster, block_m, block_delta, block_kind = [], [], [], []
    term_ptr, dense_ptr = [0], [0]
    t_p, t_r, t_s, t_k, t_lam, t_lam_lo, t_vptr = [], [], [], [], [], [], [0]
    t_vs, t_vs_lo, t_ws, t_ws_lo = [], [], [], []
    d_p, d_Aptr, d_A, d_A_lo = [], [0], [], []
    Cs, Cs_lo = [], []
    for j, cl in enumerate(sdp.blocks):
        for l, bl in enumerate(cl):
            block_cluster.append(j)
            block_m.append(bl.m)
            block_delta.append(bl.delta)
            hr = bl.high_rank
            block_kind.append(1 if hr else 0)
            Cs.append(colmajor(sdp.C[j][l].hi))
            Cs_lo.append(colmajor(sdp.C[j][l].lo_or_zero()))
            if hr:
                d = bl.entries.get((0, 0), {})
                for p in sorted(d):
                    e = d[p]
                    d_p.append(p)
                    d_A.append(colmajor(e.hi))
                    d_A_lo.append(colmajor(e.lo_or_zero()))
                    d_Aptr.append(d_Aptr[-1] + bl.n * bl.n)
            else:
                items = []
                for (r, s), d in bl.entries.items():
                    for p, e in d.items():
                        for k in range(e.rank):
                            items.append((p, r, s, k, e))
                items.sort(key=lambda t: t[:4])
                for (p, r, s, k, e) in items:
                    t_p.append(p); t_r.append(r); t_s.append(s); t_k.append(k)
                    t_lam.append(e.lam.hi[k]); t_lam_lo.append(e.lam.lo_or_zero()[k])
                    t_vs.append(e.vs.hi[k]); t_vs_lo.append(e.vs.lo_or_zero()[k])
                    t_ws.append(e.ws.hi[k]); t_ws_lo.append(e.ws.lo_or_zero()[k])
                    t_vptr.append(t_vptr[-1] + bl.delta)
            term_ptr.append(len(t_p))
            dense_ptr.append(len(d_p))

    def cat(xs):
        return np.concatenate(xs).astype(np.float64) if xs else np.zeros(0, dtype=np.float64)

    block_m_a = np.array(block_m, dtype=np.int32)
    block_delta_a = np.array(block_delta, dtype=np.int32)
    block_n = (block_m_a * block_delta_a).astype(np.int32)
    block_off = np.concatenate([[0], np.cumsum(block_n.astype(np.int64) ** 2)]).astype(np.int64)
    cluster_off = np.concatenate([[0], np.cumsum(cluster_P.astype(np.int64))]).astype(np.int64)
    S_off = np.concatenate([[0], np.cumsum(cluster_P.astype(np.int64) ** 2)]).astype(np.int64)
    return FlatSDP(
        n_clusters=J, n_free=N, cluster_P=cluster_P, B=B, B_lo=B_lo, c=c, c_lo=c_lo,
        b=sdp.b.hi.astype(np.float64).copy(), b_lo=sdp.b.lo_or_zero().astype(np.float64).copy(),
        C=cat(Cs), C_lo=cat(Cs_lo), maximize=int(bool(sdp.maximize)), constant=float(sdp.constant),
        n_blocks=len(block_cluster),
        block_cluster=np.array(block_cluster, dtype=np.int32), block_m=block_m_a,
        block_delta=block_delta_a, block_kind=np.array(block_kind, dtype=np.int32),
        term_ptr=np.array(term_ptr, dtype=np.int64),
        term_p=np.array(t_p, dtype=np.int32), term_r=np.array(t_r, dtype=np.int32),
        term_s=np.array(t_s, dtype=np.int32), term_rank=np.array(t_k, dtype=np.int32),
        term_lambda=np.array(t_lam, dtype=np.float64), term_lambda_lo=np.array(t_lam_lo, dtype=np.float64),
        term_vec_ptr=np.array(t_vptr, dtype=np.int64),
        term_vs=cat(t_vs), term_vs_lo=cat(t_vs_lo), term_ws=cat(t_ws), term_ws_lo=cat(t_ws_lo),
        dense_ptr=np.array(dense_ptr, dtype=np.int64), dense_p=np.array(d_p, dtype=np.int32),
        dense_A_ptr=np.array(d_Aptr, dtype=np.int64), dense_A=cat(d_A), dense_A_lo=cat(d_A_lo),
        block_n=block_n, block_off=block_off, cluster_off=cluster_off, S_off=S_off,
    )


def blockdiag_pack(flat: FlatSDP, mats: List[np.ndarray]) -> np.ndarray:
    """Pack a list of per-block n x n matrices into the concatenated col-major X/Y layout."""
    out = np.empty(flat.xy_len, dtype=np.float64)
    for b, M in enumerate(mats):
        out[flat.block_off[b]:flat.block_off[b + 1]] = np.asarray(M, dtype=np.float64).reshape(-1, order="F")
    return out


def blockdiag_unpack(flat: FlatSDP, v: np.ndarray) -> List[np.ndarray]:
    res = []
    for b in range(flat.n_blocks):
        n = int(flat.block_n[b])
        res.append(np.asarray(v[flat.block_off[b]:flat.block_off[b + 1]]).reshape((n, n), order="F"))
    return res


def S_unpack(flat: FlatSDP, v: np.ndarray) -> List[np.ndarray]:
    res = []
    for j in range(flat.n_clusters):
        P = int(flat.cluster_P[j])
        res.append(np.asarray(v[flat.S_off[j]:flat.S_off[j + 1]]).reshape((P, P), order="F"))
    return res


def shard_clusters(flat: FlatSDP, clusters) -> FlatSDP:
    """The sub-problem holding only `clusters` (in the given, increasing order) and all N free
    variables: what one rank of the cluster-sharded path owns (the clusters are
    the reference's outer parallel axis, src/solver.jl:1245,1257,1537,1566)."""
    clusters = [int(j) for j in clusters]
    if sorted(set(clusters)) != clusters:
        raise ValueError("clusters must be strictly increasing")
    N = flat.n_free
    newj = {j: i for i, j in enumerate(clusters)}
    blocks = [b for b in range(flat.n_blocks) if int(flat.block_cluster[b]) in newj]

    def cat(xs, dt=np.float64):
        return np.concatenate(xs).astype(dt) if xs else np.zeros(0, dtype=dt)

    def per_cluster(arr, width):
        return cat([arr[int(flat.cluster_off[j]) * width:int(flat.cluster_off[j + 1]) * width] for j in clusters])

    def per_block(arr):
        return cat([arr[int(flat.block_off[b]):int(flat.block_off[b + 1])] for b in blocks])

    term_ptr, dense_ptr, tsel, dsel = [0], [0], [], []
    for b in blocks:
        tsel.extend(range(int(flat.term_ptr[b]), int(flat.term_ptr[b + 1])))
        dsel.extend(range(int(flat.dense_ptr[b]), int(flat.dense_ptr[b + 1])))
        term_ptr.append(len(tsel)); dense_ptr.append(len(dsel))
    tsel = np.array(tsel, dtype=np.int64); dsel = np.array(dsel, dtype=np.int64)

    def vecs(arr):
        return cat([arr[int(flat.term_vec_ptr[t]):int(flat.term_vec_ptr[t + 1])] for t in tsel])

    def dmats(arr):
        return cat([arr[int(flat.dense_A_ptr[e]):int(flat.dense_A_ptr[e + 1])] for e in dsel])

    tlen = (flat.term_vec_ptr[tsel + 1] - flat.term_vec_ptr[tsel]) if len(tsel) else np.zeros(0, np.int64)
    dlen = (flat.dense_A_ptr[dsel + 1] - flat.dense_A_ptr[dsel]) if len(dsel) else np.zeros(0, np.int64)
    cluster_P = flat.cluster_P[clusters].astype(np.int32)
    block_n = flat.block_n[blocks].astype(np.int32)
    return FlatSDP(
        n_clusters=len(clusters), n_free=N, cluster_P=cluster_P,
        B=per_cluster(flat.B, N), B_lo=per_cluster(flat.B_lo, N), c=per_cluster(flat.c, 1), c_lo=per_cluster(flat.c_lo, 1),
        b=flat.b.copy(), b_lo=flat.b_lo.copy(), C=per_block(flat.C), C_lo=per_block(flat.C_lo),
        maximize=flat.maximize, constant=flat.constant, n_blocks=len(blocks),
        block_cluster=np.array([newj[int(flat.block_cluster[b])] for b in blocks], dtype=np.int32),
        block_m=flat.block_m[blocks].astype(np.int32), block_delta=flat.block_delta[blocks].astype(np.int32),
        block_kind=flat.block_kind[blocks].astype(np.int32),
        term_ptr=np.array(term_ptr, dtype=np.int64),
        term_p=flat.term_p[tsel].astype(np.int32), term_r=flat.term_r[tsel].astype(np.int32),
        term_s=flat.term_s[tsel].astype(np.int32), term_rank=flat.term_rank[tsel].astype(np.int32),
        term_lambda=flat.term_lambda[tsel].astype(np.float64), term_lambda_lo=flat.term_lambda_lo[tsel].astype(np.float64),
        term_vec_ptr=np.concatenate([[0], np.cumsum(tlen)]).astype(np.int64),
        term_vs=vecs(flat.term_vs), term_vs_lo=vecs(flat.term_vs_lo), term_ws=vecs(flat.term_ws), term_ws_lo=vecs(flat.term_ws_lo),
        dense_ptr=np.array(dense_ptr, dtype=np.int64), dense_p=flat.dense_p[dsel].astype(np.int32),
        dense_A_ptr=np.concatenate([[0], np.cumsum(dlen)]).astype(np.int64),
        dense_A=dmats(flat.dense_A), dense_A_lo=dmats(flat.dense_A_lo),
        block_n=block_n,
        block_off=np.concatenate([[0], np.cumsum(block_n.astype(np.int64) ** 2)]).astype(np.int64),
        cluster_off=np.concatenate([[0], np.cumsum(cluster_P.astype(np.int64))]).astype(np.int64),
        S_off=np.concatenate([[0], np.cumsum(cluster_P.astype(np.int64) ** 2)]).astype(np.int64),
    )


def replicate_clusters(flat, copies: int):
    """A many-cluster instance with the block shapes of `flat`: the clusters of `flat` repeated `copies`
    times (independent clusters with identical constraint data, distinct iterates).  Used for the roofline measurement of the
    assembly kernels (bench.py) and for the many-clusters-per-wave parity tests."""
    import copy
    J = flat.n_clusters
    parts = [shard_clusters(flat, list(range(J))) for _ in range(copies)]
    f = copy.copy(parts[0])
    f.tails = {}
    cat = np.concatenate
    f.n_clusters = J * copies
    f.n_blocks = flat.n_blocks * copies
    for name in ("cluster_P", "B", "B_lo", "c", "c_lo", "C", "C_lo", "block_m", "block_delta", "block_kind", "term_p", "term_r",
                 "term_s", "term_rank", "term_lambda", "term_lambda_lo", "term_vs", "term_vs_lo", "term_ws", "term_ws_lo",
                 "dense_p", "dense_A", "dense_A_lo", "block_n"):
        setattr(f, name, cat([getattr(p, name) for p in parts]))
    f.block_cluster = cat([p.block_cluster + k * J for k, p in enumerate(parts)]).astype(np.int32)

    def cat_ptr(name):
        out, off = [np.zeros(1, np.int64)], 0
        for p in parts:
            a = getattr(p, name)
            out.append(a[1:] + off)
            off += int(a[-1])
        return cat(out).astype(np.int64)

    for name in ("term_ptr", "term_vec_ptr", "dense_ptr", "dense_A_ptr"):
        setattr(f, name, cat_ptr(name))
    f.block_off = np.concatenate([[0], np.cumsum(f.block_n.astype(np.int64) ** 2)]).astype(np.int64)
    f.cluster_off = np.concatenate([[0], np.cumsum(f.cluster_P.astype(np.int64))]).astype(np.int64)
    f.S_off = np.concatenate([[0], np.cumsum(f.cluster_P.astype(np.int64) ** 2)]).astype(np.int64)
    return f
