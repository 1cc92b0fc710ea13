"""SDPA sparse format (.dat-s) import/export and the scaled dense-constraint instance of BASELINE
config 5.  Format semantics follow the reference reader (src/SDPAtoCLRS.jl:3-83):

    maximise  <F0, Y>   s.t.  <F_i, Y> = c_i  (i = 1..m),  Y block-diagonal PSD

file layout: m / nblocks / block sizes (negative = diagonal block) / c vector / entries
`matno blkno i j value` (upper triangle, 1-based).  All PSD blocks end up in one cluster with dense
("high rank") constraint matrices, N = 0 free variables; a diagonal block of size k becomes k 1x1 blocks.
"""
from __future__ import annotations

import re
from typing import List

import numpy as np

from ..sdp import Block, ClusteredLowRankSDP, HiLo


class SDPAData:
    def __init__(self, m, block_sizes, c, F):
        self.m = m
        self.block_sizes = block_sizes      # signed, as in the file
        self.c = np.asarray(c, dtype=np.float64)
        self.F = F                          # F[matno][blkno] -> dense ndarray (matno 0 = objective)


def read_sdpa(path_or_text) -> SDPAData:
    if "\n" in path_or_text:
        text = path_or_text
    else:
        with open(path_or_text) as fh:
            text = fh.read()
    lines = []
    for ln in text.splitlines():
        s = ln.strip()
        if not s or s[0] in '"*':
            continue
        lines.append(s)
    num = lambda s: [float(t) for t in re.split(r"[\s,(){}]+", s) if t and re.match(r"^[+-]?[\d.]", t)]
    m = int(num(lines[0])[0])
    nb = int(num(lines[1])[0])
    sizes = [int(v) for v in num(lines[2])][:nb]
    c = num(lines[3])[:m]
    F = [[np.zeros((abs(s), abs(s))) for s in sizes] for _ in range(m + 1)]
    for ln in lines[4:]:
        v = num(ln)
        if len(v) < 5:
            continue
        k, blk, i, j, val = int(v[0]), int(v[1]) - 1, int(v[2]) - 1, int(v[3]) - 1, v[4]
        F[k][blk][i, j] = val
        F[k][blk][j, i] = val
    return SDPAData(m, sizes, c, F)


def write_sdpa(path, data: SDPAData) -> None:
    with open(path, "w") as fh:
        fh.write(f"{data.m}\n{len(data.block_sizes)}\n")
        fh.write(" ".join(str(s) for s in data.block_sizes) + "\n")
        fh.write(" ".join(repr(float(v)) for v in data.c) + "\n")
        for k in range(data.m + 1):
            for blk, M in enumerate(data.F[k]):
                n = M.shape[0]
                for i in range(n):
                    for j in range(i, n):
                        if M[i, j] != 0.0:
                            fh.write(f"{k} {blk + 1} {i + 1} {j + 1} {float(M[i, j])!r}\n")


def sdpa_to_sdp(data: SDPAData) -> ClusteredLowRankSDP:
    """One cluster, dense blocks, no free variables (reference src/SDPAtoCLRS.jl:51-83)."""
    # constraints without any matrix are dropped, like the reference does (src/SDPAtoCLRS.jl:66-78)
    keep = [k for k in range(1, data.m + 1) if any(np.any(M != 0.0) for M in data.F[k])]
    if len(keep) != data.m:
        data = SDPAData(len(keep), data.block_sizes, data.c[[k - 1 for k in keep]], [data.F[0]] + [data.F[k] for k in keep])
    blocks: List[Block] = []
    Cs = []
    for blk, s in enumerate(data.block_sizes):
        if s < 0:  # diagonal block -> |s| scalar blocks
            for i in range(-s):
                ent = {k - 1: HiLo.of(np.array([[data.F[k][blk][i, i]]])) for k in range(1, data.m + 1)
                       if data.F[k][blk][i, i] != 0.0}
                blocks.append(Block(1, 1, {(0, 0): ent}, (blk, i)))
                Cs.append(np.array([[data.F[0][blk][i, i]]]))
        else:
            ent = {k - 1: HiLo.of(data.F[k][blk]) for k in range(1, data.m + 1) if np.any(data.F[k][blk] != 0.0)}
            blocks.append(Block(1, s, {(0, 0): ent}, blk))
            Cs.append(np.array(data.F[0][blk]))
    return ClusteredLowRankSDP(maximize=True, constant=0.0, blocks=[blocks], B=[np.zeros((data.m, 0))],
                               c=[data.c.copy()], C=[Cs], b=np.zeros(0))


def sdpa_scaled(nb=64, bs=32, m=256, seed=64, blocks_per_constraint=2) -> SDPAData:
    """BASELINE config 5 synthetic: m constraints, each a symmetric Gaussian
    matrix on `blocks_per_constraint` random blocks; c_i = <A_i, I> so that Y = I is feasible; objective -(I + G G^T / bs)
    so that the problem is bounded.  `blocks_per_constraint = nb` makes every constraint matrix full block diagonal, like the
    two constraints of test/example.dat-s (both touch both blocks): P dense matrices in EVERY block, the 7.5 GFLOP / 134 MB
    assembly SURVEY.md section 8d quotes for (64, 32, 256)."""
    rng = np.random.default_rng(seed)
    sizes = [bs] * nb
    F = [[np.zeros((bs, bs)) for _ in range(nb)] for _ in range(m + 1)]
    for blk in range(nb):
        G = rng.standard_normal((bs, bs))
        F[0][blk] = -(np.eye(bs) + G @ G.T / bs)
    c = np.zeros(m)
    for i in range(1, m + 1):
        for blk in rng.choice(nb, size=min(blocks_per_constraint, nb), replace=False):
            G = rng.standard_normal((bs, bs))
            F[i][blk] = (G + G.T) / 2
            c[i - 1] += np.trace(F[i][blk])
    return SDPAData(m, sizes, c, F)
