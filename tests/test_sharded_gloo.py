"""The cluster-sharded orchestration (clrs_amd/sharded.py) over torch.distributed `gloo`, world_size 2, on CPU.

Each rank owns a subset of the clusters, computes with the numpy stand-in of tests/numpy_local.py and exchanges
the partial Q / u with the same all-reduces the GPU path issues over RCCL.  The sharded result must equal the
single-process result on the full problem."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, name, negate_block, ret, solve_first=False):
    sys.path.insert(0, ROOT)
    import clrs_amd  # noqa: F401
    from clrs_amd.sharded import ShardedSchur
    from tests.numpy_local import NumpyLocal
    from tests.util import chol_blocks_np, flat, spd_iterates
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        f = flat(name)
        X, Y = spd_iterates(f, seed=31)
        if negate_block is not None:
            Y[f.block_off[negate_block]:f.block_off[negate_block + 1]] *= -1.0
        Xc = chol_blocks_np(f, X)
        rng = np.random.default_rng(32)
        rx, ry = rng.standard_normal(f.x_len), rng.standard_normal(f.n_free)
        sh = ShardedSchur(f, rank, world, NumpyLocal)
        sh.decompose(torch.from_numpy(sh.take_xy(Xc)), torch.from_numpy(sh.take_xy(Y)))
        dx = torch.zeros(sh.shard.x_len, dtype=torch.float64)
        dy = torch.zeros(f.n_free, dtype=torch.float64)
        if solve_first:      # the exchange of Q rides on the first solve's all-reduce of u (one collective for [Q | u])
            calls = []
            orig = sh._all_reduce
            sh._all_reduce = lambda t: (calls.append(t.numel()), orig(t))[1]
            sh.solve(torch.from_numpy(sh.take_x(rx)), torch.from_numpy(ry), dx, dy)
            assert calls == [f.n_free * f.n_free + f.n_free], calls
            dx2, dy2 = torch.zeros_like(dx), torch.zeros_like(dy)
            sh.solve(torch.from_numpy(sh.take_x(rx)), torch.from_numpy(ry), dx2, dy2)      # second solve: u alone
            assert calls[1:] == [f.n_free], calls
            assert torch.equal(dx, dx2) and torch.equal(dy, dy2)
            st = sh.status()
        else:
            st = sh.status()
            if st == 0:
                sh.solve(torch.from_numpy(sh.take_x(rx)), torch.from_numpy(ry), dx, dy)
        ret[rank] = dict(clusters=sh.clusters, dx=dx.numpy().copy(), dy=dy.numpy().copy(), status=st)
    finally:
        dist.destroy_process_group()


def _run(name, world=2, negate_block=None, solve_first=False):
    port = _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, port, name, negate_block, ret, solve_first)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(180)
            assert p.exitcode == 0
        return {k: dict(v) for k, v in ret.items()}


@pytest.mark.parametrize("solve_first", [False, True], ids=["status_then_solve", "merged_exchange"])
@pytest.mark.parametrize("name", ["ns_8_3_2", "ce_8_3"])
def test_sharded_equals_single_process(name, solve_first, oracle_built):
    from oracle.oracle import Oracle
    from tests.util import chol_blocks_np, flat, spd_iterates
    f = flat(name)
    X, Y = spd_iterates(f, seed=31)
    Xc = chol_blocks_np(f, X)
    rng = np.random.default_rng(32)
    rx, ry = rng.standard_normal(f.x_len), rng.standard_normal(f.n_free)
    o = Oracle(f)
    o.schur_assemble(Xc, Y)
    assert o.schur_factor() == 0
    dx_ref, dy_ref = o.schur_solve(rx, ry)
    res = _run(name, solve_first=solve_first)
    seen = []
    for r in range(2):
        assert res[r]["status"] == 0
        seen += res[r]["clusters"]
        got = res[r]["dx"]      # tolerance 1e-7: the summation order of Q and u differs; cond of the KKT system ~1e7 here
        ref = np.concatenate([dx_ref[f.cluster_off[j]:f.cluster_off[j + 1]] for j in res[r]["clusters"]])
        assert np.max(np.abs(got - ref)) <= 1e-7 * max(1.0, np.max(np.abs(dx_ref)))
        assert np.max(np.abs(res[r]["dy"] - dy_ref)) <= 1e-7 * max(1.0, np.max(np.abs(dy_ref)))   # dy replicated
    assert sorted(seen) == list(range(f.n_clusters))
    assert np.array_equal(res[0]["dy"], res[1]["dy"])


def test_sharded_status_is_global_cluster_number(oracle_built):
    """A non-PD S_j on one rank is reported by every rank with the GLOBAL cluster number (j+1), like
    'S was not decomposed succesfully in block j' (src/solver.jl:1249)."""
    from tests.util import flat
    f = flat("ns_8_3_2")
    b = int(np.nonzero(f.block_cluster == 3)[0][-1])      # the big block of cluster 3 (its 1x1 sibling cannot flip the sign)
    res = _run("ns_8_3_2", negate_block=b)
    assert res[0]["status"] == res[1]["status"] == 4


def test_partition_is_balanced_and_complete():
    from clrs_amd.sharded import cluster_weights, partition_clusters
    from tests.util import flat
    f = flat("ns_8_15_2")
    w = cluster_weights(f)
    for world in (1, 2, 4, 8):
        parts = partition_clusters(f, world)
        assert sorted(j for p in parts for j in p) == list(range(f.n_clusters))
        loads = [sum(w[j] for j in p) for p in parts]
        assert max(loads) <= max(w.max(), w.sum() / world * 4 / 3 + 1e-9)     # LPT bound
    assert np.argmax(w) == 1                                                     # the 96-constraint cluster dominates


def test_shard_clusters_roundtrip(oracle_built):
    from clrs_amd.sdp import shard_clusters
    from oracle.oracle import Oracle
    from tests.util import chol_blocks_np, flat, spd_iterates
    f = flat("ns_8_3_2")
    X, Y = spd_iterates(f, seed=1)
    Xc = chol_blocks_np(f, X)
    S, AY = Oracle(f).schur_assemble(Xc, Y)
    for cl in ([0, 2, 5], [1, 3, 4, 6], list(range(7))):
        g = shard_clusters(f, cl)
        bl = [b for b in range(f.n_blocks) if f.block_cluster[b] in cl]
        Xs = np.concatenate([Xc[f.block_off[b]:f.block_off[b + 1]] for b in bl])
        Ys = np.concatenate([Y[f.block_off[b]:f.block_off[b + 1]] for b in bl])
        Sg, AYg = Oracle(g).schur_assemble(Xs, Ys)
        assert np.array_equal(Sg, np.concatenate([S[f.S_off[j]:f.S_off[j + 1]] for j in cl]))
        assert np.array_equal(AYg, np.concatenate([AY[f.term_ptr[b]:f.term_ptr[b + 1]] for b in bl]))


# ---- the scalar exchanges of the sharded interior-point iteration (clrs_mw_ipm_*): protocol mirror over gloo -------------------------
def test_exchange_mirror_matches_the_cpp_sources():
    """The slot layout and the issue order of clrs_amd.sharded.ScalarExchange / IPM_EXCHANGE_SCHEDULE are those of the C++ path: the MWG_*
    macros of clrs_mw_ipm.hip.h and the mw_ipm_exchange calls of clrs_mw_ipm_host.inc (stage, stream) in source order."""
    import re
    from clrs_amd.sharded import IPM_EXCHANGE_FIRST, IPM_EXCHANGE_LAST, IPM_EXCHANGE_SCHEDULE, ipm_slot_layout
    csrc = os.path.join(ROOT, "clusteredlowranksolver.jl_amd", "csrc")
    hdr = open(os.path.join(csrc, "clrs_mw_ipm.hip.h")).read()
    mac = dict(re.findall(r"#define MWG_(\w+)\(K, N\) (.*?)\s+/\*", hdr))
    mac["LEN"] = re.search(r"#define MWG_LEN\(K, N\) (.*)", hdr).group(1)
    for K, N in ((5, 31), (6, 193), (2, 0)):
        lay = ipm_slot_layout(K, N)
        env = {"K": K, "N": N}
        env["MWG_D"] = lambda k, n: eval(mac["D"].replace("(K)", str(k)).replace("(N)", str(n)))
        for name in ("S1", "S2", "BX", "D", "XY"):
            assert eval(mac[name].replace("MWG_D(K, N)", str(lay["D"])).replace("(K)", str(K)).replace("(N)", str(N))) == lay[name], name
        assert eval(mac["LEN"].replace("MWG_D(K, N)", str(lay["D"])).replace("(K)", str(K))) == lay["LEN"]
    inc = open(os.path.join(csrc, "clrs_mw_ipm_host.inc")).read()
    body = inc[inc.index("static int mw_ipm_enqueue"):inc.index("static int mw_ipm_finish")]
    calls = [(int(s_), st) for st, s_ in re.findall(r"mw_ipm_exchange\(c, (S|M), (\d)\)", body)]
    # the objectives' exchange is issued by mw_ipm_objectives on the stream of mw_ipm_tail -- the side stream, at the head of enqueue --: stage 15 (objectives,
    # <X,Y> and the iterate's share of the residuals in one record) when an iteration follows, stage 4 alone from mw_ipm_finish; the separate stages 0 and 1 of
    # enqueue are then skipped (xy_with_tail, res_with_tail)
    assert "mw_ipm_exchange(c, stream, with_xy ? 15 : 4)" in inc and "mw_ipm_objectives(c, st->side, word_value != 0 || with_next)" in inc
    assert "if (!st->xy_with_tail) {" in body and "if (!res_with_tail) {" in body and "res_with_tail = st->xy_with_tail && q.world > 1" in body
    assert tuple(calls[:2]) == IPM_EXCHANGE_FIRST and IPM_EXCHANGE_LAST == (4, "S")
    assert [(15, "S")] + calls[2:] == list(IPM_EXCHANGE_SCHEDULE)
    assert "slot[MWG_XY(K, N) + l] = v.l[l]" in hdr and "xy_merged ? MWG_XY(K, q.N) : MWG_S1(K, q.N)" in hdr


def _exchange_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    import clrs_amd  # noqa: F401
    from clrs_amd.sharded import IPM_EXCHANGE_SCHEDULE, ScalarExchange, partition_clusters
    from tests.util import flat, mw_with_tails, spd_iterates
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        f = flat("ns_8_3_2")
        K, N = 3, f.n_free
        mine = partition_clusters(f, world)[rank]
        X, Y = spd_iterates(f, seed=5)
        x = np.random.default_rng(6).standard_normal(f.x_len)
        B = np.zeros((f.x_len, N))
        for j in range(f.n_clusters):
            P = int(f.cluster_P[j]); o = int(f.cluster_off[j])
            B[o:o + P] = f.B[o * N:(o + P) * N].reshape(P, N, order="F")
        ex = ScalarExchange(K, N, rank, world)
        got = {}
        for stage, stream in IPM_EXCHANGE_SCHEDULE:          # both channels are one gloo group here; the ORDER is what is exercised
            slot = ex.new_slot()
            if stage in (0, 15):                             # <X,Y> over this rank's blocks, as K limbs (stage 15: in the XY field behind the plain doubles)
                s = sum(float(X[f.block_off[b]:f.block_off[b + 1]] @ Y[f.block_off[b]:f.block_off[b + 1]]) for b in range(f.n_blocks) if int(f.block_cluster[b]) in mine)
                o = ex.lay["S1"] if stage == 0 else ex.lay["XY"]
                slot[o:o + K] = mw_with_tails(np.array([s]), K, seed=rank)[:, 0]
            if stage in (1, 15):                             # -B^T x over this rank's rows (planar limbs), max|P| stand-in
                rows = np.concatenate([np.arange(int(f.cluster_off[j]), int(f.cluster_off[j + 1])) for j in mine])
                part = -(B[rows].T @ x[rows])
                slot[ex.lay["BX"]:ex.lay["BX"] + K * N] = mw_with_tails(part, K, seed=10 + rank).reshape(-1)
                slot[ex.lay["D"] + 0] = float(np.max(np.abs(x[rows])))
            if stage == 3:
                slot[ex.lay["D"] + 2] = -0.25 - rank
            slots = ex.gather(slot)
            assert slots.shape == (world, ex.lay["LEN"]) and np.array_equal(slots[rank], slot)
            if stage == 0:
                got["xy"] = ex.reduce_sum(slots, "S1")[0]
            if stage == 15:
                got["xy"] = ex.reduce_sum(slots, "XY")[0]
            if stage in (1, 15):
                got["btx"] = ex.reduce_sum(slots, "BX", N)
                got["maxP"] = ex.reduce_max(slots, 0)
            if stage == 3:
                got["eig"] = ex.reduce_min(slots, 2)
        got["ref_xy"] = float(X @ Y)
        got["ref_btx"] = -(B.T @ x)
        ret[rank] = got
    finally:
        dist.destroy_process_group()


def test_scalar_exchange_over_gloo_gives_identical_bits_on_every_rank():
    """World size 2 over gloo: every rank packs its record of a stage, all-gathers, reduces the slots in rank order -- the values of mu's
    numerator, of -B^T x and of the maxima / minima are bit-identical on both ranks and equal the single-process values to rounding."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_exchange_worker, args=(r, 2, port, ret)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(180)
            assert p.exitcode == 0
        res = {k: dict(v) for k, v in ret.items()}
    a, b = res[0], res[1]
    assert a["xy"] == b["xy"] and np.array_equal(a["btx"], b["btx"]) and a["maxP"] == b["maxP"] and a["eig"] == b["eig"] == -1.25
    assert abs(a["xy"] - a["ref_xy"]) <= 1e-12 * abs(a["ref_xy"])
    assert np.allclose(a["btx"], a["ref_btx"], rtol=1e-12, atol=1e-12)
