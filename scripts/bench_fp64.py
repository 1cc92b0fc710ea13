#!/usr/bin/env python3
"""bench_fp64.py -- the round-1 bench of the fp64 path, kept as a library for bench.py (`fp64_measurements`) and as a script.

NOTE: on cohnelkies(8,15) the fp64 factorisation fails (cond(S) > 1/eps, `parity.factor_status` != 0): the step rate this file
measures is the cost of the fp64 kernels on the problem's SHAPES, not of solvable iterations.  The contractual bench line is
bench.py's (multi-word path, factor_status == 0).

bench.py -- interior-point hot-path iterations/sec + Schur-assembly roofline on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Workload (BASELINE.json: "SpherePacking d=8, 2d=30"): the Cohn-Elkies sphere-packing SDP
cohnelkies(8, 15) of the reference's examples/SpherePacking.jl:117-185 -- 2 clusters of P = 32
constraints, PSD blocks 16x16 (rank-1 constraint matrices) + one 1x1 dense block, N = 31 free variables.
With N GPUs the problem is weak-scaled along the reference's own outer parallel axis (clusters): 2 clusters
per GPU (cohnelkies_multi with 2N-1 sign-constraint clusters), sharded one shard per rank, coupled only by
the RCCL all-reduce of Q (31 x 31) and of u (31) -- SURVEY.md section 8e; the exchange of Q rides on the first solve's
exchange of u (one all-reduce of the contiguous [Q | u]), so a step has two collectives.

One step = one pass of the hot path of one interior-point iteration on device-resident iterates:
    Cholesky of the X blocks            (src/solver.jl:388-399)
    Schur assembly                      (compute_S_integrated!, :1062-1226)
    chol S_j, L^-1 B, Q, chol Q         (compute_T_decomposition!, :1244-1279)
    2 x system solve                    (predictor + corrector, compute_search_direction! :1527-1582)
`value` = units/s with one unit = one hot-path pass over one 2-cluster share: a step of the N-GPU job (one iteration of the
N-times larger problem, 2N clusters) counts as N units, so that the aggregate scales with N under weak scaling.

`roofline` is measured in the same process with HIP events around every launch (library-side, on the
stream the kernels run on) for the Schur assembly of a many-cluster instance of the SAME block shapes
(`roofline.workload`), where the launch moves enough bytes for a bandwidth figure to mean something;
`roofline_named` is the same measurement on the 2-cluster problem itself (launch-latency bound: 41 KB
per assembly).  `cpu_baseline` times the CPU oracle (oracle/clrs_oracle.c, fp64, OpenMP) on the same
step on the host cores -- a port, not the reference (which needs Julia + Arb, absent here).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
FP64_PEAK_TFLOPS = 78.6      # MI355X fp64 vector = matrix peak (256 CU x 128 flop/clk x 2.4 GHz)



def _polyopt40_flat():
    """BASELINE config 2: PolyOpt at 2d = 40 (one cluster, one simple block n = 21, P = 41)."""
    import clrs_amd
    from clrs_amd import problems as P
    return clrs_amd.flatten(P.polyopt_random(20, seed=0)[0])


def _ns2_flat():
    """BASELINE config 3: Nsphere_packing(8, 15, [1/2, 1/2])."""
    import clrs_amd
    from clrs_amd import problems as P
    return clrs_amd.flatten(P.nsphere_packing(8, 15, [0.5, 0.5]))


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def seeded_iterates(flat, seed):
    """X, Y = I + G G^T / n per block (SURVEY.md section 8d synthetic iterates), xy layout."""
    rng = np.random.default_rng(seed)
    X, Y = np.zeros(flat.xy_len), np.zeros(flat.xy_len)
    for b in range(flat.n_blocks):
        n = int(flat.block_n[b])
        for M in (X, Y):
            G = rng.standard_normal((n, n))
            M[flat.block_off[b]:flat.block_off[b + 1]] = (np.eye(n) + G @ G.T / n).reshape(-1, order="F")
    return X, Y


def build_problem(n_pairs: int):
    """cohnelkies(8,15) for n_pairs == 1; otherwise 2*n_pairs clusters (1 + (2 n_pairs - 1) radii)."""
    import clrs_amd
    from clrs_amd.problems import cohnelkies_multi
    R = 2 * n_pairs - 1
    radii = [1.0 + 0.125 * k for k in range(R)]
    return clrs_amd.flatten(cohnelkies_multi(8, 15, radii))


def kernel_profile(ctx, run_once, reps):
    """Per-kernel HIP-event timing of `reps` eager passes; returns {name: (avg seconds per launch, launches per pass)}."""
    ctx.set_graph_mode(False)
    ctx.set_kernel_timing(-1)
    for _ in range(reps):
        run_once()
    kt = ctx.kernel_times()
    ctx.set_kernel_timing(-2)
    return {k: (sec / cnt, cnt / reps, sec / reps) for k, (_, sec, cnt) in kt.items()}


def main(argv=None, emit=True):
    """Runs the fp64 measurements; `emit=False` returns the result dict instead of printing the JSON line (bench.py)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--graph", action="store_true", help="replay one hipGraph per library call instead of launching kernels one by one "
                    "(slower for this path: a call is 1-3 kernels and a graph replay costs 10-16 us of host time)")
    ap.add_argument("--roofline-copies", type=int, default=8192,
                    help="cluster replication factor of the roofline instance (default: 16384 clusters, 359 MB of traffic per launch -- "
                         "more than the 256 MiB Infinity Cache holds; profiles/r01/o_size_sweep.txt has 4096 ... 131072 clusters)")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--split", action="store_true", help="with one GPU: still run the split-phase calls and the RCCL all-reduces (1-rank group)")
    args = ap.parse_args(argv)

    # stdout carries exactly one JSON line.  Native libraries write there too (RCCL prints its version banner and its "iommu=pt"
    # warning to stdout when the process group comes up): keep the real stdout aside and point fd 1 at stderr for the whole run.
    sys.stdout.flush()
    json_fd = os.dup(1) if emit else None
    if emit:
        os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP library is the only compute path)")
    torch.cuda.set_device(local_rank)
    torch.cuda.set_stream(torch.cuda.Stream())      # a real (capturable) stream; the library runs on torch's current stream
    if (world > 1 or args.split) and not dist.is_initialized():
        if os.environ.get("NCCL_DEBUG", "VERSION").upper() == "VERSION":
            os.environ["NCCL_DEBUG"] = "WARN"      # keep RCCL's version banner off stdout: rank 0 prints exactly one JSON line
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))

    import clrs_amd  # noqa: F401
    from clrs_amd.sharded import HipLocal, ShardedSchur

    t0 = time.time()
    flat = build_problem(world)
    if rank == 0:
        log(f"problem: {flat.n_clusters} clusters P={list(flat.cluster_P[:4])}.. N={flat.n_free} blocks n={list(flat.block_n[:4])}.. "
            f"generated in {time.time() - t0:.1f}s")
    parts = [[2 * r, 2 * r + 1] for r in range(world)]          # 2 clusters per GPU (identical weights)
    use_graph = args.graph
    sh = ShardedSchur(flat, rank, world, lambda s: HipLocal(s, local_rank, graph=use_graph), parts=parts, force_split=args.split)
    f = sh.shard
    ctx = sh.local.ctx
    dev = f"cuda:{local_rank}"
    X, Y = seeded_iterates(flat, seed=1)
    rng = np.random.default_rng(2)
    rhs_x_full, rhs_y = rng.standard_normal(flat.x_len), rng.standard_normal(flat.n_free)
    tX = torch.from_numpy(sh.take_xy(X)).to(dev)
    tY = torch.from_numpy(sh.take_xy(Y)).to(dev)
    tXc = torch.empty_like(tX)
    trx = torch.from_numpy(sh.take_x(rhs_x_full)).to(dev)
    tryy = torch.from_numpy(rhs_y).to(dev)
    tdx = torch.empty_like(trx)
    tdy = torch.empty_like(tryy)

    def step():
        sh.local.cholesky_blocks(tX, tXc)
        sh.decompose(tXc, tY)
        sh.solve(trx, tryy, tdx, tdy)      # predictor
        sh.solve(trx, tryy, tdx, tdy)      # corrector

    # ---- parity of this very configuration against the CPU oracle (checker only) ----
    parity = {}
    from oracle.oracle import Oracle
    o = Oracle(f, quad=True, use_lo=False)
    Xc_ref = np.concatenate([np.linalg.cholesky(sh.take_xy(X)[f.block_off[b]:f.block_off[b + 1]].reshape(int(f.block_n[b]), -1, order="F"))
                             .reshape(-1, order="F") for b in range(f.n_blocks)])
    S_ref, _ = o.schur_assemble(Xc_ref, sh.take_xy(Y))
    sh.local.cholesky_blocks(tX, tXc)
    sh.local.assemble(tXc, tY)
    torch.cuda.synchronize()
    from clrs_amd.sharded import _DevArray
    S_dev = torch.as_tensor(_DevArray(ctx.S_buffer(), f.S_len), device=dev).cpu().numpy()
    parity["S_rel_err_vs_f128_oracle"] = float(np.max(np.abs(S_dev - S_ref)) / np.max(np.abs(S_ref)))
    parity["Xchol_rel_err"] = float(np.max(np.abs(tXc.cpu().numpy() - Xc_ref)) / np.max(np.abs(Xc_ref)))
    assert parity["Xchol_rel_err"] < 1e-12, parity
    assert parity["S_rel_err_vs_f128_oracle"] < 1e-11, parity
    step()
    torch.cuda.synchronize()
    parity["factor_status"] = sh.status()      # cond(S) > 1/eps for 2d=30 in fp64: non-zero = the reference's SolverFailure

    # the 2d=30 instance cannot be factored in fp64: check factor + solve through the very same calls
    # on the 2d=6 instance of the same family, where fp64 carries the condition number
    if rank == 0:
        import clrs_amd as _c
        from clrs_amd.problems import cohnelkies_multi
        small = _c.flatten(cohnelkies_multi(8, 3, [1.0], orth_free=True))
        sh2 = ShardedSchur(small, 0, 1, lambda s_: HipLocal(s_, local_rank, graph=False))
        X2, Y2 = seeded_iterates(small, seed=5)
        r2 = np.random.default_rng(6)
        rx2, ry2 = r2.standard_normal(small.x_len), r2.standard_normal(small.n_free)
        a = [torch.from_numpy(v).to(dev) for v in (X2, Y2, rx2, ry2)]
        c2, dx2, dy2 = torch.empty_like(a[0]), torch.empty_like(a[2]), torch.empty_like(a[3])
        sh2.local.cholesky_blocks(a[0], c2)
        sh2.decompose(c2, a[1])
        sh2.solve(a[2], a[3], dx2, dy2)
        torch.cuda.synchronize()
        assert sh2.status() == 0
        o2 = Oracle(small, quad=True, use_lo=False)
        _, L2, _ = o2.cholesky_blocks(X2)
        o2.schur_assemble(L2, Y2)
        assert o2.schur_factor() == 0
        dxr, dyr = o2.schur_solve(rx2, ry2)
        sc = max(1.0, np.max(np.abs(dxr)), np.max(np.abs(dyr)))
        parity["solve_rel_err_2d6_vs_f128_oracle"] = float(max(np.max(np.abs(dx2.cpu().numpy() - dxr)), np.max(np.abs(dy2.cpu().numpy() - dyr))) / sc)
        assert parity["solve_rel_err_2d6_vs_f128_oracle"] < 1e-7, parity
        sh2.close()

    # ---- device wake-up (untimed, before the W warmup steps): a process that starts on an idle box has been seen to run its first
    # ~100 ms of launches several times slower (a 2000-step loop at 0.47 ms per step instead of 0.054) -- a fixed number of steps, the
    # same on every rank (the steps of a sharded run contain collectives) ----
    for _ in range(3000 if emit else 300):
        step()
    torch.cuda.synchronize()
    # ---- timed region: W warmup + exactly K steps, barrier + synchronize on both sides ----
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    # unit of work = one hot-path pass over one 2-cluster cohnelkies(8,15)-sized share; a step of the N-GPU job (2N clusters) is N units
    value = world * args.steps / elapsed

    out = {
        "metric": "interior-point iterations/sec (hot path: chol X + Schur assembly + block-Cholesky factor + 2 solves)",
        "value": value, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "SpherePacking cohnelkies(8,15): d=8, 2d=30; 2 clusters/GPU P=32, blocks 16x16 r1 + 1x1 dense, N=31",
                   "clusters": int(flat.n_clusters), "clusters_per_gpu": 2, "n_free": int(flat.n_free),
                   "unit_of_work": "one hot-path pass over one 2-cluster share; a step of the N-GPU job (one iteration of the 2N-cluster problem) = N units",
                   "launch": "hipGraph" if use_graph else "eager", "collective": "2 RCCL all-reduces per step: [Q(31x31) | u(31)] with the first solve, u(31) with the second" if (world > 1 or args.split) else "none"},
        "parity": parity,
    }

    if rank == 0:
        cnt = ctx.counters()
        # ---- per-kernel profile of the named problem (eager, HIP events around each launch) ----
        prof = kernel_profile(ctx, lambda: sh.local.assemble(tXc, tY), 50)
        asm_s = sum(v[2] for v in prof.values())
        dom = max(prof.items(), key=lambda kv: kv[1][2])
        out["roofline_named"] = {"bound": "hbm", "phase": "schur_assemble", "kernel": dom[0], "kernel_avg_us": 1e6 * dom[1][0],
                                 "kernel_launches_per_assembly": dom[1][1], "assembly_us": 1e6 * asm_s,
                                 "achieved": cnt["assemble_bytes"] / asm_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": cnt["assemble_bytes"] / asm_s / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                 "algorithmic_bytes": cnt["assemble_bytes"], "algorithmic_flops": cnt["assemble_flops"],
                                 "kernels_us": {k: round(1e6 * v[2], 3) for k, v in prof.items()}}
        if use_graph:
            ctx.set_graph_mode(True)

        # ---- roofline instance: the same block shapes, many clusters per launch ----
        from clrs_amd.sdp import replicate_clusters
        big = replicate_clusters(f, args.roofline_copies)
        from clrs_amd.solver import SchurContext
        bctx = SchurContext(big, device=local_rank)
        bctx.set_stream(torch.cuda.current_stream().cuda_stream)
        bX, bY = seeded_iterates(big, seed=3)
        bXc = np.concatenate([np.linalg.cholesky(bX[big.block_off[b]:big.block_off[b + 1]].reshape(int(big.block_n[b]), -1, order="F"))
                              .reshape(-1, order="F") for b in range(big.n_blocks)])
        tbXc, tbY = torch.from_numpy(bXc).to(dev), torch.from_numpy(bY).to(dev)
        # clocks and caches in their steady state before the timed launches: the device ramps its clocks up over the first ~15 ms of sustained load (measured,
        # scripts/w3_time.py: batches of 200 launches 63.6, 59.1, 58.2, 58.2, 58.3 us per launch) -- 50 launches, as this warm-up was until round 5, end in the
        # middle of the ramp.  Batches of 100 until a batch is no longer 1 % faster than the one before it (at most 12).
        warm_us = []
        for _ in range(12):
            w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            w0.record()
            for _ in range(100):
                bctx.assemble_dev(tbXc.data_ptr(), tbY.data_ptr())
            w1.record()
            w1.synchronize()
            warm_us.append(10.0 * w0.elapsed_time(w1))
            if len(warm_us) >= 3 and warm_us[-1] > 0.99 * warm_us[-2] and warm_us[-2] > 0.99 * warm_us[-3]:
                break
        torch.cuda.synchronize()
        bprof = kernel_profile(bctx, lambda: bctx.assemble_dev(tbXc.data_ptr(), tbY.data_ptr()), 20)      # which kernels, launches per assembly
        # the timed region: 100 assemblies back to back between two HIP events on the stream the kernels run on (the library is
        # bound to torch's current stream).  One event pair per launch, as in kernel_profile, adds the event packets' own 3-5 us to
        # every launch; the batch includes the gaps between launches instead (conservative against rocprofv3's pure durations).
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_timed = 400
        ev0.record()
        for _ in range(n_timed):
            bctx.assemble_dev(tbXc.data_ptr(), tbY.data_ptr())
        ev1.record()
        ev1.synchronize()
        batch_s = 1e-3 * ev0.elapsed_time(ev1) / n_timed
        bcnt = bctx.counters()
        per_launch_events_s = sum(v[2] for v in bprof.values())
        bdom = max(bprof.items(), key=lambda kv: kv[1][2])
        assert len(bprof) == 1 and bdom[1][1] == 1.0, bprof      # the assembly of this instance is ONE launch of one kernel
        basm = batch_s
        # spot-check the big instance too: first and last cluster against the oracle
        Sb = torch.as_tensor(_DevArray(bctx.S_buffer(), big.S_len), device=dev).cpu().numpy()
        ob = Oracle(f, quad=False)
        nxy = f.xy_len
        for k in (0, args.roofline_copies - 1):
            Sk, _ = ob.schur_assemble(bXc[k * nxy:(k + 1) * nxy], bY[k * nxy:(k + 1) * nxy])
            err = np.max(np.abs(Sb[k * f.S_len:(k + 1) * f.S_len] - Sk)) / np.max(np.abs(Sk))
            assert err < 1e-10, ("roofline instance parity", k, err)
        out["roofline"] = {"bound": "hbm", "phase": "schur_assemble", "kernel": bdom[0], "kernel_avg_us": 1e6 * basm,
                           "timed_region": f"{n_timed} launches back to back between two HIP events, behind warm-up batches of 100 launches until the batch time is steady",
                           "warmup_batches_us_per_launch": warm_us,
                           "kernel_avg_us_event_pair_per_launch": 1e6 * per_launch_events_s,
                           "kernel_launches_per_assembly": bdom[1][1], "assembly_us": 1e6 * basm,
                           "achieved": bcnt["assemble_bytes"] / basm / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": bcnt["assemble_bytes"] / basm / 1e9 / HBM_PEAK_GBS, "traffic": None,
                           "workload": f"{big.n_clusters} clusters / {big.n_blocks} PSD blocks of the cohnelkies(8,15) shapes in one assembly",
                           "algorithmic_bytes": bcnt["assemble_bytes"], "algorithmic_flops": bcnt["assemble_flops"],
                           "achieved_gflops": bcnt["assemble_flops"] / basm / 1e9,
                           "kernels_us": {k: round(1e6 * v[2], 3) for k, v in bprof.items()}}
        bctx.close()
        # HBM traffic of that launch from the committed PMC passes (rocprofv3 cannot run inside this process)
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[bdom[0]]
            if tr.get("clusters") == big.n_clusters:
                out["roofline"]["traffic"] = tr["traffic_bytes"]
                out["roofline"]["traffic_source"] = tr["source"]
                # what a pure streaming kernel of the same footprint and read : write mix reaches on this device, measured now
                # (include/clrs_hip.h: clrs_test_stream) -- the practical roof beside the data-sheet peak the fraction is quoted on
                import ctypes as _ct
                from clrs_amd._lib import load as _load, check as _check
                rd, wr = int(2 * 1024 * tr["fetch_size_kib_raw"]), int(1024 * tr["write_size_kib"])
                us = _ct.c_double(0.0)
                _check(_load().clrs_test_stream(local_rank, rd, wr, 20, _ct.byref(us)))
                copy_gbs = (rd + wr) / (us.value * 1e-6) / 1e9
                traffic_gbs = tr["traffic_bytes"] / (out["roofline"]["kernel_avg_us"] * 1e-6) / 1e9
                out["roofline"]["copy_roof"] = {"read_bytes": rd, "write_bytes": wr, "us": us.value, "GB/s": copy_gbs,
                                                "kernel_traffic_GB/s": traffic_gbs, "kernel_vs_copy": traffic_gbs / copy_gbs}
        except Exception as e:
            out["roofline"]["copy_roof_error"] = repr(e)

        # ---- dense ("high rank") branch at BASELINE config 5's stated scale: SDPA import x64 blocks ----
        try:
            from clrs_amd.problems import sdpa_scaled, sdpa_to_sdp
            import clrs_amd as _cc
            fd = _cc.flatten(sdpa_to_sdp(sdpa_scaled(nb=64, bs=32, m=256, seed=64, blocks_per_constraint=64)))
            dctx = SchurContext(fd, device=local_rank)
            dctx.set_stream(torch.cuda.current_stream().cuda_stream)
            dXd, dYd = seeded_iterates(fd, seed=7)
            dXc = np.concatenate([np.linalg.cholesky(dXd[fd.block_off[b]:fd.block_off[b + 1]].reshape(int(fd.block_n[b]), -1, order="F"))
                                  .reshape(-1, order="F") for b in range(fd.n_blocks)])
            tdXc, tdY = torch.from_numpy(dXc).to(dev), torch.from_numpy(dYd).to(dev)
            for _ in range(20):
                dctx.assemble_dev(tdXc.data_ptr(), tdY.data_ptr())
            torch.cuda.synchronize()
            dprof = kernel_profile(dctx, lambda: dctx.assemble_dev(tdXc.data_ptr(), tdY.data_ptr()), 20)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                dctx.assemble_dev(tdXc.data_ptr(), tdY.data_ptr())
            e1.record(); e1.synchronize()
            d_s = 1e-3 * e0.elapsed_time(e1) / 50
            dcnt = dctx.counters()
            ddom = max(dprof.items(), key=lambda kv: kv[1][2])
            out["roofline_dense"] = {"bound": "mfma", "phase": "schur_assemble, dense branch (src/solver.jl:1089-1104)", "kernel": ddom[0],
                                     "assembly_us": 1e6 * d_s, "achieved": dcnt["assemble_flops"] / d_s / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                     "frac": dcnt["assemble_flops"] / d_s / 1e12 / FP64_PEAK_TFLOPS, "traffic": None,
                                     "workload": "sdpa_scaled(nb=64, bs=32, m=256, blocks_per_constraint=64): 64 dense 32x32 blocks, P = 256 constraint matrices in EVERY block, like "
                                                 "the two constraints of test/example.dat-s (BASELINE config 5 scaled x64; SURVEY.md section 8d: 7.5 GFLOP / 134 MB)",
                                     "algorithmic_flops": dcnt["assemble_flops"], "algorithmic_bytes": dcnt["assemble_bytes"],
                                     "flops_model": "P (6 n^3) + P^2 n^2 per block over the constraints present in it (SURVEY.md section 8d)",
                                     "kernels_us": {k: round(1e6 * v[2], 3) for k, v in dprof.items()}}
            dctx.close()
        except Exception as e:
            out["roofline_dense"] = {"error": repr(e)}

        # ---- the assembly on the other block shapes of BASELINE's configurations, many clusters per launch: PolyOpt 2d = 40 (one simple
        #      block n = 21, P = 41 per cluster: k_cluster_assemble_w4 since round 5) and Nsphere_packing(8,15,[1/2,1/2]) (its two 2 x 2 blocks of 16 x 16
        #      sub-blocks, P = 96: k_cluster_assemble_w5; the 2 x 2 block of 1 x 1 sub-blocks and the 1 x 1 dense clusters stay on the general kernel) ----
        out["roofline_shapes"] = {}
        for sname, builder, copies in (("polyopt_2d40", lambda: _polyopt40_flat(), 4096), ("Nsphere_packing(8,15,[1/2,1/2])", lambda: _ns2_flat(), 256)):
            try:
                fs = builder()
                bigs = replicate_clusters(fs, copies)
                sctx = SchurContext(bigs, device=local_rank)
                sctx.set_stream(torch.cuda.current_stream().cuda_stream)
                sX, sY = seeded_iterates(bigs, seed=5)
                sXc = np.concatenate([np.linalg.cholesky(sX[bigs.block_off[b]:bigs.block_off[b + 1]].reshape(int(bigs.block_n[b]), -1, order="F"))
                                      .reshape(-1, order="F") for b in range(bigs.n_blocks)])
                tsXc, tsY = torch.from_numpy(sXc).to(dev), torch.from_numpy(sY).to(dev)
                for _ in range(60):
                    sctx.assemble_dev(tsXc.data_ptr(), tsY.data_ptr())
                torch.cuda.synchronize()
                sprof = kernel_profile(sctx, lambda: sctx.assemble_dev(tsXc.data_ptr(), tsY.data_ptr()), 10)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(100):
                    sctx.assemble_dev(tsXc.data_ptr(), tsY.data_ptr())
                e1.record(); e1.synchronize()
                s_s = 1e-3 * e0.elapsed_time(e1) / 100
                scnt = sctx.counters()
                sdom = max(sprof.items(), key=lambda kv: kv[1][2])
                shapes = sorted(set((int(n_), int(P_)) for n_, P_ in zip(fs.block_n, fs.cluster_P[fs.block_cluster])))
                out["roofline_shapes"][sname] = {
                    "bound": "hbm", "phase": "schur_assemble", "kernel": sdom[0], "assembly_us": 1e6 * s_s,
                    "achieved": scnt["assemble_bytes"] / s_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": scnt["assemble_bytes"] / s_s / 1e9 / HBM_PEAK_GBS,
                    "traffic": None, "achieved_tflops": scnt["assemble_flops"] / s_s / 1e12, "frac_of_fp64_mfma_peak": scnt["assemble_flops"] / s_s / 1e12 / FP64_PEAK_TFLOPS,
                    "algorithmic_bytes": scnt["assemble_bytes"], "algorithmic_flops": scnt["assemble_flops"],
                    "workload": "%d clusters / %d PSD blocks with the shapes (n, P) %s in one assembly" % (bigs.n_clusters, bigs.n_blocks, shapes),
                    "clusters_by_k_cluster_assemble_w4": sctx.wave4_clusters(), "clusters_by_k_cluster_assemble_w5": sctx.wave5_clusters(),
                    "kernels_us": {k: round(1e6 * v[2], 3) for k, v in sprof.items()}}
                sctx.close()
            except Exception as e:
                out["roofline_shapes"][sname] = {"error": repr(e)}

        # ---- staged (large-block) regime on roofline instance R: config-2 structure at n = 1025, P = 2049 ----
        try:
            from clrs_amd.problems import polyopt_scaled
            import clrs_amd as _cc
            fr = _cc.flatten(polyopt_scaled(1024))
            rctx = SchurContext(fr, device=local_rank)
            rctx.set_stream(torch.cuda.current_stream().cuda_stream)
            rX, rY = seeded_iterates(fr, seed=1)
            trX, trY = torch.from_numpy(rX).to(dev), torch.from_numpy(rY).to(dev)
            trXc = torch.empty_like(trX)

            def rstep():
                rctx.cholesky_blocks_dev(trX.data_ptr(), trXc.data_ptr())
                rctx.assemble_dev(trXc.data_ptr(), trY.data_ptr())
                rctx.factor_dev()
            for _ in range(2):
                rstep()
            torch.cuda.synchronize()
            assert rctx.sync_status() == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                rstep()
            e1.record(); e1.synchronize()
            r_s = 1e-3 * e0.elapsed_time(e1) / 5
            # the Schur assembly alone (the phase SURVEY.md section 8d's 25.8 GFLOP / 67 MB are quoted for), same iterates
            e0.record()
            for _ in range(5):
                rctx.assemble_dev(trXc.data_ptr(), trY.data_ptr())
            e1.record(); e1.synchronize()
            ra_s = 1e-3 * e0.elapsed_time(e1) / 5
            rcnt = rctx.counters()
            rfl = rcnt["assemble_flops"] + rcnt["factor_flops"]
            out["roofline_R"] = {"bound": "mfma", "phase": "chol X + schur_assemble + factor, staged (large-block) kernels", "kernel": "k_gemm_f64_t + k_chol_level + k_trtri_diag",
                                 "ms": 1e3 * r_s, "achieved": rfl / r_s / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": rfl / r_s / 1e12 / FP64_PEAK_TFLOPS,
                                 "traffic": None, "algorithmic_flops": rfl,
                                 "assembly": {"ms": 1e3 * ra_s, "algorithmic_flops": rcnt["assemble_flops"], "achieved": rcnt["assemble_flops"] / ra_s / 1e12,
                                              "frac": rcnt["assemble_flops"] / ra_s / 1e12 / FP64_PEAK_TFLOPS,
                                              "note": "schur_assemble alone; the symmetric pairing matrices of W = V blocks are formed as lower tiles only, "
                                                      "the flop count is SURVEY.md section 8d's 2 (2 n^2 U + 2 n U^2) + U^2 per block all the same"},
                                 "workload": "polyopt_scaled(1024): one cluster, one PSD block n = 1025, P = 2049 rank-1 constraints (SURVEY.md section 8d, roofline "
                                             "instance R; polyopt_scaled(2048): profiles/r02/u_staged_polyopt2048.txt)",
                                 "launches": rctx.plan_info()}
            rctx.close()
        except Exception as e:
            out["roofline_R"] = {"error": repr(e)}

        # ---- the complete interior-point method, device resident, on the instances fp64 can solve ----
        # every iteration = residuals + predictor + corrector + step lengths + update around the same hot path; one host sync per iteration
        try:
            from clrs_amd.problems import delsarte, polyopt_random
            from clrs_amd.solver import solvesdp_device
            full = {}
            for label, sdp_f, expect in (("polyopt_2d40 (BASELINE config 2)", lambda: _c.flatten(polyopt_random(20, seed=0)[0]), None),
                                         ("delsarte(3,10,1/2) (BASELINE config 1)", lambda: _c.flatten(delsarte(3, 10, 0.5)), 13.158314)):
                ff = sdp_f()
                cx = SchurContext(ff, device=local_rank)
                solvesdp_device(ff, ctx=cx)                                   # warm up
                # the loop synchronises with the host once per iteration: its rate follows whatever else runs on the host's cores (boxes share a host;
                # 2.4k-5.3k iterations/s were seen for the same build on PolyOpt 2d = 40) -- the best of three rounds of five solves is reported, all three kept
                rounds = []
                for _round in range(3):
                    t1 = time.perf_counter()
                    reps, its = 5, 0
                    for _ in range(reps):
                        rr = solvesdp_device(ff, ctx=cx)
                        its += rr.iterations
                    rounds.append((time.perf_counter() - t1) / its)
                dt, its = min(rounds), 1
                ent = {"iterations_per_s": its / dt, "iterations": rr.iterations, "status": rr.status, "primal_objective": rr.primal_objective,
                       "dual_objective": rr.dual_objective, "us_per_iteration": 1e6 * dt / its, "us_per_iteration_rounds": [1e6 * x for x in rounds]}
                if expect is not None:
                    ent["expected"] = expect
                    assert abs(rr.primal_objective - expect) <= 1e-5 * abs(expect), ent
                if not args.skip_cpu:
                    oo = Oracle(ff, quad=False)
                    best_cpu = 0.0
                    for thr in sorted({1, min(8, os.cpu_count() or 1)}):
                        oo.set_num_threads(thr)
                        t1 = time.perf_counter()
                        n_it = 0
                        while time.perf_counter() - t1 < 1.5:
                            ro = oo.solvesdp(omega_p=1e4, omega_d=1e4, duality_gap_threshold=1e-7, dual_error_threshold=1e-9, primal_error_threshold=1e-9)
                            n_it += ro["iterations"]
                        best_cpu = max(best_cpu, n_it / (time.perf_counter() - t1))
                    ent["cpu_port_iterations_per_s"] = best_cpu
                cx.close()
                full[label] = ent
            out["full_ipm_device_resident"] = full
        except Exception as e:      # never lose the headline line because of the secondary measurement
            out["full_ipm_device_resident"] = {"error": repr(e)}

        # ---- CPU baseline: the fp64 OpenMP oracle on the same step, bounded sample ----
        if not args.skip_cpu and world == 1:
            oc = Oracle(f, quad=False)
            Xs, Ys = sh.take_xy(X), sh.take_xy(Y)
            rx = sh.take_x(rhs_x_full)

            def cpu_pass():
                _, Lc, _ = oc.cholesky_blocks(Xs)
                oc.schur_assemble(Lc, Ys)
                oc.schur_factor()
                oc.schur_solve(rx, rhs_y)
                oc.schur_solve(rx, rhs_y)

            def cpu_rate(budget):
                n_it, t_cpu = 0, 0.0
                while t_cpu < budget and n_it < 200000:
                    t1 = time.perf_counter()
                    cpu_pass()
                    t_cpu += time.perf_counter() - t1
                    n_it += 1
                return n_it / t_cpu, n_it, t_cpu

            ncpu = min(os.cpu_count() or 1, 32)      # (a team of 256 threads never wins and leaves an idle pool behind: see bench.py)
            best = None
            for thr in sorted({1, min(8, ncpu), ncpu}):      # these matrices are tiny: fewer threads is usually faster
                oc.set_num_threads(thr)
                rate, _, _ = cpu_rate(1.0)
                if best is None or rate > best[0]:
                    best = (rate, thr)
            oc.set_num_threads(best[1])
            rate, n_it, t_cpu = cpu_rate(10.0)
            out["cpu_baseline"] = {"value": rate, "unit": "iterations/s", "cores": best[1], "kind": "port",
                                   "sample": f"{n_it} hot-path passes of the same 2-cluster problem in {t_cpu:.1f}s "
                                             f"(oracle/clrs_oracle.c fp64 + OpenMP through ctypes; best of 1/8/{ncpu} threads)"}
    if world > 1:
        dist.barrier()
    sh.close()
    if not emit:
        return out
    if world > 1 or args.split:
        dist.destroy_process_group()
    if rank == 0:
        import ctypes
        sys.stderr.flush()
        ctypes.CDLL(None).fflush(None)             # anything native code buffered goes out (to stderr) before the JSON line
        os.write(json_fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
