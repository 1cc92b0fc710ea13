#!/usr/bin/env python3
"""bench.py -- interior-point iterations/sec + Schur-assembly roofline on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--limbs 5]          (N > 1: starts N ranks itself, one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Workload (BASELINE.json: "SpherePacking d=8, 2d=30"): the Cohn-Elkies sphere-packing SDP cohnelkies(8, 15) of the reference's
examples/SpherePacking.jl:117-185 -- 2 clusters of P = 32 constraints, PSD blocks 16x16 (rank-1 constraint matrices) + one 1x1
dense block, N = 31 free variables -- at the precision the reference solves it at: test/runtests_solver.jl:19-20 runs it with
prec = 256 bits (Arb midpoints); here every number is 5 limbs of fp64 (~262 bits; `--limbs 4` = ~209 bits), the problem data
included (round 5: the sampled problem at the working precision, as the reference holds it; `full_solve_two_data_limbs` = the form of rounds 1-4).  In fp64 this instance cannot be factored at all (`fp64.parity.factor_status` below).

One STEP = one whole interior-point iteration of `solvesdp` (src/solver.jl:348-589) with the reference's default options, device
resident: mu, residuals, Cholesky of the X blocks, the hot path (Schur assembly :1062-1226, chol S_j / L^-1 B / Q / chol Q :1244-1279,
predictor and corrector solves :1527-1582), search directions, step lengths, update, objectives.  The timed region runs solves from the
reference's starting point (X = Y = 1e10 I) back to back through `clrs_mw_ipm_solve` until exactly K iterations have run (a solve
of this problem takes 56; the last one is cut at the remainder); `value` = iterations/s = BASELINE's metric.  The solve must end
Optimal within 1e-4 of pi^4/384 (test/runtests_solver.jl:19-20) before anything is timed.  `hot_path` reports the rate of hot-path
passes alone (the round-1/2 headline) on the iterate of iteration ceil(56/2), with its parity against the 320-bit oracle.

With N GPUs the problem is weak-scaled along the reference's own outer parallel axis (clusters): cohnelkies_multi with 2N clusters
(the f^ cluster and 2N - 1 sign-constraint clusters at different radii), partitioned over the ranks by `partition_clusters`, and the
WHOLE interior-point solve runs sharded: x, X, Y stay on their rank, y and every scalar are replicated bit for bit; per iteration the
library itself all-gathers (RCCL, two communicators: one per stream that exchanges) the partial Q (limbs x 31 x 31), the partial u
(limbs x 31, three times: the predictor's solve, the corrector's solve and its refinement step) and three small records (objectives + mu + p = b - B^T x and the primal-residual error; beta_c and the dual-residual error; the step lengths)
(clrs_mw_comm_init, clrs_mw_comm_init_side, clrs_mw_ipm_set_global).  A step of the N-GPU job = one
iteration of the 2N-cluster problem = N units of work; `value` = N x iterations/s.  That number alone would flatter: ONE GPU solves the same
2N-cluster problem unsharded in little more than the time of the 2-cluster one (the instance is latency bound at two clusters per rank).  So
rank 0 also times that SAME problem unsharded on its one GPU in the same job and the line carries `multi_gpu.speedup_vs_one_gpu_same_problem`;
and a second regime -- `--filled-clusters-per-rank` clusters per rank (default 32: a rank's share fills its chip), the same comparison -- is
measured and reported beside it (`multi_gpu.filled_regime`).  `--clusters-per-rank C` makes C the primary regime's share.

`cpu_baseline`: whole iterations of the same solve in the multi-precision CPU oracle (oracle/mpx.hpp, 256-bit truncation, the stand-in
for the reference's Arb arithmetic: kind "port") on the host cores; `roofline`: the HBM-bound fp64 Schur-assembly kernel on a
many-cluster instance of the same block shapes (the BASELINE metric "Schur-assembly GB/s vs fp64 roofline"); `roofline_timed`: the
kernel that dominates the timed step (k_mw_factor) against the fp64 issue rate of the compute units it occupies; `roofline_mw`: the
multi-word Schur assembly of many clusters against the fp64 pipe (78.6 TFLOP/s).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6      # MI355X fp64 vector = matrix peak (256 CU x 128 flop/clk x 2.4 GHz)
PI4_384 = math.pi ** 4 / 384


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def visible_gpus():
    """GPUs of this node as the kernel driver lists them (KFD topology nodes with SIMDs; CPU nodes have none), narrowed by HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES when set -- read from sysfs, so that the launcher never initialises HIP in the process that starts the ranks."""
    n, base = 0, "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(base):
            try:
                props = dict(line.split()[:2] for line in open(os.path.join(base, node, "properties")) if len(line.split()) >= 2)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except OSError:
        n = 0
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([t for t in v.split(",") if t.strip() != ""]))
    return n


def weak_scaling_instance(world, clusters_per_rank=2):
    """The problem of the N-rank job: cohnelkies_multi(8, 15) with 2 `world` clusters (the f^ cluster and 2 `world` - 1 sign-constraint clusters at radius
    scalings 1, 1 + 1/16, ...: with steps of 1/8 the 16-cluster instance ends NearOptimal at 5 limbs, with 1/16 every instance up to eight ranks ends
    Optimal -- scripts/multi_instances_check.py), its clusters repeated clusters_per_rank / 2 times (identical, redundant constraints: the same optimum;
    `replicate_clusters`) so that `partition_clusters` hands every rank `clusters_per_rank` clusters."""
    import clrs_amd
    from clrs_amd.problems import cohnelkies_multi
    from clrs_amd.sdp import replicate_clusters
    if clusters_per_rank < 2 or clusters_per_rank % 2:
        raise ValueError("clusters per rank: an even number >= 2")
    full = clrs_amd.flatten(cohnelkies_multi(8, 15, [1.0 + 0.0625 * k for k in range(2 * world - 1)]))
    return full if clusters_per_rank == 2 else replicate_clusters(full, clusters_per_rank // 2)


def launch_ranks(n, cmd=None, check_devices=True):
    """Start n copies of this script (or of `cmd`) as ranks 0..n-1 of one job (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run sets them),
    wait for all of them and return the worst exit code; a rank that dies takes the others down instead of leaving them in a collective."""
    import socket
    import subprocess
    have = visible_gpus() if check_devices else n   # (from sysfs: nothing in the launcher touches HIP before the ranks exist)
    if have < n:
        log(f"bench.py: --gpus {n} needs {n} GPUs on this node, {have} visible")
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen(list(cmd) if cmd is not None else [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    worst, alive, stopped = 0, set(range(n)), set()
    while alive:
        for r in sorted(alive):
            rc = procs[r].poll()
            if rc is None:
                continue
            alive.discard(r)
            if rc != 0 and r not in stopped:
                worst = max(worst, abs(rc) or 1)
                log(f"bench.py: rank {r} exited with code {rc}; stopping the other ranks")
                for o in alive:
                    stopped.add(o)
                    procs[o].terminate()
        time.sleep(0.05)
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=560, help="interior-point iterations in the timed region")
    ap.add_argument("--warmup", type=int, default=56)
    ap.add_argument("--pass-steps", type=int, default=200, help="hot-path passes of the secondary measurement (`hot_path`)")
    ap.add_argument("--limbs", type=int, default=5, help="fp64 words per number (5 covers the reference's prec = 256)")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--skip-fp64", action="store_true", help="skip the fp64 measurements (Schur-assembly HBM roofline, fp64 step on the problem's shapes)")
    ap.add_argument("--mw-copies", type=int, default=1024, help="replication factor of the multi-word roofline instance (2 clusters each)")
    ap.add_argument("--split", action="store_true", help="with one GPU: still take the sharded code path (1-rank process group, RCCL all-gathers inside the library)")
    ap.add_argument("--clusters-per-rank", type=int, default=2, help="N-rank job: clusters a rank holds in the PRIMARY regime (2 = the named problem's share: `value` stays comparable with N = 1)")
    ap.add_argument("--filled-clusters-per-rank", type=int, default=32, help="N-rank job: clusters per rank of the second regime reported under multi_gpu.filled_regime (0: skip it)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: launch with --nproc-per-node equal to --gpus")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: this process becomes the launcher of N ranks, one per GPU -- fresh children, started before anything here has
        # touched the GPU (never a re-exec of a process that has); rank 0's child prints the JSON line on the stdout it inherits
        raise SystemExit(launch_ranks(args.gpus))

    # stdout carries exactly one JSON line; native libraries write there too: keep the real stdout aside, point fd 1 at stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP library is the only compute path)")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    sharded = world > 1 or args.split
    if sharded:
        if os.environ.get("NCCL_DEBUG", "VERSION").upper() == "VERSION":
            os.environ["NCCL_DEBUG"] = "WARN"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device(dev))

    import clrs_amd
    from clrs_amd.mw import MwSchurContext, solvesdp_mw, LIMB_BITS
    from clrs_amd.problems import cohnelkies
    K = args.limbs
    bits = LIMB_BITS[K]
    thr = dict(dual_error_threshold=1e-30, primal_error_threshold=1e-30, duality_gap_threshold=1e-15) if K >= 5 else \
        dict(dual_error_threshold=1e-25, primal_error_threshold=1e-25, duality_gap_threshold=1e-12)      # what ~209 bits can reach
    t0 = time.time()
    # the sampled problem at the working precision, as the reference holds it (convert_to_prec, src/interface.jl:1078-1112): K limb planes per number
    # (round 5; rounds 1-4 passed two -- a neighbouring problem whose optimum differs in the 10th digit: `full_solve_two_data_limbs` below keeps that form)
    from clrs_amd.sdp import data_planes
    with data_planes(K):
        flat = clrs_amd.flatten(cohnelkies(8, 15))
    log(f"problem: cohnelkies(8,15), {flat.n_clusters} clusters P={list(flat.cluster_P)} N={flat.n_free} blocks n={list(flat.block_n)}, generated in {time.time() - t0:.1f}s")

    # ---- the job's problem: the named one on one GPU, the 2N-cluster weak-scaled one sharded over N ranks ----
    from clrs_amd.mw import shard_problem
    shard_info = None
    prob = flat
    def sharded_context(full_problem):
        """this rank's share of `full_problem` on a context with the library's two communicators (collective: every rank calls it)"""
        prob_, info_ = shard_problem(full_problem, rank, world)
        c_ = MwSchurContext(prob_, limbs=K, device=local_rank)
        ids = torch.zeros(2, 128, dtype=torch.uint8, device=dev)
        if rank == 0:
            ids = torch.tensor([list(MwSchurContext.comm_unique_id()), list(MwSchurContext.comm_unique_id())], dtype=torch.uint8, device=dev)
        dist.broadcast(ids, 0)
        c_.comm_init(bytes(ids[0].cpu().tolist()), rank, world)
        c_.comm_init_side(bytes(ids[1].cpu().tolist()))
        return prob_, info_, c_

    full = None
    if sharded:
        full = weak_scaling_instance(world, args.clusters_per_rank)
        prob, shard_info, ctx = sharded_context(full)
        log(f"rank {rank}: clusters {list(shard_info['cluster_ids'])} of {full.n_clusters}")
    else:
        ctx = MwSchurContext(prob, limbs=K, device=local_rank, data_limbs=K)

    def solve(**kw):
        return solvesdp_mw(prob, ctx=ctx, shard_info=shard_info, **thr, **kw)

    # ---- the solve itself, untimed: cold (first) and warm, with the reference's default options at K = 5 ----
    t0 = time.perf_counter()
    r_cold = solve()
    t_cold = time.perf_counter() - t0
    r = solve()
    n_it = r.iterations
    if not sharded:
        assert r.error_code == 0 and r.status == "Optimal", (r.status, r.error_code)
        assert abs(r.primal_objective - PI4_384) <= 1e-4, r.primal_objective       # test/runtests_solver.jl:19-20
    elif not (r.error_code == 0 and r.status == "Optimal"):
        # the weak-scaled instances end Optimal unsharded and through the in-process rehearsal of 2, 4 and 8 ranks (profiles/r04/j_*, k_*): a job whose solve
        # does not is not a measurement of this workload -- no JSON line, non-zero exit
        raise SystemExit(f"bench.py: rank {rank}: the sharded solve ended {r.status} (code {r.error_code}) after {r.iterations} iterations: nothing is reported")
    assert n_it > 0 and r_cold.iterations == n_it
    full_solve = {"iterations": n_it, "status": r.status, "error_code": r.error_code, "primal_objective": r.primal_objective, "dual_objective": r.dual_objective,
                  "expected": PI4_384 if not sharded else None, "tolerance": 1e-4, "first_solve_s": t_cold, "solve_s": r.time_total,
                  "iterations_per_s": n_it / r.time_total, "ms_per_iteration": 1e3 * r.time_total / n_it,
                  "what": "one whole solve from the reference's starting point with its default options (one call of clrs_mw_ipm_solve: iterations enqueued one "
                          "ahead of the record the host reads, termination test on the device); first_solve_s includes context warm-up on a cold device"}

    def run_iterations(n):
        """exactly n interior-point iterations: whole solves back to back, the last one cut at the remainder"""
        done = 0
        while done < n:
            rr = solve(maxiterations=n - done)
            assert rr.iterations > 0
            done += rr.iterations

    # ---- timed region: W warmup iterations + exactly K iterations, barrier + synchronize on both sides ----
    run_iterations(args.warmup)
    torch.cuda.synchronize()
    if sharded:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_iterations(args.steps)
    torch.cuda.synchronize()
    if sharded:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if sharded:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        ry0 = torch.tensor(r.y, device=dev)
        ry_ref = ry0.clone()
        dist.broadcast(ry_ref, 0)
        assert torch.equal(ry0, ry_ref), "the free variables y differ between ranks"
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * args.steps / elapsed
    multi = None
    if sharded:
        # what the exchanges of one iteration cost on this job's communicator (HIP events around all-gathers of the three message sizes, back to back:
        # clrs_mw_comm_probe), and the one-GPU rate of the named problem measured by rank 0 in this same job for comparison
        probe = ctx.comm_probe(50)
        per_iter = probe["q_us"] + 3 * probe["u_us"] + 3 * probe["record_us"]
        multi = {"ranks_in_process_group": dist.get_world_size(), "ranks_in_library_communicator": probe["world"], "backend": probe["backend"],
                 "allgather_us": {"partial_Q": probe["q_us"], "partial_u": probe["u_us"], "scalar_record": probe["record_us"]},
                 "exchanges_per_iteration": {"partial_Q": 1, "partial_u": 3, "scalar_record": 3},
                 "exchange_us_per_iteration_back_to_back": per_iter,
                 "what": "all-gathers of one sharded iteration: the partial Q once, the partial u three times (the predictor's solve; the corrector's and "
                         "its refinement step's), three scalar records (objectives + <X,Y> + p; beta_c and errors; step lengths), on two communicators (main / side stream); "
                         "the sum is what they cost issued back to back on one stream -- inside the iteration the side stream's one overlaps the factorisations"}
        def time_unsharded(problem, steps):
            """rank 0's one GPU alone on `problem`: (status, iterations per solve, seconds per iteration) over `steps` iterations"""
            c1 = MwSchurContext(problem, limbs=K, device=local_rank)
            r1 = solvesdp_mw(problem, ctx=c1, **thr)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            done = 0
            while done < steps:
                done += solvesdp_mw(problem, ctx=c1, maxiterations=steps - done, **thr).iterations
            torch.cuda.synchronize()
            t_single = time.perf_counter() - t1
            c1.close()
            return r1, t_single / steps

        if rank == 0:
            r1, s1 = time_unsharded(flat, args.steps)
            multi["single_gpu_iterations_per_s_named_problem"] = 1.0 / s1
            multi["single_gpu_ms_per_iteration_named_problem"] = 1e3 * s1
            # the honest comparison: the SAME problem the N ranks share, unsharded on one GPU
            r_same, s_same = time_unsharded(full, args.steps)
            multi["clusters_per_rank"] = args.clusters_per_rank
            multi["problem_clusters"] = int(full.n_clusters)
            multi["sharded_ms_per_iteration"] = ms_per_step
            multi["one_gpu_same_problem"] = {"status": r_same.status, "iterations": r_same.iterations, "ms_per_iteration": 1e3 * s_same,
                                             "primal_objective": r_same.primal_objective, "sharded_primal_objective": r.primal_objective}
            multi["speedup_vs_one_gpu_same_problem"] = 1e3 * s_same / ms_per_step
            multi["speedup_what"] = ("time per iteration of ONE GPU on the whole %d-cluster problem, unsharded, divided by the time per iteration of the %d-rank job on "
                                     "that same problem: below 1 the sharding loses (the instance is latency bound at %d clusters per rank: seven dependent all-gathers "
                                     "per iteration against kernels that use a few per cent of a chip)" % (full.n_clusters, world, args.clusters_per_rank))
        dist.barrier()
        # ---- second regime: a share that fills a rank's chip ----
        if args.filled_clusters_per_rank and args.filled_clusters_per_rank != args.clusters_per_rank:
            Cf = args.filled_clusters_per_rank
            full_f = weak_scaling_instance(world, Cf)
            prob_f, info_f, ctx_f = sharded_context(full_f)
            steps_f = min(args.steps, 112)
            rf = solvesdp_mw(prob_f, ctx=ctx_f, shard_info=info_f, **thr)
            torch.cuda.synchronize()
            dist.barrier()
            t1 = time.perf_counter()
            done = 0
            while done < steps_f:
                done += solvesdp_mw(prob_f, ctx=ctx_f, shard_info=info_f, maxiterations=steps_f - done, **thr).iterations
            torch.cuda.synchronize()
            dist.barrier()
            tf = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
            dist.all_reduce(tf, op=dist.ReduceOp.MAX)
            ms_f = 1e3 * float(tf.item()) / steps_f
            filled = {"clusters_per_rank": Cf, "problem_clusters": int(full_f.n_clusters), "status": rf.status, "error_code": rf.error_code, "iterations": rf.iterations,
                      "primal_objective": rf.primal_objective, "sharded_ms_per_iteration": ms_f, "iterations_timed": steps_f,
                      "valid": bool(rf.error_code == 0 and rf.status == "Optimal")}
            if rank == 0:
                r_same, s_same = time_unsharded(full_f, steps_f)
                filled["one_gpu_same_problem"] = {"status": r_same.status, "iterations": r_same.iterations, "ms_per_iteration": 1e3 * s_same, "primal_objective": r_same.primal_objective}
                filled["speedup_vs_one_gpu_same_problem"] = 1e3 * s_same / ms_f
            dist.barrier()
            ctx_f.comm_destroy()
            ctx_f.close()
            multi["filled_regime"] = filled

    # ---- the same solve with TWO data limbs (the form rounds 1-4 measured: a neighbouring problem, entries of B reach 1e27) for comparison ----
    data_k = None
    if rank == 0 and not sharded and K > 2:
        try:
            import copy as _copy
            flat_2 = _copy.copy(flat)
            flat_2.tails = {}
            c2 = MwSchurContext(flat_2, limbs=K, device=local_rank, data_limbs=2)
            solvesdp_mw(flat_2, ctx=c2, **thr)
            r2 = min((solvesdp_mw(flat_2, ctx=c2, **thr) for _ in range(3)), key=lambda r_: r_.time_total)
            c2.close()
            data_k = {"data_limbs": 2, "status": r2.status, "iterations": r2.iterations, "primal_objective": r2.primal_objective,
                      "ms_per_iteration": 1e3 * r2.time_total / r2.iterations, "headline_primal_objective": r.primal_objective,
                      "what": "whole solves of the same instance with the first two limb planes of its data only (what rounds 1-4 timed): another problem -- its optimum "
                              "differs from the headline's in the 10th digit (tests/test_mw_parity.py::test_problem_data_at_the_working_precision)"}
        except Exception as e:
            data_k = {"error": repr(e)}

    # ---- secondary: the hot path alone (chol X + assembly + factorisation + 2 solves) on a mid-trajectory iterate, single GPU ----
    hot = None
    parity = {}
    if rank == 0 and not sharded:
        mid = (n_it + 1) // 2
        r_mid = solve(maxiterations=mid)                                        # stops with code 2 after `mid` iterations; its iterate is (X, Y) of iteration mid + 1
        X, Y = r_mid.X, r_mid.Y
        dX, dY = torch.tensor(X, device=dev), torch.tensor(Y, device=dev)
        dXc = torch.empty_like(dX)
        rx, ry = np.zeros((K, flat.x_len)), np.zeros((K, flat.n_free))
        rx[0], ry[0] = 1.0, 1.0
        drx, dry = torch.tensor(rx, device=dev), torch.tensor(ry, device=dev)
        ddx, ddy = torch.empty_like(drx), torch.empty_like(dry)
        torch.cuda.synchronize()

        def step():
            ctx.cholesky_blocks_dev(dX.data_ptr(), dXc.data_ptr())
            ctx.assemble_dev(dXc.data_ptr(), dY.data_ptr())
            ctx.factor_dev()
            ctx.solve_dev(drx.data_ptr(), dry.data_ptr(), ddx.data_ptr(), ddy.data_ptr())      # predictor
            ctx.solve_dev(drx.data_ptr(), dry.data_ptr(), ddx.data_ptr(), ddy.data_ptr())      # corrector

        # ---- parity of this very pass against the CPU oracle at 320 bits (checker only) ----
        step()
        parity["factor_status"] = ctx.sync_status()
        parity["cholesky_status"] = ctx.sync_status_cholesky()
        assert parity["factor_status"] == 0 and parity["cholesky_status"] == 0, parity
        import math as _m
        from oracle.oracle import Oracle

        def relerr(a, b):
            d = [abs(_m.fsum(list(a[:, i]) + list(-b[:, i]))) for i in range(a.shape[1])]
            return float(max(d) / np.max(np.abs(b[0])))
        o = Oracle(flat, mp_bits=320)
        pad = lambda a: np.vstack([a, np.zeros((1, a.shape[1]))])
        st, Xc_ref = o.cholesky_blocks_mw(pad(X))
        S_ref, _ = o.schur_assemble_mw(Xc_ref, pad(Y))
        assert st == 0 and o.schur_factor() == 0
        dx_ref, dy_ref = o.schur_solve_mw(pad(rx), pad(ry))
        S_gpu, _ = ctx.compute_S_integrated(dXc.cpu().numpy(), Y)
        parity["iterate"] = f"(X, Y) after {mid} of {n_it} iterations of the solve"
        parity["S_rel_err_vs_320bit_oracle"] = relerr(S_gpu, S_ref)
        parity["dx_rel_err_vs_320bit_oracle"] = relerr(ddx.cpu().numpy(), dx_ref)
        parity["dy_rel_err_vs_320bit_oracle"] = relerr(ddy.cpu().numpy(), dy_ref)
        bx, by = o.kkt_backward_error_mw(S_ref, ddx.cpu().numpy(), ddy.cpu().numpy(), rx, ry)
        parity["kkt_backward_error_x_rows"], parity["kkt_backward_error_y_rows"] = bx, by
        parity["limb_bits"] = bits
        # the assembly inherits cond(X) of a mid-trajectory iterate (~1e9 here); the solve inherits cond(S) (~1e34): reported, and bounded loosely
        assert parity["S_rel_err_vs_320bit_oracle"] <= 2.0 ** -(53 * K - 64), parity
        assert parity["dx_rel_err_vs_320bit_oracle"] <= 1e-20 and parity["dy_rel_err_vs_320bit_oracle"] <= 1e-20, parity
        assert bx <= 2.0 ** -(53 * K - 60) and by <= 2.0 ** -(53 * K - 90), parity      # the explicit inverse factors lose up to ~45 bits on x rows, ~65 on y rows (tests/test_mw_parity.py)
        ctx.assemble_dev(dXc.data_ptr(), dY.data_ptr())      # compute_S_integrated staged host copies of the factors: restore the device-resident state
        ctx.factor_dev()
        for _ in range(30):
            step()
        ctx.sync_status()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.pass_steps):
            step()
        ctx.sync_status()
        torch.cuda.synchronize()
        t_pass = (time.perf_counter() - t1) / args.pass_steps
        hot = {"what": "hot-path passes alone on device-resident planar limbs: chol X + Schur assembly + chol S_j / L^-1 B / Q / chol Q + predictor and "
                       "corrector solve (15 kernels), on the iterate of iteration %d of %d" % (mid + 1, n_it),
               "passes_per_s": 1.0 / t_pass, "ms_per_pass": 1e3 * t_pass, "passes_timed": args.pass_steps}

    out = {
        "metric": "interior-point iterations/sec",
        "value": value, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,      # (BASELINE.json publishes no number for this workload)
        "dtype": f"f64x{K} (multi-word fp64: {K} limbs per number, ~{bits} bits; problem data f64x{2 if sharded else K})",
        "data": "generated: cohnelkies(8,15) built from the mathematics of the reference's examples/SpherePacking.jl (no dataset, no checkpoint); every "
                "iterate is the solve's own, from the reference's starting point X = Y = 1e10 I",
        "config": {"workload": "SpherePacking cohnelkies(8,15): d=8, 2d=30; 2 clusters P=32, blocks 16x16 r1 + 1x1 dense, N=31; whole interior-point "
                               f"iterations with the reference's default options at its precision (prec=256 -> {K} limbs), {n_it} per solve",
                   "clusters": int(full.n_clusters) if sharded else int(flat.n_clusters), "n_free": int(flat.n_free), "limbs": K, "data_limbs": 2 if sharded else K,
                   "unit_of_work": "one interior-point iteration over one 2-cluster share; a step of the N-GPU job (one iteration of the 2N-cluster problem) = N units",
                   "multi_gpu": (f"{int(full.n_clusters)} clusters partitioned over {world} ranks (partition_clusters); per iteration RCCL all-gathers of the partial Q, the partial u "
                                 "(three times: predictor, corrector and its refinement step) and three scalar records (objectives + mu + p; beta_c and errors; step lengths) inside the C ABI, two communicators "
                                 "(clrs_mw_comm_init, clrs_mw_comm_init_side, clrs_mw_ipm_set_global); y bit-identical on all ranks (asserted); hardware scaling curve: unmeasured by the builder (one-GPU boxes)") if sharded else "single GPU",
                   "launch": "eager, two streams, 35 kernels per iteration, one host wait per iteration on a record that is one iteration old"},
        "full_solve": full_solve,
    }
    if data_k is not None:
        out["full_solve_two_data_limbs"] = data_k
    if multi is not None:
        out["multi_gpu"] = multi
    if hot is not None:
        out["hot_path"] = hot
        out["parity"] = parity
        ms_pass = hot["ms_per_pass"]

    if rank == 0 and not sharded:
        ctx.set_timing(True)
        for _ in range(3):
            step()
        tm = ctx.timings()
        out["stage_us"] = {"schur_assemble": 1e6 * tm[0], "cholS_and_LinvB": 1e6 * tm[1], "Q": 1e6 * tm[3], "cholQ": 1e6 * tm[4], "one_solve": 1e6 * tm[5]}
        ctx.set_timing(False)

        # ---- the same step at the other supported limb counts (the iterate truncated / zero-extended): precision against time ----
        sweep = {str(K): ms_pass}
        for Kx in (4, 6, 8, 10):
            if Kx == K:
                continue
            cx = MwSchurContext(flat, limbs=Kx, device=local_rank)
            cut = lambda a: np.ascontiguousarray(np.vstack([a[:min(K, Kx)], np.zeros((max(0, Kx - K), a.shape[1]))]))
            eX, eY, erx, ery = (torch.tensor(cut(a), device=dev) for a in (X, Y, rx, ry))
            eXc, edx, edy = torch.empty_like(eX), torch.empty_like(erx), torch.empty_like(ery)

            def xstep():
                cx.cholesky_blocks_dev(eX.data_ptr(), eXc.data_ptr())
                cx.assemble_dev(eXc.data_ptr(), eY.data_ptr())
                cx.factor_dev()
                cx.solve_dev(erx.data_ptr(), ery.data_ptr(), edx.data_ptr(), edy.data_ptr())
                cx.solve_dev(erx.data_ptr(), ery.data_ptr(), edx.data_ptr(), edy.data_ptr())
            for _ in range(10):
                xstep()
            ok = cx.sync_status() == 0 and cx.sync_status_cholesky() == 0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                xstep()
            cx.sync_status()
            torch.cuda.synchronize()
            sweep[str(Kx)] = 1e3 * (time.perf_counter() - t0) / 100 if ok else None
            cx.close()
        out["hot_path"]["ms_per_pass_by_limbs"] = dict(sorted(sweep.items(), key=lambda kv: int(kv[0])))

        # ---- CPU baseline: whole iterations of the same solve in the multi-precision oracle, bounded sample ----
        if not args.skip_cpu and world == 1:
            from oracle.oracle import Oracle
            oc = Oracle(flat, mp_bits=256)
            # (threads: 1, 8, and the host's count capped at 32 -- a box gives one GPU's job 16 cores; a team of os.cpu_count() = 256 OpenMP threads never won
            # and left a pool of idle threads behind that slowed every host-driven GPU loop measured after it in this process: the fp64 device loop on
            # PolyOpt 2d = 40 showed 3.2k iterations/s behind it and 5.2k without)
            ncpu = min(os.cpu_count() or 1, 32)
            best = None
            for th in sorted({1, min(8, ncpu), ncpu}):      # these matrices are tiny: more threads is usually slower
                oc.set_num_threads(th)
                t1 = time.perf_counter()
                ro = oc.solvesdp(maxiterations=12)
                rate = ro["iterations"] / (time.perf_counter() - t1)
                if best is None or rate > best[0]:
                    best = (rate, th)
            oc.set_num_threads(best[1])
            n_cpu, t_cpu, ro = 0, 0.0, None
            while t_cpu < 10.0:
                t1 = time.perf_counter()
                ro = oc.solvesdp()
                t_cpu += time.perf_counter() - t1
                n_cpu += ro["iterations"]
            assert ro["error_code"] == 0 and abs(ro["p_obj"] - r.primal_objective) <= 1e-10
            rate = n_cpu / t_cpu
            out["cpu_baseline"] = {"value": rate, "unit": "iterations/s", "cores": best[1], "kind": "port",
                                   "sample": f"{n_cpu} interior-point iterations ({n_cpu // ro['iterations']} whole solves of the same problem with the same options, "
                                             f"{ro['iterations']} iterations each, same objective to 1e-10) in {t_cpu:.1f}s: oracle/clrs_oracle.c on the multi-limb type of "
                                             f"oracle/mpx.hpp truncated to 256 bits per operation (the stand-in for the reference's Arb midpoints at prec = 256; the "
                                             f"reference itself needs Julia + Arb), best of 1/8/{ncpu} threads; a PORT whose multiply-add (~21 ns at 256 bits) has not been calibrated against "
                                             f"Arb's (plausibly 1.5-2x faster): the ratio to it bounds nothing about the reference more tightly than that",
                                   "precision_bits": 256}
            out["speedup_vs_cpu_baseline"] = value / rate
            out["full_solve"]["cpu_oracle_256bit"] = {"iterations": ro["iterations"], "iterations_per_s": rate, "primal_objective": ro["p_obj"], "threads": best[1]}
            # the hot path alone in the oracle, same iterate (the secondary measurement's counterpart)
            oc.set_num_threads(1)

            def cpu_pass():
                st_, Lc = oc.cholesky_blocks_mw(X)
                oc.schur_assemble_mw(Lc, Y)
                assert st_ == 0 and oc.schur_factor() == 0
                oc.schur_solve_mw(rx, ry)
                oc.schur_solve_mw(rx, ry)
            n_p, t_p = 0, 0.0
            while t_p < 3.0:
                t1 = time.perf_counter()
                cpu_pass()
                t_p += time.perf_counter() - t1
                n_p += 1
            out["hot_path"]["cpu_oracle_256bit_passes_per_s"] = n_p / t_p
            out["hot_path"]["speedup_vs_cpu_oracle"] = out["hot_path"]["passes_per_s"] / (n_p / t_p)
            # the OpenMP pool goes back to two threads (a team of two makes libgomp release the others): idle pool threads take cores from the host
            # thread of the host-driven GPU loops measured below (fp64 device loop on PolyOpt 2d = 40: 4.5k iterations/s behind a pool of 32, 5.2k without)
            oc.set_num_threads(2)
            oc.solvesdp(maxiterations=1)
            oc.set_num_threads(1)

        # ---- the one rate the reference documents: the solver log of min_f(2) (docs/src/solving.md:38-46) ----
        # iterations 3 -> 56 between log times 13.4 s and 13.9 s at 0.1 s resolution: ~100 iterations/s (+-20 %), hardware and thread count not stated.
        # Same instance (tests/golden/min_f_2.npz: rebuilt from the mathematics of examples/PolyOpt.jl:40-86, its log reproduced digit for digit by
        # tests/test_reference_vectors.py), same options, 5 limbs = prec 256.  NOT the headline workload: reported beside it.
        try:
            import dataclasses
            from clrs_amd.sdp import FlatSDP
            z = np.load(os.path.join(ROOT, "tests", "golden", "min_f_2.npz"), allow_pickle=False)
            mf = FlatSDP(**{fl.name: (z[fl.name] if z[fl.name].ndim else z[fl.name].item()) for fl in dataclasses.fields(FlatSDP) if fl.name in z.files})
            cm = MwSchurContext(mf, limbs=K, device=local_rank)
            rm = solvesdp_mw(mf, ctx=cm, **thr)
            assert rm.error_code == 0 and rm.status == "Optimal" and abs(rm.primal_objective - (-2.112913881423605)) <= 1e-10, (rm.status, rm.primal_objective)
            torch.cuda.synchronize()
            n_m, t1 = 0, time.perf_counter()
            while time.perf_counter() - t1 < 1.0:
                n_m += solvesdp_mw(mf, ctx=cm, **thr).iterations
            torch.cuda.synchronize()
            rate_m = n_m / (time.perf_counter() - t1)
            cm.close()
            out["reference_documented_rate"] = {
                "instance": "min_f(2) (examples/PolyOpt.jl:40-86): 1 cluster, 11 constraints, blocks 4x4 rank-1 + 3x3 rank-2, 1 free variable; prec = 256",
                "reference_iterations_per_s": 100.0, "reference_source": "docs/src/solving.md:41-44: iterations 3 -> 56 between log times 13.4 s and 13.9 s (0.1 s resolution): "
                "~100 iterations/s +-20 %; hardware and thread count not stated; the only timing the reference publishes (BASELINE.md section 1)",
                "gpu_iterations_per_s": rate_m, "gpu_iterations": rm.iterations, "gpu_primal_objective": rm.primal_objective,
                "ratio": rate_m / 100.0}
        except Exception as e:
            out["reference_documented_rate"] = {"error": repr(e)}

        # ---- multi-word Schur assembly on a many-cluster instance: against the fp64 pipe ----
        try:
            from clrs_amd.sdp import replicate_clusters

            def mw_assembly(copies, reps=10):
                big_ = replicate_clusters(flat, copies)
                bctx_ = MwSchurContext(big_, limbs=K, device=local_rank, timing=True)
                tbX, tbY = torch.tensor(np.tile(X, (1, copies)), device=dev), torch.tensor(np.tile(Y, (1, copies)), device=dev)
                tbXc = torch.empty_like(tbX)
                bctx_.cholesky_blocks_dev(tbX.data_ptr(), tbXc.data_ptr())
                for _ in range(3):
                    bctx_.assemble_dev(tbXc.data_ptr(), tbY.data_ptr())
                bctx_.set_timing(False)
                torch.cuda.synchronize()
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s_ = torch.cuda.ExternalStream(bctx_.stream())
                with torch.cuda.stream(s_):
                    ev0.record()
                    for _ in range(reps):
                        bctx_.assemble_dev(tbXc.data_ptr(), tbY.data_ptr())
                    ev1.record()
                ev1.synchronize()
                return big_, bctx_, 1e-3 * ev0.elapsed_time(ev1) / reps, reps

            # the instance that fills the chip (two workgroups of k_mws_pair per compute unit: 512 at a time, 3072 low-rank blocks here) ...
            big, bctx, asm_s, reps = mw_assembly(args.mw_copies)
            muladds = bctx.counters()["assemble_muladds"]
            flops_alg = muladds * K * (K + 1)
            out["roofline_mw"] = {"bound": "mfma", "phase": "schur_assemble (multi-word)", "kernel": "k_mws_pair (exact slice products, v_mfma_f64_16x16x4) + k_mw_dense_t + k_mw_saccum",
                                  "assembly_us": 1e6 * asm_s, "achieved": flops_alg / asm_s / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": flops_alg / asm_s / 1e12 / FP64_PEAK_TFLOPS, "traffic": None,
                                  "workload": f"{big.n_clusters} clusters / {big.n_blocks} PSD blocks of the cohnelkies(8,15) shapes in one assembly, {K} limbs",
                                  "algorithmic_muladds": muladds,
                                  "algorithmic_flops": flops_alg,
                                  "flops_model": f"K(K+1) = {K * (K + 1)} fp64 flops per K-limb multiply-add: the K(K+1)/2 limb products with i + j < K, one multiply and "
                                                 f"one add each -- the count an exact-product scheme needs as well; the error-free transformations that make the "
                                                 f"adds exact are overhead, not algorithmic work (DESIGN.md section 6)",
                                  "timed_region": f"{reps} assemblies back to back between two HIP events on the context stream"}
            # what the launch really executes: fp64 VALU instructions per multiply-add from the committed counter pass (rocprofv3 cannot run
            # inside this process), times this run's multiply-adds, against the issue capacity of the chip's 1024 SIMDs during this run's time
            try:
                pc = json.load(open(os.path.join(ROOT, "profiles", "mw_counters.json")))
                if pc["limbs"] == K:
                    scale = muladds / pc["algorithmic_muladds"]
                    wave_insts = pc["valu_wave_instructions_total"] * scale
                    mfma = sum(pc.get("mfma_wave_instructions", {}).values()) * scale
                    pipe_cycles = (wave_insts - mfma) * 4.0 + mfma * 64.0        # an fp64 MFMA 16x16x4 holds the pipe for 16 passes, any other fp64 VALU instruction for one
                    out["roofline_mw"]["executed"] = {
                        "valu_lane_instructions_per_muladd": pc["lane_instructions_per_muladd"], "source": pc["source"],
                        "before_exact_products": pc.get("before_exact_products"),
                        "fp64_pipe_utilisation": pipe_cycles / (1024 * 2.4e9 * asm_s),
                        "what": "wave instructions per multiply-add from the committed counter pass (MFMAs included, one instruction each); share of the fp64 pipe "
                                "cycles of the chip's 1024 SIMDs (4 per fp64 VALU instruction, 64 per v_mfma_f64_16x16x4, 2.4 GHz) the assembly filled -- the pairing "
                                "matrices come from exact 23-bit slice products on the matrix cores (k_mws_pair), S_j from K-limb expansions (k_mw_saccum)"}
            except Exception as e:
                out["roofline_mw"]["executed_error"] = repr(e)
            bctx.close()
            # ... and the 256-cluster instance of rounds 1-2 (384 low-rank blocks: less than one round of workgroups, the launch is bound by the
            # latency of one workgroup, not by the pipe)
            big2, bctx2, asm2, _ = mw_assembly(128)
            m2 = bctx2.counters()["assemble_muladds"]
            out["roofline_mw"]["instance_of_rounds_1_2"] = {"workload": f"{big2.n_clusters} clusters / {big2.n_blocks} PSD blocks", "assembly_us": 1e6 * asm2,
                                                            "achieved": m2 * K * (K + 1) / asm2 / 1e12, "frac": m2 * K * (K + 1) / asm2 / 1e12 / FP64_PEAK_TFLOPS}
            bctx2.close()
        except Exception as e:
            out.setdefault("roofline_mw", {})["error"] = repr(e)

        # ---- the kernel that dominates the TIMED step: k_mw_factor (one 32 x 32 Cholesky + inverse factor per cluster, 4 workgroups each) ----
        # Its roof is the fp64 issue rate of the compute units it occupies (one fp64 VALU wave instruction holds its SIMD for 4 cycles); the
        # executed instruction count comes from the committed counter pass (rocprofv3 cannot run inside this process), the duration is live.
        try:
            fc = json.load(open(os.path.join(ROOT, "profiles", "mw_factor_counters.json")))
            if fc["limbs"] == K:
                # the factor stage as the TIMED loop runs it: in the reduced limb count of the mixed-precision refinement (clrs_mw_options.factor_limbs; inside
                # clrs_mw_ipm_* that is automatic, the stand-alone entry points this measurement goes through take it on request)
                kf = K - 1 if K in (5, 6) else K
                cf = MwSchurContext(flat, limbs=K, device=local_rank, factor_limbs=kf)

                def fstep():
                    cf.cholesky_blocks_dev(dX.data_ptr(), dXc.data_ptr())
                    cf.assemble_dev(dXc.data_ptr(), dY.data_ptr())
                    cf.factor_dev()
                    cf.solve_dev(drx.data_ptr(), dry.data_ptr(), ddx.data_ptr(), ddy.data_ptr())
                for _ in range(5):
                    fstep()
                cf.set_timing(True)
                for _ in range(5):
                    fstep()
                t_f = cf.timings()[1]                       # chol S_j + L^-1 B of the last pass (HIP events on the context stream)
                cf.set_timing(False)
                tm5 = cf.timings()
                cf.close()
                insts, cus = fc["k_mw_factor"]["SQ_INSTS_VALU"], fc["k_mw_factor"]["workgroups"]
                t_k = fc["k_mw_factor"]["share_of_stage"] * t_f
                peak = cus * 4 * 2.4e9 / 4.0                # fp64 VALU wave instructions per second of the occupied compute units
                kname = fc["k_mw_factor"].get("kernel", "k_mw_factor")
                out["roofline_timed"] = {"kernel": kname, "bound": "fp64 VALU issue of the occupied compute units", "unit": "G wave-instructions/s",
                                         "achieved": insts / t_k / 1e9, "peak": peak / 1e9, "frac": insts / t_k / peak, "traffic": None,
                                         "kernel_us": 1e6 * t_k, "compute_units": cus, "of_256_compute_units": cus / 256.0,
                                         "valu_wave_instructions": insts, "source": fc["source"], "factor_limbs": kf,
                                         "what": "SQ_INSTS_VALU x 4 cycles / (occupied CUs x 4 SIMDs x kernel cycles): the chain of 32 dependent pivot steps of a "
                                                 "fraction-free elimination -- since round 4 a pipeline of column-block stages over workgroups (clrs_mw_pipe.hip.h) -- "
                                                 "keeps this many of 256 compute units busy at this share of their fp64 issue slots",
                                         "other_chains": {}}
                # the two other pivot chains of the timed step: Cholesky of Q (duration live: HIP events of the context), Cholesky of the X blocks (duration of the counter pass)
                for key, t_live in (("k_mw_potrf_q", tm5[4]), ("k_mw_potrf_x", None)):
                    if key in fc:
                        e = fc[key]
                        t_c = t_live if t_live else 1e-6 * e["duration_us"]
                        pk = e["workgroups"] * 4 * 2.4e9 / 4.0
                        out["roofline_timed"]["other_chains"][key] = {"kernel_us": 1e6 * t_c, "duration": "live (HIP events)" if t_live else "counter pass", "compute_units": e["workgroups"],
                                                                      "valu_wave_instructions": e["SQ_INSTS_VALU"], "frac": e["SQ_INSTS_VALU"] / t_c / pk}
        except Exception as e:
            out["roofline_timed"] = {"error": repr(e)}

        # ---- fp64 measurements: the HBM-bound Schur assembly (BASELINE metric) and the fp64 kernels on this problem's shapes ----
        if not args.skip_fp64 and world == 1:
            try:
                sys.path.insert(0, os.path.join(ROOT, "scripts"))
                import bench_fp64
                f64 = bench_fp64.main(["--steps", "500", "--warmup", "100"] + (["--skip-cpu"] if args.skip_cpu else []), emit=False)
                out["roofline"] = f64.get("roofline")
                out["roofline_named"] = f64.get("roofline_named")
                out["roofline_R"] = f64.get("roofline_R")
                out["roofline_dense"] = f64.get("roofline_dense")
                out["roofline_shapes"] = f64.get("roofline_shapes")
                out["fp64"] = {"what": "the fp64 kernels on the same problem SHAPES with synthetic well-conditioned iterates; the factorisation of S_j fails in fp64 "
                                       "(factor_status != 0 = the reference's SolverFailure, src/solver.jl:1249), so this is a kernel cost, not a rate of solvable iterations",
                               "steps_per_s": f64["value"], "ms_per_step": f64["ms_per_step"], "parity": f64["parity"],
                               "cpu_baseline_fp64": f64.get("cpu_baseline"), "full_ipm_device_resident": f64.get("full_ipm_device_resident")}
            except Exception as e:
                out["fp64"] = {"error": repr(e)}
    if sharded:
        dist.barrier()
        ctx.comm_destroy()
    ctx.close()
    if sharded:
        dist.destroy_process_group()
    if rank == 0:
        import ctypes
        sys.stderr.flush()
        ctypes.CDLL(None).fflush(None)
        os.write(json_fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
