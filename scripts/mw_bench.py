"""Timing of the multi-word path on one GPU against the multi-precision CPU oracle on the same host (diagnostic; bench.py
carries the contractual numbers).  usage: mw_bench.py [instance] [limbs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests.util import flat, mw_from_double
from clrs_amd.mw import MwSchurContext, solvesdp_mw
from oracle.oracle import Oracle

name = sys.argv[1] if len(sys.argv) > 1 else "ce_8_15"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
f = flat(name)
bits = {2: 106, 3: 160, 4: 212, 5: 256}[K]

# ---- whole solve
t0 = time.time(); r = solvesdp_mw(f, limbs=K, dual_error_threshold=1e-30 if K == 5 else 1e-25, primal_error_threshold=1e-30 if K == 5 else 1e-25,
                                  duality_gap_threshold=1e-15 if K == 5 else 1e-12); t1 = time.time()
r2 = solvesdp_mw(f, limbs=K, dual_error_threshold=1e-30 if K == 5 else 1e-25, primal_error_threshold=1e-30 if K == 5 else 1e-25,
                 duality_gap_threshold=1e-15 if K == 5 else 1e-12)
print("GPU solve   %s K=%d: %s code %d, %d iterations, %.3f s (second run %.3f s = %.2f ms/iteration), objective %.12f" %
      (name, K, r.status, r.error_code, r.iterations, t1 - t0, r2.time_total, 1e3 * r2.time_total / max(r2.iterations, 1), r.primal_objective))
o = Oracle(f, mp_bits=bits)
for nt in (1, os.cpu_count()):
    o.set_num_threads(nt)
    t0 = time.time(); ro = o.solvesdp(); t1 = time.time()
    print("CPU oracle  %d bits, %d threads: code %d, %d iterations, %.3f s = %.2f ms/iteration" % (bits, nt, ro["error_code"], ro["iterations"], t1 - t0, 1e3 * (t1 - t0) / max(ro["iterations"], 1)))

# ---- hot-path step on a mid-trajectory iterate
its = [max(2, ro["iterations"] // 2)]
o.set_num_threads(1)
rs = o.solvesdp(snapshots=its, snapshot_limbs=K)
X, Y, rx, ry = rs["snap"]["X"][0], rs["snap"]["Y"][0], rs["snap"]["rhs_x"][0], rs["snap"]["rhs_y"][0]
ctx = MwSchurContext(f, limbs=K, timing=True)
dev = torch.device("cuda:0")
dX, dY = torch.tensor(X, device=dev), torch.tensor(Y, device=dev)
dXc = torch.empty_like(dX); drx = torch.tensor(rx, device=dev); dry = torch.tensor(ry if f.n_free else np.zeros((K, 1)), device=dev)
ddx = torch.empty_like(drx); ddy = torch.empty_like(dry)
torch.cuda.synchronize()
def step():
    ctx.cholesky_blocks_dev(dX.data_ptr(), dXc.data_ptr())
    ctx.assemble_dev(dXc.data_ptr(), dY.data_ptr())
    ctx.factor_dev()
    ctx.solve_dev(drx.data_ptr(), dry.data_ptr() if f.n_free else 0, ddx.data_ptr(), ddy.data_ptr() if f.n_free else 0)
    ctx.solve_dev(drx.data_ptr(), dry.data_ptr() if f.n_free else 0, ddx.data_ptr(), ddy.data_ptr() if f.n_free else 0)
for _ in range(5): step()
assert ctx.sync_status() == 0 and ctx.sync_status_cholesky() == 0
n = 50
t0 = time.time()
for _ in range(n): step()
ctx.sync_status()
t = (time.time() - t0) / n
tm = ctx.timings()
print("GPU hot-path step (chol X + assemble + factor + 2 solves), iterate of iteration %d: %.1f us; last-call stage times (us): schur %.1f, cholS+LinvB %.1f, Q %.1f, cholQ %.1f, solve %.1f"
      % (its[0], 1e6 * t, *(1e6 * tm[i] for i in (0, 1, 3, 4, 5))))
# the same step in the oracle
for nt in (1, os.cpu_count()):
    o.set_num_threads(nt)
    reps = 3
    t0 = time.time()
    for _ in range(reps):
        st, Xc = o.cholesky_blocks_mw(X)
        o.schur_assemble_mw(Xc, Y)
        assert o.schur_factor() == 0
        o.schur_solve_mw(rx, ry); o.schur_solve_mw(rx, ry)
    tc = (time.time() - t0) / reps
    print("CPU oracle hot-path step, %d bits, %d threads: %.2f ms  -> GPU/CPU = %.1fx" % (bits, nt, 1e3 * tc, tc / t))
ctx.close()
