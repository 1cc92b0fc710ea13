#!/usr/bin/env python3
"""Generates tests/golden/<name>_ipm.npz and <name>_traj.npz: the WHOLE interior-point loop of the reference restated in mpmath
at the reference's default precision (256 bits), and kernel-level fixtures on real iterates of that run.

Why: the reference (Julia + Arblib/FLINT) cannot run in the build container, and the only thing its tests
hold for BASELINE config 3 is the objective of cohnelkies(8,15): pi^4/384 within 1e-4 at prec = 256
(test/runtests_solver.jl:19-20).  This script reproduces that number with a restatement that shares neither the arithmetic
(mpmath's Python integers; the CPU oracle uses oracle/mpx.hpp, the product fp64 expansions) nor the factorisations of the
product and the oracle: S is formed from DENSE constraint matrices through the trace formula, X^-1 is an LU inverse, the
Newton system [S -B; B^T 0] is solved by one LU decomposition of the full KKT matrix (src/solver.jl:1527 states the system).
It pins (i) the problem generator (clusteredlowranksolver.jl_amd/problems/spherepacking.py), (ii) the oracle's loop
(tests/test_oracle_cpu.py compares objective, iteration count and the mu / step-length trace) and (iii), through the trajectory
fixture, S, dx, dy on iterates with mu from 1e20 down to 1e-15 for the oracle and for the HIP path.

Loop (src/solver.jl:348-589, conventions of SURVEY.md section 3.6): mu = <X,Y>/K; R = mu_p I - XY; residuals P, p, d; predictor;
beta_c; corrector; step lengths alpha = min(-gamma / (lambda_min(L^-1 dM L^-T) - 1e-5), 1) with lambda_min taken in fp64 as the
reference does (Float64 Lanczos, :1659); update; termination :921-950 with the reference's DEFAULT options.

Run from the repo root (several minutes):  python tests/golden/make_golden_ipm.py [name] [prec]
"""
import os
import sys
import time

import mpmath as mp
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.util import flat  # noqa: E402

LIMBS_OUT = 6


def mpv(hi, lo, i):
    return mp.mpf(float(hi[i])) + (mp.mpf(float(lo[i])) if lo is not None else 0)


def to_limbs(vals, k=LIMBS_OUT):
    out = np.zeros((k, len(vals)))
    for i, v in enumerate(vals):
        r = mp.mpf(v)
        for l in range(k):
            h = float(r)
            out[l, i] = h
            r -= mp.mpf(h)
    return out


def dense_A(f, b):
    """{p: n x n list-of-lists} of block b from the (hi, lo) data of the FlatSDP (the same 106-bit data the oracle and the GPU get)."""
    n, dl = int(f.block_n[b]), int(f.block_delta[b])
    out = {}
    if f.block_kind[b] == 0:
        for t in range(int(f.term_ptr[b]), int(f.term_ptr[b + 1])):
            p, r, s = int(f.term_p[t]), int(f.term_r[t]), int(f.term_s[t])
            lam = mpv(f.term_lambda, f.term_lambda_lo, t)
            v0 = int(f.term_vec_ptr[t])
            A = out.setdefault(p, mp.zeros(n, n))
            for i in range(dl):
                vi = lam * mpv(f.term_vs, f.term_vs_lo, v0 + i)
                for k in range(dl):
                    A[r * dl + i, s * dl + k] += vi * mpv(f.term_ws, f.term_ws_lo, v0 + k)
    else:
        for e in range(int(f.dense_ptr[b]), int(f.dense_ptr[b + 1])):
            a0 = int(f.dense_A_ptr[e])
            A = mp.zeros(n, n)
            for col in range(n):
                for row in range(n):
                    A[row, col] = mpv(f.dense_A, f.dense_A_lo, a0 + row + col * n)
            out[int(f.dense_p[e])] = A
    return out


def frob(A, B):
    s = mp.mpf(0)
    for i in range(A.rows):
        for k in range(A.cols):
            s += A[i, k] * B[i, k]
    return s


def sym(M):
    return (M + M.T) / 2


class Problem:
    def __init__(self, f):
        self.f = f
        self.J, self.N, self.NB = f.n_clusters, f.n_free, f.n_blocks
        self.P = [int(p) for p in f.cluster_P]
        self.coff = [int(v) for v in f.cluster_off]
        self.nx = f.x_len
        self.blk_n = [int(v) for v in f.block_n]
        self.blk_j = [int(v) for v in f.block_cluster]
        self.A = [dense_A(f, b) for b in range(self.NB)]
        self.C = []
        for b in range(self.NB):
            n, o = self.blk_n[b], int(f.block_off[b])
            M = mp.zeros(n, n)
            for col in range(n):
                for row in range(n):
                    M[row, col] = mpv(f.C, f.C_lo, o + row + col * n)
            self.C.append(M)
        # B stacked (nx x N), c, b
        self.B = mp.zeros(self.nx, max(self.N, 1))
        off = 0
        for j in range(self.J):
            for a in range(self.N):
                for r in range(self.P[j]):
                    self.B[self.coff[j] + r, a] = mpv(f.B, f.B_lo, off + r + a * self.P[j])
            off += self.P[j] * self.N
        self.c = [mpv(f.c, f.c_lo, i) for i in range(self.nx)]
        self.b = [mpv(f.b, f.b_lo, i) for i in range(self.N)]
        self.sgn = 1 if f.maximize else -1
        self.const = mp.mpf(float(f.constant))
        self.K = sum(self.blk_n)

    def weighted(self, a):
        out = []
        for b in range(self.NB):
            n = self.blk_n[b]
            M = mp.zeros(n, n)
            o = self.coff[self.blk_j[b]]
            for p, Ap in self.A[b].items():
                M += a[o + p] * Ap
            out.append(M)
        return out

    def trace(self, Ms):
        res = [mp.mpf(0)] * self.nx
        for b in range(self.NB):
            o = self.coff[self.blk_j[b]]
            for p, Ap in self.A[b].items():
                res[o + p] += frob(Ap, Ms[b])
        return res

    def schur(self, Xinv, Y):
        S = mp.zeros(self.nx, self.nx)
        for b in range(self.NB):
            o = self.coff[self.blk_j[b]]
            T = {q: Xinv[b] * Aq * Y[b] for q, Aq in self.A[b].items()}
            ps = sorted(self.A[b])
            for p in ps:
                for q in ps:
                    if q < p:
                        continue
                    v = frob(self.A[b][p], T[q])
                    S[o + p, o + q] += v
                    if q != p:
                        S[o + q, o + p] += v
        return S


def min_eig_congruence(M, dM):
    """lambda_min(L^-1 dM L^-T), L = chol(M): the congruence in multi-precision, the eigenvalue in fp64 (src/solver.jl:1644-1662)."""
    n = M.rows
    if n == 1:
        return float(dM[0, 0] / M[0, 0])
    L = mp.cholesky(M)
    Li = mp.inverse(L)
    W = Li * dM * Li.T
    Wd = np.array([[float((W[i, k] + W[k, i]) / 2) for k in range(n)] for i in range(n)])
    return float(np.linalg.eigvalsh(Wd)[0]) - 1e-5


def run(name, prec, snaps):
    mp.mp.prec = prec
    f = flat(name)
    pb = Problem(f)
    nx, N, NB = pb.nx, pb.N, pb.NB
    beta_inf, beta_feas, gamma = mp.mpf(0.3), mp.mpf(0.1), mp.mpf(0.9)       # the doubles, as the keyword defaults are       # src/solver.jl:103-126
    omega = mp.mpf(10) ** 10
    gap_thr, err_thr = mp.mpf(10) ** -15, mp.mpf(10) ** -30
    x = [mp.mpf(0)] * nx
    y = [mp.mpf(0)] * N
    X = [omega * mp.eye(n) for n in pb.blk_n]
    Y = [omega * mp.eye(n) for n in pb.blk_n]
    hist, traj = [], []
    pd_feas = False
    dual_err = primal_err = gap = mp.inf
    d_obj = p_obj = pb.const
    it = 1
    t0 = time.time()
    while True:
        if dual_err < err_thr and primal_err < err_thr and gap < gap_thr:
            break
        if it > 200:
            break
        mu = sum(frob(X[b], Y[b]) for b in range(NB)) / pb.K
        mu_p = mp.mpf(0) if pd_feas else beta_inf * mu
        Xinv = [mp.inverse(Xb) for Xb in X]
        S = pb.schur(Xinv, Y)
        KKT = mp.zeros(nx + N, nx + N)
        for i in range(nx):
            for k in range(nx):
                KKT[i, k] = S[i, k]
            for a in range(N):
                KKT[i, nx + a] = -pb.B[i, a]
                KKT[nx + a, i] = pb.B[i, a]
        LU, perm = mp.mp.LU_decomp(KKT)
        # residuals (src/solver.jl:863-918)
        WA = pb.weighted(x)
        Pm = [WA[b] - X[b] - pb.sgn * pb.C[b] for b in range(NB)]
        trY = pb.trace(Y)
        d = [pb.c[i] - trY[i] - sum(pb.B[i, a] * y[a] for a in range(N)) for i in range(nx)]
        pv = [pb.sgn * pb.b[a] - sum(pb.B[i, a] * x[i] for i in range(nx)) for a in range(N)]
        maxP = max(abs(Pm[b][i, k]) for b in range(NB) for i in range(pb.blk_n[b]) for k in range(pb.blk_n[b]))
        dual_err = max([maxP] + [abs(v) for v in pv])
        primal_err = max(abs(v) for v in d)
        xy = mu * pb.K

        def direction(R):
            Z = [sym(Xinv[b] * (Pm[b] * Y[b] - R[b])) for b in range(NB)]
            trZ = pb.trace(Z)
            rhs = mp.matrix([-d[i] - trZ[i] for i in range(nx)] + pv)
            sol = mp.mp.U_solve(LU, mp.mp.L_solve(LU, rhs, perm))
            dx, dy = [sol[i] for i in range(nx)], [sol[nx + a] for a in range(N)]
            WAd = pb.weighted(dx)
            dX = [WAd[b] + Pm[b] for b in range(NB)]
            dY = [sym(Xinv[b] * (R[b] - dX[b] * Y[b])) for b in range(NB)]
            return dx, dy, dX, dY, rhs

        R = [mu_p * mp.eye(pb.blk_n[b]) - X[b] * Y[b] for b in range(NB)]
        dx, dy, dX, dY, rhs_pred = direction(R)
        if it in snaps:
            sol = mp.mp.U_solve(LU, mp.mp.L_solve(LU, rhs_pred, perm))
            traj.append(dict(it=it, mu=mu, X=[Xb.copy() for Xb in X], Y=[Yb.copy() for Yb in Y], rhs=rhs_pred.copy(), S=S.copy(), sol=sol))
        r = (xy + sum(frob(X[b], dY[b]) + frob(dX[b], Y[b]) + frob(dX[b], dY[b]) for b in range(NB))) / (mu * pb.K)
        beta = r * r if r < 1 else r
        beta_c = min(max(beta_feas, beta), mp.mpf(1)) if pd_feas else max(beta_inf, beta)        # :429-434 (previous feasibility)
        mu_c = beta_c * mu
        pd_feas = dual_err < err_thr and primal_err < err_thr                                       # :441-447
        R = [mu_c * mp.eye(pb.blk_n[b]) - X[b] * Y[b] - dX[b] * dY[b] for b in range(NB)]
        dx, dy, dX, dY, _ = direction(R)

        def step(Ms, dMs):
            mn = min(min_eig_congruence(Ms[b], dMs[b]) for b in range(NB))
            if mn > -float(gamma):
                return mp.mpf(1)
            return -gamma / mp.mpf(mn)
        alpha_d, alpha_p = step(X, dX), step(Y, dY)
        hist.append([it, float(mu), float(d_obj), float(p_obj), float(gap), float(maxP), float(max([abs(v) for v in pv] + [0])), float(primal_err),
                     float(alpha_d), float(alpha_p), float(beta_c)])
        print("%3d %7.1fs mu %.4e dobj %.6e pobj %.6e gap %.2e derr %.2e perr %.2e ad %.4f ap %.4f" %
              (it, time.time() - t0, float(mu), float(d_obj), float(p_obj), float(gap), float(dual_err), float(primal_err), float(alpha_d), float(alpha_p)), flush=True)
        if min(alpha_d, alpha_p) < mp.mpf(10) ** -7:
            break
        if pd_feas:
            alpha_d = alpha_p = min(alpha_d, alpha_p)                                              # safe_step, :480-483
        x = [x[i] + alpha_d * dx[i] for i in range(nx)]
        y = [y[a] + alpha_p * dy[a] for a in range(N)]
        X = [X[b] + alpha_d * dX[b] for b in range(NB)]
        Y = [Y[b] + alpha_p * dY[b] for b in range(NB)]
        d_obj = pb.sgn * sum(pb.c[i] * x[i] for i in range(nx)) + pb.const
        p_obj = sum(frob(pb.C[b], Y[b]) for b in range(NB)) + sum(pb.b[a] * y[a] for a in range(N)) + pb.const
        gap = abs(d_obj - p_obj) / max(mp.mpf(1), abs(d_obj + p_obj))
        it += 1
    return f, pb, dict(iterations=it - 1, d_obj=d_obj, p_obj=p_obj, gap=gap, dual_err=dual_err, primal_err=primal_err, hist=np.array(hist)), traj


def flatten_blocks(f, Ms):
    vals = []
    for b, M in enumerate(Ms):
        n = M.rows
        vals += [M[i, k] for k in range(n) for i in range(n)]      # column-major
    return vals


def refresh_expected(name):
    """Recompute S, dx, dy of an existing trajectory fixture from its stored iterates (no new interior-point run)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"{name}_traj.npz")
    g = dict(np.load(path))
    prec = int(g["prec"][0])
    mp.mp.prec = prec + 200
    f = flat(name)
    pb = Problem(f)
    nx, N = pb.nx, pb.N

    def val(a, i):
        return mp.fsum(mp.mpf(float(a[l, i])) for l in range(a.shape[0]))
    Ss, dxs, dys = [], [], []
    for s_ in range(len(g["iters"])):
        Xb, Yb = [], []
        for b in range(pb.NB):
            n, o = pb.blk_n[b], int(f.block_off[b])
            Xb.append(mp.matrix([[val(g["X"][s_], o + i + k * n) for k in range(n)] for i in range(n)]))
            Yb.append(mp.matrix([[val(g["Y"][s_], o + i + k * n) for k in range(n)] for i in range(n)]))
        S = pb.schur([mp.inverse(M) for M in Xb], Yb)
        Sl = []
        for j in range(pb.J):
            o, P = pb.coff[j], pb.P[j]
            Sl += [S[o + p, o + q] for q in range(P) for p in range(P)]
        Ss.append(to_limbs(Sl))
        KKT = mp.zeros(nx + N, nx + N)
        for i in range(nx):
            for k in range(nx):
                KKT[i, k] = S[i, k]
            for a in range(N):
                KKT[i, nx + a] = -pb.B[i, a]
                KKT[nx + a, i] = pb.B[i, a]
        rhs = mp.matrix([val(g["rhs_x"][s_], i) for i in range(nx)] + [val(g["rhs_y"][s_], a) for a in range(N)])
        sol = mp.lu_solve(KKT, rhs)
        dxs.append(to_limbs([sol[i] for i in range(nx)]))
        dys.append(to_limbs([sol[nx + a] for a in range(N)]))
        print("refreshed snapshot", s_, flush=True)
    g.update(S=np.array(Ss), dx=np.array(dxs), dy=np.array(dys))
    np.savez_compressed(path, **g)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--refresh-expected":
        refresh_expected(sys.argv[2])
        sys.exit(0)
    name = sys.argv[1] if len(sys.argv) > 1 else "ce_8_15"
    prec = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    snaps = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 28, 55]
    f, pb, res, traj = run(name, prec, set(snaps))
    out = os.path.dirname(os.path.abspath(__file__))
    print("iterations", res["iterations"], "d_obj", mp.nstr(res["d_obj"], 30), "p_obj", mp.nstr(res["p_obj"], 30), "gap", mp.nstr(res["gap"], 5))
    np.savez_compressed(os.path.join(out, f"{name}_ipm.npz"), prec=np.array([prec]), iterations=np.array([res["iterations"]]),
                        d_obj=to_limbs([res["d_obj"]]), p_obj=to_limbs([res["p_obj"]]), gap=np.array([float(res["gap"])]),
                        dual_error=np.array([float(res["dual_err"])]), primal_error=np.array([float(res["primal_err"])]), hist=res["hist"])
    if traj:
        mp.mp.prec = prec + 200                                    # expected values well beyond the precision under test
        pb = Problem(f)                                            # the dense A_p again, now exact products of the (hi, lo) data
        nx, N = pb.nx, pb.N
        pack = dict(iters=np.array([t["it"] for t in traj]), mu=np.array([float(t["mu"]) for t in traj]), prec=np.array([prec]))
        Xs, Ys, rxs, rys, Ss, dxs, dys = [], [], [], [], [], [], []
        for t in traj:
            Xs.append(to_limbs(flatten_blocks(f, t["X"])))
            Ys.append(to_limbs(flatten_blocks(f, t["Y"])))
            rxs.append(to_limbs([t["rhs"][i] for i in range(nx)]))
            rys.append(to_limbs([t["rhs"][nx + a] for a in range(N)]))
            # expected S (S layout) and the solution of the KKT system, recomputed at the higher precision from the stored iterate
            Xinv = [mp.inverse(Xb) for Xb in t["X"]]
            S = pb.schur(Xinv, t["Y"])
            Sl = []
            for j in range(pb.J):
                o, P = pb.coff[j], pb.P[j]
                Sl += [S[o + p, o + q] for q in range(P) for p in range(P)]
            Ss.append(to_limbs(Sl))
            KKT = mp.zeros(nx + N, nx + N)
            for i in range(nx):
                for k in range(nx):
                    KKT[i, k] = S[i, k]
                for a in range(N):
                    KKT[i, nx + a] = -pb.B[i, a]
                    KKT[nx + a, i] = pb.B[i, a]
            sol = mp.lu_solve(KKT, t["rhs"])
            dxs.append(to_limbs([sol[i] for i in range(nx)]))
            dys.append(to_limbs([sol[nx + a] for a in range(N)]))
        pack.update(X=np.array(Xs), Y=np.array(Ys), rhs_x=np.array(rxs), rhs_y=np.array(rys), S=np.array(Ss), dx=np.array(dxs), dy=np.array(dys))
        np.savez_compressed(os.path.join(out, f"{name}_traj.npz"), **pack)
        print("trajectory fixture:", {k: v.shape for k, v in pack.items()})
