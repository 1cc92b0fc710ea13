"""Host-side mirror of the reference's interface for the hot path, driving the HIP library.

Names, argument meaning and error behaviour follow the reference (src/solver.jl); Python cannot
spell `!`, so `compute_T_decomposition!` is `compute_T_decomposition` etc.

    ctx = precompute_matrices_bilinear_pairings(sdp)         # src/solver.jl:985-1059 (+ prealloc :298-317)
    X_inv = ctx.cholesky_blocks(X)                           # :388-399   (X_inv is the Cholesky factor, as in the reference)
    times = compute_T_decomposition(ctx, X_inv, Y)           # :1229-1287 (Schur assembly + factorisation)
    dx, dy = solve_system(ctx, rhs_x, rhs_y)                 # :1527-1582 (solve stage of compute_search_direction!)
    status, primal, dual, t, code = solvesdp(sdp, ...)       # :100-744   (host orchestration, fp64)

All compute on the path goes through libclrs_hip.so (`_lib.load()` raises when it is missing);
there is no CPU fallback.  The orchestration around the path (residuals, search-direction assembly,
step length) stays on the host, as BASELINE.json's north_star prescribes ("Host orchestration stays
in Julia"); here it is numpy fp64.
"""
from __future__ import annotations

import ctypes as C
import time
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

from . import _lib
from .sdp import ClusteredLowRankSDP, FlatSDP, flatten


class SolverFailure(Exception):
    """reference: struct SolverFailure (src/solver.jl:1-10), thrown when a Cholesky factorisation fails."""


def _dp(a):
    return a.ctypes.data_as(_lib.p_d)


def _c(a, dt=np.float64):
    return np.ascontiguousarray(a, dtype=dt)


class SchurContext:
    """Device context of the hot path for one SDP (one per GPU / process).

    Holds what `precompute_matrices_bilinear_pairings` returns in the reference (de-duplicated
    left/right vector tables, pointer tables, `high_ranks`) plus every preallocated buffer of
    src/solver.jl:298-317 -- all device resident."""

    def __init__(self, sdp, device: int = 0, graph: bool = False, timing: bool = False, fused: Optional[bool] = None,
                 wave: Optional[bool] = None, wave2: Optional[bool] = None, wave3: Optional[bool] = None, wave4: Optional[bool] = None, wave5: Optional[bool] = None,
                 solve_small2: Optional[bool] = None, factor_small: Optional[int] = None,
                 split_blocks: Optional[bool] = None):
        """`fused=False` forces the staged grouped-GEMM / blocked-BLAS path everywhere (default: clusters that fit in
        one CU's LDS take the fused per-cluster assembly, factor and solve kernels).  `wave=False` keeps the
        fused assembly on the 4-waves-per-block kernel even where the wave-per-block kernel applies.  `wave2=True`
        takes the cluster-per-wave assembly even for few clusters (default: from 64 clusters on); `wave3=False` keeps it on
        the LDS-staged kernel (k_cluster_assemble_w2) instead of the register-resident one (k_cluster_assemble_w3);
        `wave4=False` keeps simple blocks of 17-32 rows / clusters of 33-64 constraints off k_cluster_assemble_w4 (on the general kernels);
        `solve_small2=False` keeps the one-launch solve stage on k_solve_small instead of k_solve_small2; `factor_small`
        (0 / 1 / 2) selects when the factorisation stage is the single launch k_factor_small (include/clrs_hip.h);
        `split_blocks=False` keeps the general fused assembly at one workgroup per cluster (default: one per PSD block when
        there are few clusters)."""
        self.flat: FlatSDP = sdp if isinstance(sdp, FlatSDP) else flatten(sdp)
        f = self.flat
        self.L = _lib.load()
        if fused is not None:
            _lib.check(self.L.clrs_config_set(b"fused_assemble", int(bool(fused))))
            _lib.check(self.L.clrs_config_set(b"fused_factor", int(bool(fused))))
        if wave is not None:
            _lib.check(self.L.clrs_config_set(b"wave_assemble", int(bool(wave))))
        if wave2 is not None:
            _lib.check(self.L.clrs_config_set(b"wave2_assemble", 2 if wave2 else 0))
        if wave3 is not None:
            _lib.check(self.L.clrs_config_set(b"wave3_assemble", int(bool(wave3))))
        if wave4 is not None:
            _lib.check(self.L.clrs_config_set(b"wave4_assemble", int(bool(wave4))))
        if wave5 is not None:
            _lib.check(self.L.clrs_config_set(b"wave5_assemble", int(bool(wave5))))
        if solve_small2 is not None:
            _lib.check(self.L.clrs_config_set(b"solve_small2", int(bool(solve_small2))))
        if factor_small is not None:
            _lib.check(self.L.clrs_config_set(b"factor_small", int(factor_small)))
        if split_blocks is not None:
            _lib.check(self.L.clrs_config_set(b"split_blocks", int(bool(split_blocks))))
        k = self._keep = {}

        def hold(name, arr, dt):
            k[name] = _c(arr, dt)
            return k[name]

        d = _lib.SdpDesc()
        d.n_clusters, d.n_free, d.n_blocks = f.n_clusters, f.n_free, f.n_blocks
        d.cluster_P = hold("cluster_P", f.cluster_P, np.int32).ctypes.data_as(_lib.p_i32)
        d.B = _dp(hold("B", f.B, np.float64))
        for name in ("block_cluster", "block_m", "block_delta", "block_kind", "term_p", "term_r", "term_s", "term_rank", "dense_p"):
            setattr(d, name, hold(name, getattr(f, name), np.int32).ctypes.data_as(_lib.p_i32))
        for name in ("term_ptr", "term_vec_ptr", "dense_ptr", "dense_A_ptr"):
            setattr(d, name, hold(name, getattr(f, name), np.int64).ctypes.data_as(_lib.p_i64))
        for name in ("term_lambda", "term_vs", "term_ws", "dense_A"):
            setattr(d, name, _dp(hold(name, getattr(f, name), np.float64)))
        h = C.c_void_p()
        try:
            _lib.check(self.L.clrs_ctx_create(C.byref(d), int(device), C.byref(h)))
        finally:
            if fused is not None:
                self.L.clrs_config_set(b"fused_assemble", 1)
                self.L.clrs_config_set(b"fused_factor", 1)
            if wave is not None:
                self.L.clrs_config_set(b"wave_assemble", 1)
            if wave2 is not None:
                self.L.clrs_config_set(b"wave2_assemble", 1)
            if wave3 is not None:
                self.L.clrs_config_set(b"wave3_assemble", 1)
            if wave4 is not None:
                self.L.clrs_config_set(b"wave4_assemble", 1)
            if wave5 is not None:
                self.L.clrs_config_set(b"wave5_assemble", 1)
            if solve_small2 is not None:
                self.L.clrs_config_set(b"solve_small2", 1)
            if factor_small is not None:
                self.L.clrs_config_set(b"factor_small", 1)
            if split_blocks is not None:
                self.L.clrs_config_set(b"split_blocks", 1)
        self.h = h
        self.device = device
        if graph:
            self.set_graph_mode(True)
        if timing:
            self.set_timing(True)

    def close(self):
        if getattr(self, "h", None):
            self.L.clrs_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- introspection ---------------------------------------------------------------------------
    def dims(self):
        dm = _lib.Dims()
        _lib.check(self.L.clrs_get_dims(self.h, C.byref(dm)))
        return dm

    def unique_counts(self, block: int) -> Tuple[List[int], List[int]]:
        """Sizes of rightvecs[j][l][r] / leftvecs[j][l][r] after de-duplication (src/solver.jl:1018-1051)."""
        R, Lc = [], []
        for r in range(int(self.flat.block_m[block])):
            a, b = C.c_int32(), C.c_int32()
            _lib.check(self.L.clrs_get_unique_counts(self.h, block, r, C.byref(a), C.byref(b)))
            R.append(a.value); Lc.append(b.value)
        return R, Lc

    def fused_clusters(self) -> int:
        """Number of clusters assembled by the fused per-cluster kernel."""
        return int(self.L.clrs_fused_clusters(self.h))

    def wave_clusters(self) -> int:
        """Number of clusters assembled with one wave per PSD block (k_cluster_assemble_w1)."""
        return int(self.L.clrs_wave_clusters(self.h))

    def wave2_clusters(self) -> int:
        """Number of clusters assembled by one wave per cluster with S in registers (k_cluster_assemble_w2)."""
        return int(self.L.clrs_wave2_clusters(self.h))

    def wave5_clusters(self) -> int:
        """Number of clusters assembled by k_cluster_assemble_w5 (2 x 2 blocks of 16 x 16 sub-blocks on shared sample vectors)."""
        return int(self.L.clrs_wave5_clusters(self.h))

    def wave4_clusters(self) -> int:
        """Number of clusters assembled by k_cluster_assemble_w4 (simple blocks of up to 32 rows, up to 64 constraints)."""
        return int(self.L.clrs_wave4_clusters(self.h))

    def high_ranks(self) -> List[bool]:
        """`high_ranks[j][l]` flattened over blocks (src/solver.jl:1000)."""
        return [bool(v) for v in self.flat.block_kind]

    def set_graph_mode(self, on: bool):
        _lib.check(self.L.clrs_set_graph_mode(self.h, int(on)))

    def set_timing(self, on: bool):
        _lib.check(self.L.clrs_set_timing(self.h, int(on)))

    def timings(self) -> np.ndarray:
        t = np.zeros(6)
        _lib.check(self.L.clrs_get_timings(self.h, _dp(t)))
        return t

    def counters(self):
        v = [C.c_double() for _ in range(4)]
        _lib.check(self.L.clrs_get_counters(self.h, *[C.byref(x) for x in v]))
        return dict(assemble_bytes=v[0].value, assemble_flops=v[1].value, factor_flops=v[2].value, solve_flops=v[3].value)

    def set_stream(self, hip_stream: int):
        """Run on the caller's hipStream_t (e.g. torch.cuda.current_stream().cuda_stream)."""
        _lib.check(self.L.clrs_set_stream(self.h, C.c_void_p(hip_stream)))

    def set_kernel_timing(self, kind: int):
        """HIP-event timing around every launch of kernel `kind` (-1 all kinds, -2 off); eager mode only."""
        _lib.check(self.L.clrs_set_kernel_timing(self.h, int(kind)))

    def kernel_times(self) -> dict:
        """{kernel name: (kind, total seconds, launches)} since the last set_kernel_timing."""
        sec = np.zeros(64)
        cnt = np.zeros(64, dtype=np.int64)
        n = _lib.check(self.L.clrs_get_kernel_times(self.h, 64, _dp(sec), cnt.ctypes.data_as(_lib.p_i64)))
        return {self.L.clrs_kernel_name(k).decode(): (k, float(sec[k]), int(cnt[k])) for k in range(min(n, 64)) if cnt[k]}

    def plan_info(self):
        v = [C.c_int32() for _ in range(3)]
        _lib.check(self.L.clrs_plan_info(self.h, *[C.byref(x) for x in v]))
        return dict(assemble=v[0].value, factor=v[1].value, solve=v[2].value)

    # -- the path --------------------------------------------------------------------------------
    def cholesky_blocks(self, X: np.ndarray) -> np.ndarray:
        """approx_cholesky!(X_inv_blk, X_blk) for every block (src/solver.jl:388-399)."""
        f = self.flat
        X = _c(X)
        out = np.empty(f.xy_len)
        st = _lib.check(self.L.clrs_cholesky_blocks(self.h, _dp(X), _dp(out)))
        if st > 0:
            b = st - 1
            j = int(f.block_cluster[b])
            l = b - int(np.searchsorted(f.block_cluster, j))
            raise SolverFailure(f"The cholesky decomposition of X was not computed correctly in block ({j + 1},{l + 1}). "
                                f"Try again with higher precision")
        return out

    def compute_S_integrated(self, X_inv: np.ndarray, Y: np.ndarray, want_S: bool = True, want_AY: bool = True):
        """compute_S_integrated! (src/solver.jl:1062-1226).  `X_inv` holds the Cholesky factors of X."""
        f = self.flat
        S = np.empty(f.S_len) if want_S else None
        AY = np.empty(f.n_terms) if want_AY else None
        _lib.check(self.L.clrs_schur_assemble(self.h, _dp(_c(X_inv)), _dp(_c(Y)), _dp(S) if want_S else None,
                                              _dp(AY) if (want_AY and f.n_terms) else None))
        return S, AY

    def factor(self) -> int:
        return _lib.check(self.L.clrs_schur_factor(self.h))

    def get_factor(self):
        f = self.flat
        Lf = np.empty(f.S_len); LinvB = np.empty(f.x_len * f.n_free); LQ = np.empty(f.n_free * f.n_free)
        _lib.check(self.L.clrs_get_factor(self.h, _dp(Lf), _dp(LinvB) if f.n_free else None, _dp(LQ) if f.n_free else None))
        return Lf, LinvB, LQ

    def solve(self, rhs_x: np.ndarray, rhs_y: np.ndarray):
        f = self.flat
        dx = np.empty(f.x_len); dy = np.empty(max(f.n_free, 1))
        ry = _c(rhs_y) if f.n_free else np.zeros(1)
        _lib.check(self.L.clrs_schur_solve(self.h, _dp(_c(rhs_x)), _dp(ry), _dp(dx), _dp(dy)))
        return dx, dy[:f.n_free]

    # -- device-pointer / split-phase API (used by bench.py and the sharded driver) ---------------
    def cholesky_blocks_dev(self, d_X: int, d_Xchol: int):
        _lib.check(self.L.clrs_cholesky_blocks_dev(self.h, C.c_void_p(d_X), C.c_void_p(d_Xchol)))

    def assemble_dev(self, d_Xchol: int, d_Y: int):
        _lib.check(self.L.clrs_schur_assemble_dev(self.h, C.c_void_p(d_Xchol), C.c_void_p(d_Y)))

    def factor_dev(self):
        _lib.check(self.L.clrs_schur_factor_dev(self.h))

    def solve_dev(self, d_rhs_x: int, d_rhs_y: int, d_dx: int, d_dy: int):
        _lib.check(self.L.clrs_schur_solve_dev(self.h, C.c_void_p(d_rhs_x), C.c_void_p(d_rhs_y) if d_rhs_y else None,
                                               C.c_void_p(d_dx), C.c_void_p(d_dy) if d_dy else None))

    def factor_local_dev(self):
        _lib.check(self.L.clrs_schur_factor_local_dev(self.h))

    def factor_finish_dev(self):
        _lib.check(self.L.clrs_schur_factor_finish_dev(self.h))

    def solve_fwd_dev(self, d_rhs_x: int):
        _lib.check(self.L.clrs_schur_solve_fwd_dev(self.h, C.c_void_p(d_rhs_x)))

    def solve_bwd_dev(self, d_rhs_y: int, d_dx: int, d_dy: int):
        _lib.check(self.L.clrs_schur_solve_bwd_dev(self.h, C.c_void_p(d_rhs_y) if d_rhs_y else None,
                                                   C.c_void_p(d_dx) if d_dx else None, C.c_void_p(d_dy) if d_dy else None))

    def sync_status(self) -> int:
        return _lib.check(self.L.clrs_sync_status(self.h))

    def sync_status_cholesky(self) -> int:
        return _lib.check(self.L.clrs_sync_status_cholesky(self.h))

    def q_buffer(self) -> int:
        return int(self.L.clrs_q_buffer_dev(self.h) or 0)

    def u_buffer(self) -> int:
        return int(self.L.clrs_u_buffer_dev(self.h) or 0)

    def S_buffer(self) -> int:
        return int(self.L.clrs_S_buffer_dev(self.h) or 0)

    def stream(self) -> int:
        return int(self.L.clrs_stream(self.h) or 0)


# ------------------------------------------------------------------------------------------------
# functions with the reference's names
# ------------------------------------------------------------------------------------------------

def precompute_matrices_bilinear_pairings(sdp, device: int = 0, **kw) -> SchurContext:
    """src/solver.jl:985-1059: de-duplicate the sampled vectors and build the pointer tables (on the device)."""
    return SchurContext(sdp, device=device, **kw)


def _raise_factor_failure(ctx: SchurContext, st: int):
    J = ctx.flat.n_clusters
    if st == J + 1:
        raise SolverFailure("Q was not decomposed correctly. Try restarting with a higher precision. If this occurred in the "
                            "first iteration, remove linear dependencies between free variables or turn preprocessing on.")
    raise SolverFailure(f"S was not decomposed succesfully in block {st}, try again with higher precision. If this occurred in "
                        f"the first iteration, remove linear dependencies in the PSD part of the constraints or turn preprocessing on.")


def compute_T_decomposition(ctx: SchurContext, X_inv: np.ndarray, Y: np.ndarray, want_S: bool = False, want_AY: bool = True):
    """compute_T_decomposition! (src/solver.jl:1229-1287): Schur assembly, Cholesky of S_j, L^-1 B, Q, Cholesky of Q.
    Returns ((time_schur, time_cholS, time_LinvB, time_Q, time_cholQ), S, A_Y); raises SolverFailure like the reference."""
    S, AY = ctx.compute_S_integrated(X_inv, Y, want_S=want_S, want_AY=want_AY)
    st = ctx.factor()
    if st > 0:
        _raise_factor_failure(ctx, st)
    t = ctx.timings()
    return tuple(t[:5]), S, AY


def solve_system(ctx: SchurContext, rhs_x: np.ndarray, rhs_y: np.ndarray):
    """The 'solve system' stage of compute_search_direction! (src/solver.jl:1527-1582)."""
    return ctx.solve(rhs_x, rhs_y)


# ------------------------------------------------------------------------------------------------
# host orchestration around the path (numpy fp64): src/solver.jl:100-744
# ------------------------------------------------------------------------------------------------

class _HostBlocks:
    """Per-block numpy views of the constraint data, for the off-path pieces that stay on the host
    (compute_weighted_A! :1410-1470, trace_A :1290-1407)."""

    def __init__(self, f: FlatSDP):
        self.f = f
        self.blocks = []
        for b in range(f.n_blocks):
            n, m, dl = int(f.block_n[b]), int(f.block_m[b]), int(f.block_delta[b])
            j = int(f.block_cluster[b])
            info = dict(n=n, m=m, dl=dl, j=j, kind=int(f.block_kind[b]), off=int(f.block_off[b]))
            if info["kind"] == 0:
                t0, t1 = int(f.term_ptr[b]), int(f.term_ptr[b + 1])
                v0 = int(f.term_vec_ptr[t0])
                V = f.term_vs[v0:v0 + (t1 - t0) * dl].reshape(t1 - t0, dl)
                W = f.term_ws[v0:v0 + (t1 - t0) * dl].reshape(t1 - t0, dl)
                p, r, s, lam = f.term_p[t0:t1], f.term_r[t0:t1], f.term_s[t0:t1], f.term_lambda[t0:t1]
                groups = []
                for rr in range(m):
                    for ss in range(rr + 1):
                        sel = np.nonzero((r == rr) & (s == ss))[0]
                        if sel.size:
                            groups.append((rr, ss, sel, V[sel], W[sel], lam[sel], p[sel].astype(np.int64)))
                info.update(t0=t0, t1=t1, groups=groups)
            else:
                d0, d1 = int(f.dense_ptr[b]), int(f.dense_ptr[b + 1])
                a0 = int(f.dense_A_ptr[d0])
                A = f.dense_A[a0:a0 + (d1 - d0) * n * n].reshape(d1 - d0, n * n)   # rows = vec(A_p), column-major
                info.update(p=f.dense_p[d0:d1].astype(np.int64), A=A)
            self.blocks.append(info)

    def weighted_A(self, a: np.ndarray) -> np.ndarray:
        """sum_i a_i A_i in the xy layout (compute_weighted_A!, src/solver.jl:1410-1470)."""
        f = self.f
        out = np.zeros(f.xy_len)
        for k in self.blocks:
            n, dl = k["n"], k["dl"]
            aj = a[f.cluster_off[k["j"]]:f.cluster_off[k["j"] + 1]]
            if k["kind"] != 0:
                if len(k["p"]):
                    out[k["off"]:k["off"] + n * n] = aj[k["p"]] @ k["A"]
                continue
            M = np.zeros((n, n))
            for (rr, ss, sel, V, W, lam, p) in k["groups"]:
                M[rr * dl:(rr + 1) * dl, ss * dl:(ss + 1) * dl] += (V * (aj[p] * lam)[:, None]).T @ W
            if k["m"] > 1:
                il = np.tril_indices(n, -1)
                M.T[il] = M[il]
            out[k["off"]:k["off"] + n * n] = M.reshape(-1, order="F")
        return out

    def trace_A(self, Z: np.ndarray) -> np.ndarray:
        """<A_*, Z> for symmetric block-diagonal Z in the xy layout (trace_A, src/solver.jl:1290-1366)."""
        f = self.f
        res = np.zeros(f.x_len)
        for k in self.blocks:
            n, dl = k["n"], k["dl"]
            rj = res[f.cluster_off[k["j"]]:f.cluster_off[k["j"] + 1]]
            Zb = Z[k["off"]:k["off"] + n * n]
            if k["kind"] != 0:
                if len(k["p"]):
                    np.add.at(rj, k["p"], k["A"] @ Zb)
                continue
            Zm = Zb.reshape(n, n, order="F")
            for (rr, ss, sel, V, W, lam, p) in k["groups"]:
                Zrs = Zm[rr * dl:(rr + 1) * dl, ss * dl:(ss + 1) * dl]
                val = np.einsum("ti,ij,tj->t", W, Zrs, V) * lam
                if rr != ss:
                    val = 2.0 * val
                np.add.at(rj, p, val)
        return res

    def trace_A_from_AY(self, Y: np.ndarray, AY: np.ndarray) -> np.ndarray:
        """trace_A(sdp, (Y, A_Y), ...) (src/solver.jl:1368-1407): reuse the pairings w^T Y v of the assembly."""
        f = self.f
        res = np.zeros(f.x_len)
        for k in self.blocks:
            n = k["n"]
            rj = res[f.cluster_off[k["j"]]:f.cluster_off[k["j"] + 1]]
            if k["kind"] != 0:
                if len(k["p"]):
                    np.add.at(rj, k["p"], k["A"] @ Y[k["off"]:k["off"] + n * n])
                continue
            for (rr, ss, sel, V, W, lam, p) in k["groups"]:
                val = AY[k["t0"] + sel] * lam
                if rr != ss:
                    val = 2.0 * val
                np.add.at(rj, p, val)
        return res


@dataclass
class SolveResult:
    status: str
    x: np.ndarray
    X: np.ndarray
    y: np.ndarray
    Y: np.ndarray
    time_total: float
    error_code: int
    iterations: int
    dual_objective: float
    primal_objective: float
    duality_gap: float
    dual_error: float
    primal_error: float
    history: np.ndarray
    timings: dict


def _blocks(f: FlatSDP, v: np.ndarray):
    for b in range(f.n_blocks):
        n = int(f.block_n[b])
        yield b, n, v[f.block_off[b]:f.block_off[b + 1]].reshape(n, n, order="F")


def _block_apply(f: FlatSDP, fn, *arrs) -> np.ndarray:
    out = np.empty(f.xy_len)
    for b in range(f.n_blocks):
        n = int(f.block_n[b])
        sl = slice(int(f.block_off[b]), int(f.block_off[b + 1]))
        out[sl] = fn(*[a[sl].reshape(n, n, order="F") for a in arrs]).reshape(-1, order="F")
    return out


def _potrs_blocks(f: FlatSDP, Lc: np.ndarray, M: np.ndarray) -> np.ndarray:
    import scipy.linalg as sla
    return _block_apply(f, lambda L, A: sla.cho_solve((L, True), A, check_finite=False), Lc, M)


def compute_step_length(f: FlatSDP, M: np.ndarray, dM: np.ndarray, gamma: float, unsafe_step: bool) -> float:
    """compute_step_length (src/solver.jl:1620-1693): alpha = min(-gamma / eigmin(L^-1 dM L^-T), 1).
    The reference finds the eigenvalue with a Float64 Lanczos (tol 1e-5, then subtracts 1e-5); here LAPACK."""
    import scipy.linalg as sla
    min_eig = np.inf
    for b, n, Mb in _blocks(f, M):
        dMb = dM[f.block_off[b]:f.block_off[b + 1]].reshape(n, n, order="F")
        if n == 1:
            e = dMb[0, 0] / Mb[0, 0]
        else:
            try:
                L = np.linalg.cholesky(Mb)
            except np.linalg.LinAlgError:
                raise SolverFailure("The cholesky decomposition could not be computed during the computation of the step length. "
                                    "Please try again with a higher precision.")
            W = sla.solve_triangular(L, dMb, lower=True, check_finite=False)
            W = sla.solve_triangular(L, W.T, lower=True, check_finite=False)
            e = np.linalg.eigvalsh((W + W.T) / 2)[0] - 1e-5
        min_eig = min(min_eig, e)
    if min_eig > -gamma and not unsafe_step:
        return 1.0
    return -gamma / min_eig


def solvesdp(sdp, ctx: Optional[SchurContext] = None, device: int = 0, maxiterations: int = 500,
             beta_infeasible: float = 0.3, beta_feasible: float = 0.1, gamma: float = 0.9,
             omega_p: float = 1e4, omega_d: float = 1e4,
             duality_gap_threshold: float = 1e-7, dual_error_threshold: float = 1e-9, primal_error_threshold: float = 1e-9,
             max_complementary_gap: float = 1e100, need_dual_feasible: bool = False, need_primal_feasible: bool = False,
             verbose: bool = False, step_length_threshold: float = 1e-7, safe_step: bool = True) -> SolveResult:
    """Primal-dual interior-point loop of the reference (src/solver.jl:100-744) with the hot path on the GPU.

    Same algorithm and keyword names as the reference; the defaults that depend on the working precision
    (omega, thresholds) are set for fp64 (the reference's 1e10 / 1e-30 / 1e-15 assume 256-bit Arb): near
    gap 1e-8 the Schur complement of e.g. delsarte(3,10) has cond(S) ~ 1e17 and whether its Cholesky
    succeeds in fp64 is decided by rounding noise, so the fp64 default gap threshold is 1e-7.
    Returns a SolveResult; `status` is one of the reference's Optimal / NearOptimal / Feasible /
    PrimalFeasible / DualFeasible / NotConverged (src/solver.jl:727-741); error_code as :64-70 of docs/src/solving.md."""
    f = sdp if isinstance(sdp, FlatSDP) else flatten(sdp)
    own_ctx = ctx is None
    if ctx is None:
        ctx = SchurContext(f, device=device)
    ctx.set_timing(True)
    hb = _HostBlocks(f)
    N, nx, nxy = f.n_free, f.x_len, f.xy_len
    sgn = 1.0 if f.maximize else -1.0
    x, y = np.zeros(nx), np.zeros(N)
    X, Y = np.zeros(nxy), np.zeros(nxy)
    K = 0
    eye = np.zeros(nxy)
    for b in range(f.n_blocks):
        n = int(f.block_n[b])
        K += n
        eye[f.block_off[b]:f.block_off[b + 1]] = np.eye(n).reshape(-1)
    X[:] = omega_p * eye                                                           # :187-201
    Y[:] = omega_d * eye
    Bm = [f.B[int(f.cluster_off[j]) * N: int(f.cluster_off[j + 1]) * N].reshape(int(f.cluster_P[j]), N, order="F") for j in range(f.n_clusters)]

    def objectives():
        d_obj = sgn * float(f.c @ x) + f.constant                                  # :793-799
        p_obj = float(f.C @ Y) + float(f.b @ y) + f.constant                       # :802-804
        return d_obj, p_obj, abs(d_obj - p_obj) / max(1.0, abs(d_obj + p_obj))     # :844-847

    def residuals(AY=None):
        Pm = hb.weighted_A(x) - X - sgn * f.C                                      # :882-893
        tr = hb.trace_A(Y) if AY is None else hb.trace_A_from_AY(Y, AY)
        d = f.c - tr                                                               # :863-879
        p = sgn * f.b.copy()                                                       # :899-916
        for j in range(f.n_clusters):
            sl = slice(int(f.cluster_off[j]), int(f.cluster_off[j + 1]))
            if N:
                d[sl] -= Bm[j] @ y
                p -= Bm[j].T @ x[sl]
        return Pm, p, d

    def maxabs(v):
        return float(np.max(np.abs(v))) if v.size else 0.0

    hist = []
    tm = dict(schur=0.0, cholS=0.0, LinvB=0.0, Q=0.0, cholQ=0.0, solve=0.0, host=0.0)
    d_obj, p_obj, gap = objectives()
    Pm, pv, dv = residuals()
    dual_error, primal_error = max(maxabs(pv), maxabs(Pm)), maxabs(dv)
    pd_feas = dual_error < dual_error_threshold and primal_error < primal_error_threshold
    error_code, it = 0, 1
    t_start = time.time()
    alpha_p = alpha_d = beta_c = 0.0
    try:
        while True:
            dual_feas, primal_feas = dual_error < dual_error_threshold, primal_error < primal_error_threshold
            if (need_dual_feasible and dual_feas) or (need_primal_feasible and primal_feas):      # :921-950
                break
            if dual_feas and primal_feas and gap < duality_gap_threshold:
                break
            if it > maxiterations:
                error_code = 2
                break
            mu = float(X @ Y) / K                                                  # :369
            mu_p = 0.0 if pd_feas else beta_infeasible * mu                        # :373
            if mu > max_complementary_gap:
                error_code = 3
                break
            R = mu_p * eye - _block_apply(f, lambda a, b_: a @ b_, X, Y)          # :961-970
            X_inv = ctx.cholesky_blocks(X)                                         # :388-399   (GPU)
            times, _, AY = compute_T_decomposition(ctx, X_inv, Y)                  # :406-408   (GPU)
            for kname, tv in zip(("schur", "cholS", "LinvB", "Q", "cholQ"), times):
                tm[kname] += tv
            Pm, pv, dv = residuals(AY)                                             # :415
            xy = float(X @ Y)
            dX = dY = None
            for corrector in (False, True):
                if corrector:
                    r = (xy + float(X @ dY) + float(dX @ Y) + float(dX @ dY)) / (mu * K)      # :429
                    beta = r * r if r < 1 else r
                    beta_c = min(max(beta_feasible, beta), 1.0) if pd_feas else max(beta_infeasible, beta)
                    mu_c = beta_c * mu
                    R = mu_c * eye - _block_apply(f, lambda a, b_, c_, d_: a @ b_ + c_ @ d_, X, Y, dX, dY)   # :972-983
                    dual_error, primal_error = max(maxabs(pv), maxabs(Pm)), maxabs(dv)                        # :441-447
                    pd_feas = dual_error < dual_error_threshold and primal_error < primal_error_threshold
                # compute_search_direction! (:1474-1616)
                Z = _potrs_blocks(f, X_inv, _block_apply(f, lambda a, b_: a @ b_, Pm, Y) - R)               # :1501-1514
                Z = _block_apply(f, lambda a: (a + a.T) / 2, Z)
                rhs_x = -dv - hb.trace_A(Z)                                                                  # :1522-1523
                dx, dy = solve_system(ctx, rhs_x, pv)                                                        # :1527-1582 (GPU)
                tm["solve"] += ctx.timings()[5]
                dX = hb.weighted_A(dx) + Pm                                                                  # :1588-1591
                dY = _potrs_blocks(f, X_inv, R - _block_apply(f, lambda a, b_: a @ b_, dX, Y))              # :1598-1612
                dY = _block_apply(f, lambda a: (a + a.T) / 2, dY)
            alpha_d = compute_step_length(f, X, dX, gamma, pd_feas and not safe_step)                        # :462-463
            alpha_p = compute_step_length(f, Y, dY, gamma, pd_feas and not safe_step)
            hist.append([it, mu, d_obj, p_obj, gap, maxabs(Pm), maxabs(pv), maxabs(dv), alpha_d, alpha_p, beta_c])
            if verbose:
                print("%5d %8.1f %11.3e %11.3e %11.3e %10.2e %10.2e %10.2e %10.2e %10.2e %10.2e %10.2e" %
                      (it, time.time() - t_start, mu, d_obj, p_obj, gap, maxabs(Pm), maxabs(pv), maxabs(dv), alpha_d, alpha_p, beta_c))
            if min(alpha_d, alpha_p) < step_length_threshold:                       # :470-475
                error_code = 4
                break
            if pd_feas and safe_step:                                              # :480-483
                alpha_p = alpha_d = min(alpha_p, alpha_d)
            x += alpha_d * dx; y += alpha_p * dy                                   # :485-495
            X += alpha_d * dX; Y += alpha_p * dY
            d_obj, p_obj, gap = objectives()                                       # :586-588
            it += 1
    except SolverFailure as e:                                                     # :594-623
        if verbose:
            print("SolverFailure:", e)
            print("We return the current solution and optimality status.")
        error_code = 1
    t_total = time.time() - t_start
    d_obj, p_obj, gap = objectives()
    if pd_feas and gap < duality_gap_threshold:                                    # :727-741
        status = "Optimal"
    elif (pd_feas and gap < 1e-8) or (dual_error < 1e-15 and primal_error < 1e-15 and gap < 1e-8):
        status = "NearOptimal"
    elif pd_feas:
        status = "Feasible"
    elif primal_error < primal_error_threshold:
        status = "PrimalFeasible"
    elif dual_error < dual_error_threshold:
        status = "DualFeasible"
    else:
        status = "NotConverged"
    if own_ctx:
        ctx.close()
    return SolveResult(status, x, X, y, Y, t_total, error_code, it - 1, d_obj, p_obj, gap, dual_error, primal_error,
                       np.array(hist).reshape(-1, 11), tm)


# ------------------------------------------------------------------------------------------------
# device-resident loop: the same algorithm with every iterate in HBM (clrs_ipm_*)
# ------------------------------------------------------------------------------------------------

def solvesdp_device(sdp, ctx: Optional[SchurContext] = None, device: int = 0, maxiterations: int = 500,
                    beta_infeasible: float = 0.3, beta_feasible: float = 0.1, gamma: float = 0.9,
                    omega_p: float = 1e4, omega_d: float = 1e4,
                    duality_gap_threshold: float = 1e-7, dual_error_threshold: float = 1e-9, primal_error_threshold: float = 1e-9,
                    max_complementary_gap: float = 1e100, need_dual_feasible: bool = False, need_primal_feasible: bool = False,
                    verbose: bool = False, step_length_threshold: float = 1e-7, safe_step: bool = True,
                    prec: Optional[int] = None, limbs: Optional[int] = None) -> SolveResult:
    """`solvesdp` (src/solver.jl:100-744) with the whole loop body on the GPU: residuals, both search directions (through
    the Schur assembly / factor / solve of the path), step lengths and the update never leave HBM; the host reads one record
    per iteration and decides termination (src/solver.jl:921-950).  Same keywords and defaults as `solvesdp` above.
    Raises ClrsError when a PSD block is too large for the LDS-resident kernels (use `solvesdp` then).

    `prec` (bits, the reference's keyword, src/solver.jl:73) or `limbs` selects the multi-word path (`mw.solvesdp_mw`): the same loop
    with every number `limbs` fp64 words (prec = 256 -> 5).  The fp64 defaults above (omega 1e4, gap 1e-7, errors 1e-9) are then
    replaced by the reference's own (1e10, 1e-15, 1e-30) unless given explicitly."""
    if prec is not None or limbs is not None:
        from .mw import solvesdp_mw
        kw = dict(maxiterations=maxiterations, beta_infeasible=beta_infeasible, beta_feasible=beta_feasible, gamma=gamma,
                  max_complementary_gap=max_complementary_gap, need_dual_feasible=need_dual_feasible,
                  need_primal_feasible=need_primal_feasible, verbose=verbose, step_length_threshold=step_length_threshold, safe_step=safe_step)
        if (omega_p, omega_d) != (1e4, 1e4):
            kw.update(omega_p=omega_p, omega_d=omega_d)
        if (duality_gap_threshold, dual_error_threshold, primal_error_threshold) != (1e-7, 1e-9, 1e-9):
            kw.update(duality_gap_threshold=duality_gap_threshold, dual_error_threshold=dual_error_threshold, primal_error_threshold=primal_error_threshold)
        return solvesdp_mw(sdp, limbs=limbs, prec=prec, device=device, **kw)
    f = sdp if isinstance(sdp, FlatSDP) else flatten(sdp)
    own_ctx = ctx is None
    if ctx is None:
        ctx = SchurContext(f, device=device)
    L = ctx.L
    keep = [_c(f.C), _c(f.c), _c(f.b if f.n_free else np.zeros(1))]
    data = _lib.IpmData(_dp(keep[0]), _dp(keep[1]), _dp(keep[2]), int(f.maximize), 0, float(f.constant))
    _lib.check(L.clrs_ipm_create(ctx.h, C.byref(data)))
    prm = _lib.IpmParams(beta_infeasible, beta_feasible, gamma, dual_error_threshold, primal_error_threshold, max_complementary_gap,
                         step_length_threshold, int(safe_step), 0)      # (corrector_only: multi-word loop only)
    _lib.check(L.clrs_ipm_set_params(ctx.h, C.byref(prm)))
    _lib.check(L.clrs_ipm_init(ctx.h, float(omega_p), float(omega_d)))
    rec = _lib.IpmRecord()
    hist = []
    t_start = time.time()
    error_code, it = 0, 1
    dual_error = primal_error = np.inf      # computed by the first iteration; no termination test can pass before
    gap = 0.0                               # x = 0, y = 0: both objectives equal the constant (src/solver.jl:319-321)
    d_obj = p_obj = f.constant
    pd_feas = False
    while True:
        dual_feas, primal_feas = dual_error < dual_error_threshold, primal_error < primal_error_threshold
        if (need_dual_feasible and dual_feas) or (need_primal_feasible and primal_feas):          # :921-950
            break
        if dual_feas and primal_feas and gap < duality_gap_threshold:
            break
        if it > maxiterations:
            error_code = 2
            break
        _lib.check(L.clrs_ipm_iterate(ctx.h, C.byref(rec)))
        hist.append([it, rec.mu, d_obj, p_obj, gap, rec.max_P, rec.max_p, rec.max_d, rec.alpha_d, rec.alpha_p, rec.beta_c])
        if verbose:
            print("%5d %8.1f %11.3e %11.3e %11.3e %10.2e %10.2e %10.2e %10.2e %10.2e %10.2e %10.2e" %
                  (it, time.time() - t_start, rec.mu, d_obj, p_obj, gap, rec.max_P, rec.max_p, rec.max_d, rec.alpha_d, rec.alpha_p, rec.beta_c))
        # the errors of the record belong to the iterate BEFORE this update (like the table row of the reference); they
        # are what the next termination test sees together with the new objectives
        dual_error, primal_error, pd_feas = rec.dual_error, rec.primal_error, bool(rec.pd_feas)
        if rec.error_code:
            error_code = rec.error_code
            if verbose and rec.error_code == 1:
                print("SolverFailure: factor status %d, Cholesky status %d" % (rec.factor_status, rec.cholesky_status))
            break
        d_obj, p_obj, gap = rec.d_obj, rec.p_obj, rec.gap
        it += 1
    t_total = time.time() - t_start
    x, y = np.zeros(f.x_len), np.zeros(max(f.n_free, 1))
    X, Y = np.zeros(f.xy_len), np.zeros(f.xy_len)
    _lib.check(L.clrs_ipm_get(ctx.h, _dp(x), _dp(y), _dp(X), _dp(Y)))
    if pd_feas and gap < duality_gap_threshold:                                    # :727-741
        status = "Optimal"
    elif (pd_feas and gap < 1e-8) or (dual_error < 1e-15 and primal_error < 1e-15 and gap < 1e-8):
        status = "NearOptimal"
    elif pd_feas:
        status = "Feasible"
    elif primal_error < primal_error_threshold:
        status = "PrimalFeasible"
    elif dual_error < dual_error_threshold:
        status = "DualFeasible"
    else:
        status = "NotConverged"
    if own_ctx:
        ctx.close()
    return SolveResult(status, x, X, y[:f.n_free], Y, t_total, error_code, it - 1, d_obj, p_obj, gap, dual_error, primal_error,
                       np.array(hist).reshape(-1, 11), dict(loop="device"))
