"""ctypes wrapper around the CPU ORACLE (oracle/clrs_oracle.c) -- test infrastructure only.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this module; the
product package (clusteredlowranksolver.jl_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HIST_COLS = 11
HIST_NAMES = ["iter", "mu", "d_obj", "p_obj", "gap", "P_err", "p_err", "d_err", "alpha_d", "alpha_p", "beta"]

_p_d = C.POINTER(C.c_double)
_p_i = C.POINTER(C.c_int)
_p_l = C.POINTER(C.c_longlong)


class _OracleSDP(C.Structure):
    _fields_ = [
        ("n_clusters", C.c_int), ("n_free", C.c_int), ("cluster_P", _p_i),
        ("B", _p_d), ("B_lo", _p_d), ("c", _p_d), ("c_lo", _p_d), ("b", _p_d), ("b_lo", _p_d),
        ("C", _p_d), ("C_lo", _p_d), ("maximize", C.c_int), ("constant", C.c_double), ("constant_lo", C.c_double),
        ("n_blocks", C.c_int), ("block_cluster", _p_i), ("block_m", _p_i), ("block_delta", _p_i), ("block_kind", _p_i),
        ("term_ptr", _p_l), ("term_p", _p_i), ("term_r", _p_i), ("term_s", _p_i), ("term_rank", _p_i),
        ("term_lambda", _p_d), ("term_lambda_lo", _p_d), ("term_vec_ptr", _p_l),
        ("term_vs", _p_d), ("term_vs_lo", _p_d), ("term_ws", _p_d), ("term_ws_lo", _p_d),
        ("dense_ptr", _p_l), ("dense_p", _p_i), ("dense_A_ptr", _p_l), ("dense_A", _p_d), ("dense_A_lo", _p_d),
        ("n_tail", C.c_int), ("B_tail", _p_d), ("c_tail", _p_d), ("b_tail", _p_d), ("C_tail", _p_d), ("term_lambda_tail", _p_d),
        ("term_vs_tail", _p_d), ("term_ws_tail", _p_d), ("dense_A_tail", _p_d),
    ]


class OracleParams(C.Structure):
    _fields_ = [
        ("maxiterations", C.c_int),
        ("beta_infeasible", C.c_double), ("beta_feasible", C.c_double), ("gamma", C.c_double),
        ("omega_p", C.c_double), ("omega_d", C.c_double),
        ("duality_gap_threshold", C.c_double), ("dual_error_threshold", C.c_double),
        ("primal_error_threshold", C.c_double), ("max_complementary_gap", C.c_double),
        ("step_length_threshold", C.c_double),
        ("need_dual_feasible", C.c_int), ("need_primal_feasible", C.c_int), ("safe_step", C.c_int), ("verbose", C.c_int),
        ("correctoronly", C.c_int),
    ]


def build(force: bool = False) -> None:
    """Compile the oracle shared libraries with the committed Makefile."""
    libs = ("libclrs_oracle_f64.so", "libclrs_oracle_f128.so", "libclrs_oracle_mp.so", "libclrs_oracle_mp10.so")
    need = force or not all(os.path.exists(os.path.join(_HERE, f)) for f in libs)
    if not need:
        src = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("clrs_oracle.c", "clrs_oracle_mp.cpp", "mpx.hpp"))
        need = any(os.path.getmtime(os.path.join(_HERE, f)) < src for f in libs)
    if need:
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)


_libs = {}


MP_LIMB_BITS = 320   # mantissa bits of libclrs_oracle_mp.so (mpx<5>)
MP10_LIMB_BITS = 640 # mantissa bits of libclrs_oracle_mp10.so (mpx<10>), the checker of the 6- and 8-limb GPU paths


def _lib(quad):
    key = quad if quad in ("mp", "mp10") else ("f128" if quad else "f64")
    if key not in _libs:
        path = os.path.join(_HERE, f"libclrs_oracle_{key}.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.POINTER(_OracleSDP)]
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_dims.argtypes = [C.c_void_p, _p_l, _p_l, _p_l, _p_l]
        L.oracle_cholesky_blocks.restype = C.c_int
        L.oracle_cholesky_blocks.argtypes = [C.c_void_p, _p_d, _p_d, _p_d, _p_d]
        L.oracle_schur_assemble.argtypes = [C.c_void_p] + [_p_d] * 8
        L.oracle_schur_dense_check.argtypes = [C.c_void_p] + [_p_d] * 6
        L.oracle_schur_factor.restype = C.c_int
        L.oracle_schur_factor.argtypes = [C.c_void_p]
        L.oracle_get_factor.argtypes = [C.c_void_p] + [_p_d] * 6
        L.oracle_schur_solve.argtypes = [C.c_void_p] + [_p_d] * 8
        L.oracle_default_params.argtypes = [C.POINTER(OracleParams)]
        L.oracle_solvesdp.restype = C.c_int
        L.oracle_solvesdp.argtypes = [C.c_void_p, C.POINTER(OracleParams), _p_i, _p_d, _p_d, C.c_int, _p_d, _p_d, _p_d, _p_d]
        L.oracle_unique_counts.restype = C.c_int
        L.oracle_unique_counts.argtypes = [C.c_void_p, C.c_int, _p_i, _p_i]
        L.oracle_real_bits.restype = C.c_int
        L.oracle_set_precision_bits.argtypes = [C.c_int]
        L.oracle_set_matmul_precision_bits.argtypes = [C.c_int]
        L.oracle_real_op.argtypes = [C.c_int, C.c_int, C.c_longlong, _p_d, _p_d, _p_d]
        L.oracle_cholesky_blocks_mw.restype = C.c_int
        L.oracle_cholesky_blocks_mw.argtypes = [C.c_void_p, C.c_int, _p_d, _p_d]
        L.oracle_schur_assemble_mw.argtypes = [C.c_void_p, C.c_int, _p_d, _p_d, _p_d, _p_d]
        L.oracle_set_S_mw.argtypes = [C.c_void_p, C.c_int, _p_d]
        L.oracle_get_factor_mw.argtypes = [C.c_void_p, C.c_int, _p_d, _p_d, _p_d]
        L.oracle_schur_solve_mw.argtypes = [C.c_void_p, C.c_int, _p_d, _p_d, _p_d, _p_d]
        L.oracle_set_snapshots.argtypes = [C.c_void_p, C.c_int, _p_i, C.c_int, _p_d, _p_d, _p_d, _p_d]
        L.oracle_set_start.argtypes = [C.c_void_p, C.c_int, _p_d, _p_d, _p_d, _p_d]
        L.oracle_kkt_backward_error_mw.argtypes = [C.c_void_p, C.c_int] + [_p_d] * 6
        L.oracle_last_objectives_mw.argtypes = [C.c_void_p, C.c_int, _p_d]
        L.oracle_snapshot_count.restype = C.c_int
        L.oracle_snapshot_count.argtypes = [C.c_void_p]
        L.oracle_num_threads.restype = C.c_int
        L.oracle_set_num_threads.argtypes = [C.c_int]
        _libs[key] = L
    return _libs[key]


def _dp(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_p_d)


def _ip(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_p_i)


def _lp(a):
    assert a.dtype == np.int64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_p_l)


def _c(a, dt=np.float64):
    return np.ascontiguousarray(a, dtype=dt)


def real_op(op: str, a: np.ndarray, b: np.ndarray, mp_bits: int = MP_LIMB_BITS) -> np.ndarray:
    """One elementwise operation ('add', 'sub', 'mul', 'div', 'sqrt') of the multi-precision oracle's arithmetic on planar k-limb
    arrays of shape (k, n), truncated to `mp_bits` bits -- for unit tests of oracle/mpx.hpp."""
    L = _lib("mp" if mp_bits <= MP_LIMB_BITS else "mp10")
    L.oracle_set_precision_bits(int(mp_bits))
    a, b = _c(a), _c(b)
    out = np.zeros_like(a)
    L.oracle_real_op({"add": 0, "sub": 1, "mul": 2, "div": 3, "sqrt": 4}[op], a.shape[0], a.shape[1], _dp(a), _dp(b), _dp(out))
    return out


class Oracle:
    """CPU oracle context for a FlatSDP.  `quad=True` computes in __float128 using the (hi, lo) inputs;
    `use_lo=False` forces the fp64-rounded problem data (what the HIP path sees) even in quad.
    `mp_bits=p` computes in the multi-limb type of mpx.hpp truncated to p bits per operation (p <= 320, or <= 640 with the mpx<10> build; the stand-in for
    the reference's Arb midpoints at `prec=p`).  The precision is a property of the loaded library, not of the
    context: it is re-applied by every method of a multi-precision context."""

    def __init__(self, flat, quad: bool = False, use_lo: bool = True, mp_bits: Optional[int] = None, matmul_bits: Optional[int] = None):
        self.flat = flat
        self.mp_bits = mp_bits
        self.matmul_bits = matmul_bits      # the reference's matmul_prec (src/solver.jl:125): bits of the pairing products; None = the working precision
        if mp_bits is not None:
            if not 1 <= mp_bits <= MP10_LIMB_BITS:
                raise ValueError("mp_bits must be within 1..%d" % MP10_LIMB_BITS)
            quad = "mp" if mp_bits <= MP_LIMB_BITS else "mp10"
        self.quad = quad
        self.L = _lib(quad)
        self._prec()
        f = flat
        keep = self._keep = {}

        def hold(name, arr, dt=np.float64):
            keep[name] = _c(arr, dt)
            return keep[name]

        def lo(name, arr):
            if not (quad and use_lo) or arr is None:
                return None
            return _dp(hold(name, arr))

        d = _OracleSDP()
        d.n_clusters, d.n_free = f.n_clusters, f.n_free
        d.cluster_P = _ip(hold("cluster_P", f.cluster_P, np.int32))
        d.B, d.B_lo = _dp(hold("B", f.B)), lo("B_lo", f.B_lo)
        d.c, d.c_lo = _dp(hold("c", f.c)), lo("c_lo", f.c_lo)
        d.b, d.b_lo = _dp(hold("b", f.b)), lo("b_lo", f.b_lo)
        d.C, d.C_lo = _dp(hold("C", f.C)), lo("C_lo", f.C_lo)
        d.maximize, d.constant, d.constant_lo = f.maximize, f.constant, 0.0
        d.n_blocks = f.n_blocks
        d.block_cluster = _ip(hold("bc", f.block_cluster, np.int32))
        d.block_m = _ip(hold("bm", f.block_m, np.int32))
        d.block_delta = _ip(hold("bd", f.block_delta, np.int32))
        d.block_kind = _ip(hold("bk", f.block_kind, np.int32))
        d.term_ptr = _lp(hold("tptr", f.term_ptr, np.int64))
        d.term_p = _ip(hold("tp", f.term_p, np.int32)); d.term_r = _ip(hold("tr", f.term_r, np.int32))
        d.term_s = _ip(hold("ts", f.term_s, np.int32)); d.term_rank = _ip(hold("tk", f.term_rank, np.int32))
        d.term_lambda, d.term_lambda_lo = _dp(hold("tl", f.term_lambda)), lo("tl_lo", f.term_lambda_lo)
        d.term_vec_ptr = _lp(hold("tvp", f.term_vec_ptr, np.int64))
        d.term_vs, d.term_vs_lo = _dp(hold("tvs", f.term_vs)), lo("tvs_lo", f.term_vs_lo)
        d.term_ws, d.term_ws_lo = _dp(hold("tws", f.term_ws)), lo("tws_lo", f.term_ws_lo)
        d.dense_ptr = _lp(hold("dptr", f.dense_ptr, np.int64))
        d.dense_p = _ip(hold("dp", f.dense_p, np.int32))
        d.dense_A_ptr = _lp(hold("dAp", f.dense_A_ptr, np.int64))
        d.dense_A, d.dense_A_lo = _dp(hold("dA", f.dense_A)), lo("dA_lo", f.dense_A_lo)
        # limb planes 3.. of the data (FlatSDP.tails: generators run under clrs_amd.sdp.data_planes): the sampled problem at the working precision
        tails = getattr(f, "tails", None) or {}
        d.n_tail = 0
        if tails and quad and use_lo:
            d.n_tail = int(next(iter(tails.values())).shape[0])
            for name in ("B", "c", "b", "C", "term_lambda", "term_vs", "term_ws", "dense_A"):
                t = tails.get(name)
                setattr(d, name + "_tail", _dp(hold(name + "_tail", t if t is not None and t.size else np.zeros((d.n_tail, 1)))))
        self.ctx = self.L.oracle_create(C.byref(d))
        if not self.ctx:
            raise RuntimeError("oracle_create failed")

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                self.L.oracle_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass

    def _prec(self):
        if self.mp_bits is not None:
            self.L.oracle_set_precision_bits(int(self.mp_bits))
            self.L.oracle_set_matmul_precision_bits(int(self.matmul_bits or 0))

    # -- hot path on k-limb planar arrays (shape (k, len); value = sum over axis 0) -----------------
    def cholesky_blocks_mw(self, X):
        self._prec()
        X = _c(np.atleast_2d(X)); k = X.shape[0]
        L = np.zeros_like(X)
        st = self.L.oracle_cholesky_blocks_mw(self.ctx, k, _dp(X), _dp(L))
        return int(st), L

    def schur_assemble_mw(self, Xchol, Y):
        self._prec()
        f = self.flat
        Xchol, Y = _c(np.atleast_2d(Xchol)), _c(np.atleast_2d(Y)); k = Xchol.shape[0]
        assert Y.shape == Xchol.shape == (k, f.xy_len)
        S, AY = np.zeros((k, f.S_len)), np.zeros((k, max(f.n_terms, 1)))
        self.L.oracle_schur_assemble_mw(self.ctx, k, _dp(Xchol), _dp(Y), _dp(S), _dp(AY) if f.n_terms else None)
        return S, AY[:, :f.n_terms]

    def set_S_mw(self, S):
        self._prec()
        S = _c(np.atleast_2d(S))
        self.L.oracle_set_S_mw(self.ctx, S.shape[0], _dp(S))

    def get_factor_mw(self, k):
        self._prec()
        f = self.flat
        L = np.zeros((k, f.S_len)); LinvB = np.zeros((k, max(f.x_len * f.n_free, 1))); Q = np.zeros((k, max(f.n_free * f.n_free, 1)))
        self.L.oracle_get_factor_mw(self.ctx, k, _dp(L), _dp(LinvB) if f.n_free else None, _dp(Q) if f.n_free else None)
        return L, LinvB[:, :f.x_len * f.n_free], Q[:, :f.n_free * f.n_free]

    def schur_solve_mw(self, rhs_x, rhs_y):
        self._prec()
        f = self.flat
        rx = _c(np.atleast_2d(rhs_x)); k = rx.shape[0]
        ry = _c(np.atleast_2d(rhs_y)) if f.n_free else np.zeros((k, 1))
        dx, dy = np.zeros((k, f.x_len)), np.zeros((k, max(f.n_free, 1)))
        self.L.oracle_schur_solve_mw(self.ctx, k, _dp(rx), _dp(ry), _dp(dx), _dp(dy))
        return dx, dy[:, :f.n_free]

    def kkt_backward_error_mw(self, S, dx, dy, rhs_x, rhs_y):
        """Normwise backward errors (x rows, y rows) of (dx, dy) as a solution of [S -B; B^T 0](dx; dy) = (rhs_x; rhs_y)
        (src/solver.jl:1527), evaluated at the oracle's working precision; all arguments planar limbs, S the unfactored matrix."""
        self._prec()
        f = self.flat
        k = max(np.atleast_2d(a).shape[0] for a in (S, dx, rhs_x))

        def pad(a, n):
            a = _c(np.atleast_2d(a)) if n else np.zeros((k, 1))
            return _c(np.vstack([a, np.zeros((k - a.shape[0], a.shape[1]))])) if a.shape[0] < k else a
        S, dx, rx = pad(S, f.S_len), pad(dx, f.x_len), pad(rhs_x, f.x_len)
        dy, ry = pad(dy, f.n_free), pad(rhs_y, f.n_free)
        out = np.zeros(2)
        self.L.oracle_kkt_backward_error_mw(self.ctx, k, _dp(S), _dp(dx), _dp(dy), _dp(rx), _dp(ry), _dp(out))
        return float(out[0]), float(out[1])

    # -- hot path -------------------------------------------------------------------------------
    def cholesky_blocks(self, X, X_lo=None):
        f = self.flat
        L, L_lo = np.zeros(f.xy_len), np.zeros(f.xy_len)
        st = self.L.oracle_cholesky_blocks(self.ctx, _dp(_c(X)), _dp(_c(X_lo)) if X_lo is not None else None, _dp(L), _dp(L_lo))
        return st, L, L_lo

    def schur_assemble(self, Xchol, Y, Xchol_lo=None, Y_lo=None, want_lo=False):
        f = self.flat
        S, AY = np.zeros(f.S_len), np.zeros(f.n_terms)
        S_lo, AY_lo = (np.zeros(f.S_len), np.zeros(f.n_terms)) if want_lo else (None, None)
        self.L.oracle_schur_assemble(self.ctx, _dp(_c(Xchol)), _dp(_c(Xchol_lo)) if Xchol_lo is not None else None,
                                     _dp(_c(Y)), _dp(_c(Y_lo)) if Y_lo is not None else None,
                                     _dp(S), _dp(S_lo), _dp(AY), _dp(AY_lo))
        return (S, AY, S_lo, AY_lo) if want_lo else (S, AY)

    def schur_dense_check(self, Xchol, Y, Xchol_lo=None, Y_lo=None):
        S = np.zeros(self.flat.S_len)
        self.L.oracle_schur_dense_check(self.ctx, _dp(_c(Xchol)), _dp(_c(Xchol_lo)) if Xchol_lo is not None else None,
                                        _dp(_c(Y)), _dp(_c(Y_lo)) if Y_lo is not None else None, _dp(S), None)
        return S

    def schur_factor(self) -> int:
        return int(self.L.oracle_schur_factor(self.ctx))

    def get_factor(self):
        f = self.flat
        L = np.zeros(f.S_len); LinvB = np.zeros(f.x_len * f.n_free); Q = np.zeros(f.n_free * f.n_free)
        self.L.oracle_get_factor(self.ctx, _dp(L), None, _dp(LinvB), None, _dp(Q), None)
        return L, LinvB, Q

    def schur_solve(self, rhs_x, rhs_y, rhs_x_lo=None, rhs_y_lo=None):
        f = self.flat
        dx, dy = np.zeros(f.x_len), np.zeros(max(f.n_free, 1))
        ry = _c(rhs_y) if f.n_free else np.zeros(1)
        self.L.oracle_schur_solve(self.ctx, _dp(_c(rhs_x)), _dp(_c(rhs_x_lo)) if rhs_x_lo is not None else None,
                                  _dp(ry), _dp(_c(rhs_y_lo)) if (rhs_y_lo is not None and f.n_free) else None,
                                  _dp(dx), None, _dp(dy), None)
        return dx, dy[:f.n_free]

    def unique_counts(self, b: int):
        m = int(self.flat.block_m[b])
        UR, UL = np.zeros(m, np.int32), np.zeros(m, np.int32)
        k = self.L.oracle_unique_counts(self.ctx, b, _ip(UR), _ip(UL))
        return (UR[:k], UL[:k])

    # -- solver loop ---------------------------------------------------------------------------
    def default_params(self) -> OracleParams:
        p = OracleParams()
        self.L.oracle_default_params(C.byref(p))
        return p

    def solvesdp(self, params: Optional[OracleParams] = None, hist_rows: int = 600, snapshots=None, snapshot_limbs: int = 1, start=None, **kw):
        """`start`: (x, y, X, Y) as k-limb planar arrays -- the dualsol / primalsol warm start of src/solver.jl:202-239.
        `snapshots`: ascending 1-based iteration numbers; the result then carries `snap` = dict(iters, X, Y, rhs_x, rhs_y)
        with k-limb planar arrays of shape (n, k, len): the iterate at the top of those iterations and the predictor's
        right-hand sides (trajectory fixtures)."""
        self._prec()
        p = params or self.default_params()
        for k, v in kw.items():
            setattr(p, k, v)
        f = self.flat
        snap = None
        if snapshots is not None and len(snapshots):
            its = _c(sorted(int(i) for i in snapshots), np.int32)
            ns, kl = len(its), int(snapshot_limbs)
            snap = dict(iters=its, X=np.zeros((ns, kl, f.xy_len)), Y=np.zeros((ns, kl, f.xy_len)),
                        rhs_x=np.zeros((ns, kl, f.x_len)), rhs_y=np.zeros((ns, kl, max(f.n_free, 1))))
            self.L.oracle_set_snapshots(self.ctx, ns, _ip(its), kl, _dp(snap["X"]), _dp(snap["Y"]), _dp(snap["rhs_x"]), _dp(snap["rhs_y"]))
        if start is not None:
            st_ = [np.ascontiguousarray(np.atleast_2d(a), dtype=np.float64) for a in start]
            if f.n_free == 0:
                st_[1] = np.zeros((st_[0].shape[0], 1))
            assert len({a.shape[0] for a in st_}) == 1 and st_[0].shape[1] == f.x_len and st_[2].shape[1] == f.xy_len and st_[3].shape[1] == f.xy_len
            self.L.oracle_set_start(self.ctx, st_[0].shape[0], _dp(st_[0]), _dp(st_[1]), _dp(st_[2]), _dp(st_[3]))
        iters = C.c_int(0)
        out = np.zeros(6)
        hist = np.zeros((hist_rows, HIST_COLS))
        x, y = np.zeros(f.x_len), np.zeros(max(f.n_free, 1))
        X, Y = np.zeros(f.xy_len), np.zeros(f.xy_len)
        code = self.L.oracle_solvesdp(self.ctx, C.byref(p), C.byref(iters), _dp(out), _dp(hist), hist_rows, _dp(x), _dp(y), _dp(X), _dp(Y))
        n = min(iters.value, hist_rows)
        if start is not None:
            self.L.oracle_set_start(self.ctx, 0, None, None, None, None)
        if snap is not None:
            cnt = int(self.L.oracle_snapshot_count(self.ctx))
            self.L.oracle_set_snapshots(self.ctx, 0, None, 1, None, None, None, None)
            snap = {k_: (v[:cnt] if k_ != "rhs_y" else v[:cnt, :, :f.n_free]) for k_, v in snap.items()}
        obj_limbs = np.zeros((6, 3))
        self.L.oracle_last_objectives_mw(self.ctx, 6, _dp(obj_limbs))
        return dict(snap=snap, objectives_limbs=obj_limbs.T.copy(), error_code=int(code), iterations=int(iters.value), d_obj=out[0], p_obj=out[1], gap=out[2],
                    dual_error=out[3], primal_error=out[4], pd_feas=bool(out[5]), hist=hist[:n],
                    x=x, y=y[:f.n_free], X=X, Y=Y)

    @property
    def real_bits(self):
        return int(self.L.oracle_real_bits())

    def set_num_threads(self, n: int):
        self.L.oracle_set_num_threads(int(n))

    @property
    def num_threads(self):
        return int(self.L.oracle_num_threads())
