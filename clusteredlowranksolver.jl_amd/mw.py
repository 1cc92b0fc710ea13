"""Host-side mirror of the hot path at the reference's working precision (multi-word fp64, clrs_mw_* of the C ABI).

The reference runs this path on Arb midpoints at `prec` bits (256 by default, src/solver.jl:73,103); here a number is an
unevaluated sum of `limbs` doubles and an array of them is PLANAR: shape (limbs, len), value = sum over axis 0, limb 0 the
value rounded to fp64.  `MwSchurContext` offers the same calls as `solver.SchurContext` on such arrays:

    ctx = MwSchurContext(sdp, limbs=5)                 # precompute_matrices_bilinear_pairings   src/solver.jl:985-1059
    X_inv = ctx.cholesky_blocks(X)                     # approx_cholesky!(X_inv_blk, X_blk)      :388-399
    S, A_Y = ctx.compute_S_integrated(X_inv, Y)        # compute_S_integrated!                   :1062-1226
    ctx.factor()                                       # steps 3-4 of compute_T_decomposition!   :1244-1279
    dx, dy = ctx.solve(rhs_x, rhs_y)                   # solve stage of compute_search_direction! :1527-1582

All compute is in libclrs_hip.so; there is no CPU fallback.  `to_limbs` / `from_limbs` convert between mpmath numbers (the
stand-in for Julia's BigFloat / Arb on the caller's side) and planar limbs.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from .sdp import FlatSDP, flatten
from .solver import SolverFailure

LIMB_BITS = {2: 104, 3: 157, 4: 209, 5: 262}     # guaranteed bits of an operation's result, about 53 K - K


def limbs_for_precision(prec: int) -> int:
    """Smallest supported limb count whose operations carry at least `prec` bits (the reference's `prec` keyword)."""
    for k in sorted(LIMB_BITS):
        if LIMB_BITS[k] >= prec:
            return k
    raise ValueError(f"prec = {prec} bits needs more than 5 limbs")


def to_limbs(values, limbs: int) -> np.ndarray:
    """mpmath numbers / floats (any array shape) -> planar limbs of shape (limbs, size): successive roundings to nearest."""
    import mpmath as mp
    a = np.asarray(values, dtype=object).reshape(-1)
    out = np.zeros((limbs, a.size))
    with mp.workprec(64 * limbs + 128):
        for i, v in enumerate(a):
            r = mp.mpf(v)
            for l in range(limbs):
                h = float(r)
                out[l, i] = h
                if h == 0.0:
                    break
                r = r - mp.mpf(h)
    return out


def from_limbs(a: np.ndarray):
    """Planar limbs (limbs, len) -> object array of mpmath numbers (exact sums)."""
    import mpmath as mp
    a = np.atleast_2d(a)
    out = np.empty(a.shape[1], dtype=object)
    with mp.workprec(64 * a.shape[0] + 1200):
        for i in range(a.shape[1]):
            out[i] = mp.fsum(mp.mpf(float(a[l, i])) for l in range(a.shape[0]))
    return out


def limbs_of_double(a: np.ndarray, limbs: int) -> np.ndarray:
    """fp64 array -> planar limbs with zero tails."""
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
    out = np.zeros((limbs, a.size))
    out[0] = a
    return out


def _dp(a):
    return a.ctypes.data_as(_lib.p_d)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class MwSchurContext:
    """Device context of the hot path for one SDP at `limbs` words per number (see module docstring)."""

    def __init__(self, sdp, limbs: int = 4, device: int = 0, timing: bool = False):
        self.flat: FlatSDP = sdp if isinstance(sdp, FlatSDP) else flatten(sdp)
        f = self.flat
        self.limbs = int(limbs)
        self.L = _lib.load()
        k = self._keep = {}

        def hold(name, arr, dt):
            k[name] = np.ascontiguousarray(arr, dtype=dt)
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
        _lib.check(self.L.clrs_mw_create(C.byref(d), int(device), self.limbs, C.byref(h)))
        self.h = h
        self.device = device
        if timing:
            self.set_timing(True)

    def close(self):
        if getattr(self, "h", None):
            self.L.clrs_mw_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _planar(self, a, length):
        a = _c(a)
        if a.shape != (self.limbs, length):
            raise ValueError(f"expected planar limbs of shape ({self.limbs}, {length}), got {a.shape}")
        return a

    # -- introspection ---------------------------------------------------------------------------
    def unique_count(self, block: int) -> int:
        v = C.c_int32()
        _lib.check(self.L.clrs_mw_get_unique_count(self.h, int(block), C.byref(v)))
        return v.value

    def set_timing(self, on: bool):
        _lib.check(self.L.clrs_mw_set_timing(self.h, int(on)))

    def timings(self) -> np.ndarray:
        t = np.zeros(6)
        _lib.check(self.L.clrs_mw_get_timings(self.h, _dp(t)))
        return t

    def counters(self):
        v = [C.c_double() for _ in range(3)]
        _lib.check(self.L.clrs_mw_get_counters(self.h, *[C.byref(x) for x in v]))
        return dict(assemble_muladds=v[0].value, factor_muladds=v[1].value, solve_muladds=v[2].value)

    # -- the path --------------------------------------------------------------------------------
    def cholesky_blocks(self, X: np.ndarray) -> np.ndarray:
        f = self.flat
        X = self._planar(X, f.xy_len)
        out = np.empty_like(X)
        st = _lib.check(self.L.clrs_mw_cholesky_blocks(self.h, _dp(X), _dp(out)))
        if st > 0:
            b = st - 1
            j = int(f.block_cluster[b])
            l = b - int(np.searchsorted(f.block_cluster, j))
            raise SolverFailure(f"The cholesky decomposition of X was not computed correctly in block ({j + 1},{l + 1}). "
                                f"Try again with higher precision")
        return out

    def compute_S_integrated(self, X_inv: np.ndarray, Y: np.ndarray, want_S: bool = True, want_AY: bool = True):
        f = self.flat
        Xc, Y = self._planar(X_inv, f.xy_len), self._planar(Y, f.xy_len)
        S = np.empty((self.limbs, f.S_len)) if want_S else None
        AY = np.empty((self.limbs, f.n_terms)) if (want_AY and f.n_terms) else None
        _lib.check(self.L.clrs_mw_schur_assemble(self.h, _dp(Xc), _dp(Y), _dp(S) if S is not None else None, _dp(AY) if AY is not None else None))
        return S, AY

    def factor(self) -> int:
        return _lib.check(self.L.clrs_mw_schur_factor(self.h))

    def get_factor(self):
        f = self.flat
        K = self.limbs
        Lf = np.empty((K, f.S_len)); LinvB = np.empty((K, f.x_len * f.n_free)); LQ = np.empty((K, f.n_free * f.n_free))
        _lib.check(self.L.clrs_mw_get_factor(self.h, _dp(Lf), _dp(LinvB) if f.n_free else None, _dp(LQ) if f.n_free else None))
        return Lf, LinvB, LQ

    def solve(self, rhs_x: np.ndarray, rhs_y: Optional[np.ndarray]):
        f = self.flat
        K = self.limbs
        rx = self._planar(rhs_x, f.x_len)
        ry = self._planar(rhs_y, f.n_free) if f.n_free else np.zeros((K, 1))
        dx, dy = np.empty((K, f.x_len)), np.empty((K, max(f.n_free, 1)))
        _lib.check(self.L.clrs_mw_schur_solve(self.h, _dp(rx), _dp(ry), _dp(dx), _dp(dy)))
        return dx, dy[:, :f.n_free]

    # -- device pointers ---------------------------------------------------------------------------
    def cholesky_blocks_dev(self, d_X: int, d_Xchol: int):
        _lib.check(self.L.clrs_mw_cholesky_blocks_dev(self.h, C.c_void_p(d_X), C.c_void_p(d_Xchol)))

    def assemble_dev(self, d_Xchol: int, d_Y: int):
        _lib.check(self.L.clrs_mw_schur_assemble_dev(self.h, C.c_void_p(d_Xchol), C.c_void_p(d_Y)))

    def factor_dev(self):
        _lib.check(self.L.clrs_mw_schur_factor_dev(self.h))

    def solve_dev(self, d_rhs_x: int, d_rhs_y: int, d_dx: int, d_dy: int):
        _lib.check(self.L.clrs_mw_schur_solve_dev(self.h, C.c_void_p(d_rhs_x), C.c_void_p(d_rhs_y) if d_rhs_y else None,
                                                  C.c_void_p(d_dx), C.c_void_p(d_dy) if d_dy else None))

    def sync_status(self) -> int:
        return _lib.check(self.L.clrs_mw_sync_status(self.h))

    def sync_status_cholesky(self) -> int:
        return _lib.check(self.L.clrs_mw_sync_status_cholesky(self.h))

    def stream(self) -> int:
        return int(self.L.clrs_mw_stream(self.h) or 0)

    def set_stream(self, hip_stream: int):
        _lib.check(self.L.clrs_mw_set_stream(self.h, C.c_void_p(hip_stream)))
