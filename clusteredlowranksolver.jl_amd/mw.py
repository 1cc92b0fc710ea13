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

LIMB_BITS = {2: 104, 3: 157, 4: 209, 5: 262, 6: 315, 8: 420, 10: 525}     # guaranteed bits of an operation's result, about 53 K - K


def limbs_for_precision(prec: int) -> int:
    """Smallest supported limb count whose operations carry at least `prec` bits (the reference's `prec` keyword)."""
    for k in sorted(LIMB_BITS):
        if LIMB_BITS[k] >= prec:
            return k
    raise ValueError(f"prec = {prec} bits needs more than 10 limbs")


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

    def __init__(self, sdp, limbs: int = 4, device: int = 0, timing: bool = False, data_limbs: int = 2, exact_products: Optional[bool] = None,
                 refine: Optional[int] = None, pipeline=None, refine_predictor: Optional[bool] = None, factor_limbs: Optional[int] = None,
                 matmul_limbs: Optional[int] = None):
        """`matmul_limbs`: the reference's `matmul_prec` (src/solver.jl:125) as a limb count (`limbs_for_precision(matmul_prec)`): the products that form the
        pairing matrices in fewer limbs than the rest; None / 0 = `limbs`.
        `factor_limbs`: mixed-precision iterative refinement (csrc/clrs_mw_kernels.hip.h::mw_kf_of) -- limbs of the factor stage and of the solve
        stage's inverse-factor products, the residuals of the refinement step and the solution keep all `limbs`: None / 0 = automatic (`limbs - 1` for
        5 and 6 limbs inside `solvesdp_mw` while the measured first-pass accuracy allows, all limbs in `factor()` / `solve()`), `limbs` = never reduce,
        `limbs - 1` = reduced in `factor()` / `solve()` too.
        `exact_products`: the pairing matrices through exact slice products on the matrix cores (k_mws_pair, csrc/clrs_mw_exact.hip.h):
        None = automatic (contexts with >= 256 eligible PSD blocks), True = always, False = never.
        `refine`: iterative refinement of the solve stage (k_mw_refine): None = the library default (one step), 0 = none (products with the inverse
        factors only), 1 = the default, 2 = one step with the correction in fewer limbs (cheaper; as good while twice the lost bits fit in them).
        `refine_predictor`: in `solvesdp_mw`, whether the predictor's solve takes the refinement step too (None / False: the corrector's only).
        `pipeline`: Cholesky + inverse factor of matrices of at most 32 rows as a pipeline of workgroups (csrc/clrs_mw_pipe.hip.h): None = the library
        default (the clusters' S_j), False / 0 = never, 1 = S_j, True / 2 = S_j and Q.
        `data_limbs` = 2 (default): the problem data (sampled vectors, lambda, dense A_p, B and, in `solvesdp_mw`, C, c, b) are
        passed as double-double, the (hi, lo) pairs a FlatSDP carries; 1: the fp64 roundings only; `limbs`: at the working precision, as the reference holds
        the sampled problem (convert_to_prec, src/interface.jl:1078-1112) -- the planes beyond (hi, lo) come from `FlatSDP.tails` (generators run under
        `clrs_amd.sdp.data_planes(limbs)`; zero when the FlatSDP has none)."""
        self.flat: FlatSDP = sdp if isinstance(sdp, FlatSDP) else flatten(sdp)
        f = self.flat
        self.limbs = int(limbs)
        self.data_limbs = int(data_limbs)
        if self.data_limbs not in (1, 2, self.limbs):
            raise ValueError("data_limbs must be 1, 2 or `limbs` (the problem data at the working precision: FlatSDP.tails, sdp.data_planes)")
        self.L = _lib.load()
        k = self._keep = {}

        def hold(name, arr, dt):
            k[name] = np.ascontiguousarray(arr, dtype=dt)
            return k[name]

        def data(name):
            """planar (data_limbs, len) copy of a data array of the FlatSDP: hi [, lo [, the planes of FlatSDP.tails]]"""
            if self.data_limbs == 1:
                return hold(name, np.ascontiguousarray(getattr(f, name), dtype=np.float64).reshape(-1), np.float64)
            return hold(name, f.data_planes_of(name, self.data_limbs), np.float64)

        self._data = data
        d = _lib.SdpDesc()
        d.n_clusters, d.n_free, d.n_blocks = f.n_clusters, f.n_free, f.n_blocks
        d.cluster_P = hold("cluster_P", f.cluster_P, np.int32).ctypes.data_as(_lib.p_i32)
        d.B = _dp(data("B"))
        for name in ("block_cluster", "block_m", "block_delta", "block_kind", "term_p", "term_r", "term_s", "term_rank", "dense_p"):
            setattr(d, name, hold(name, getattr(f, name), np.int32).ctypes.data_as(_lib.p_i32))
        for name in ("term_ptr", "term_vec_ptr", "dense_ptr", "dense_A_ptr"):
            setattr(d, name, hold(name, getattr(f, name), np.int64).ctypes.data_as(_lib.p_i64))
        for name in ("term_lambda", "term_vs", "term_ws", "dense_A"):
            setattr(d, name, _dp(data(name)))
        h = C.c_void_p()
        opts = _lib.MwOptions(-1 if exact_products is None else (2 if exact_products else 0), -1 if refine is None else int(refine),
                              -1 if pipeline is None else (2 if pipeline is True else int(pipeline)), -1 if refine_predictor is None else int(bool(refine_predictor)),
                              -1 if factor_limbs is None else int(factor_limbs), 0 if matmul_limbs is None else int(matmul_limbs))
        _lib.check(self.L.clrs_mw_create_opts(C.byref(d), self.data_limbs, int(device), self.limbs, C.byref(opts), C.byref(h)))
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

    def get_S(self):
        """S_j and A_Y of the last assembly (also after the device-pointer entry points), planar limbs."""
        f = self.flat
        S = np.empty((self.limbs, f.S_len))
        AY = np.empty((self.limbs, f.n_terms)) if f.n_terms else None
        _lib.check(self.L.clrs_mw_get_S(self.h, _dp(S), _dp(AY) if AY is not None else None))
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

    # -- cluster sharding (one process per GPU): this context holds the clusters of one rank -----------------
    def set_shard(self, rank: int, world: int):
        _lib.check(self.L.clrs_mw_set_shard(self.h, int(rank), int(world)))

    @staticmethod
    def comm_unique_id() -> bytes:
        """ncclGetUniqueId (128 bytes): created on one rank, distributed by the caller, passed to comm_init on every rank."""
        buf = C.create_string_buffer(128)
        _lib.check(_lib.load().clrs_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        """After this call factor_dev / solve_dev are collective: the library all-gathers the partial Q and u itself (RCCL)."""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        _lib.check(self.L.clrs_mw_comm_init(self.h, buf, int(rank), int(world)))

    def comm_init_side(self, unique_id: bytes):
        """Second communicator (its own unique id) for the exchanges the sharded interior-point iteration issues on its side stream."""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        _lib.check(self.L.clrs_mw_comm_init_side(self.h, buf))

    def comm_init_local(self, group: "LocalGroup", rank: int):
        """Rank `rank` of an in-process group (all contexts on one device, one host thread per rank): for single-GPU tests."""
        _lib.check(self.L.clrs_mw_comm_init_local(self.h, group.h, int(rank)))
        self._group = group

    def comm_probe(self, reps: int = 50):
        """microseconds per all-gather of (partial Q, partial u, scalar record) on this context's communicator, and (rank, world, backend): collective"""
        us, info = np.zeros(3), np.zeros(3, dtype=np.int32)
        _lib.check(self.L.clrs_mw_comm_probe(self.h, int(reps), _dp(us), info.ctypes.data_as(_lib.p_i32)))
        return dict(q_us=float(us[0]), u_us=float(us[1]), record_us=float(us[2]), rank=int(info[0]), world=int(info[1]),
                    backend={0: "none", 1: "rccl", 2: "in-process group"}[int(info[2])])

    def comm_destroy(self):
        _lib.check(self.L.clrs_mw_comm_destroy(self.h))

    def factor_local_dev(self):
        _lib.check(self.L.clrs_mw_schur_factor_local_dev(self.h))

    def factor_finish_dev(self):
        _lib.check(self.L.clrs_mw_schur_factor_finish_dev(self.h))

    def solve_fwd_dev(self, d_rhs_x: int):
        _lib.check(self.L.clrs_mw_schur_solve_fwd_dev(self.h, C.c_void_p(d_rhs_x)))

    def solve_bwd_dev(self, d_rhs_y: int, d_dx: int, d_dy: int):
        _lib.check(self.L.clrs_mw_schur_solve_bwd_dev(self.h, C.c_void_p(d_rhs_y) if d_rhs_y else None, C.c_void_p(d_dx),
                                                      C.c_void_p(d_dy) if d_dy else None))

    def solve_refine_dev(self, d_rhs_y: int, d_dx: int, d_dy: int):
        """second half of the refinement step of a split-phase solve: after solve_bwd_dev and one more exchange of the u gather slots"""
        _lib.check(self.L.clrs_mw_schur_solve_refine_dev(self.h, C.c_void_p(d_rhs_y) if d_rhs_y else None, C.c_void_p(d_dx),
                                                         C.c_void_p(d_dy) if d_dy else None))

    def q_gather(self) -> int:
        return int(self.L.clrs_mw_q_gather_dev(self.h) or 0)

    def u_gather(self) -> int:
        return int(self.L.clrs_mw_u_gather_dev(self.h) or 0)

    def sync_status(self) -> int:
        return _lib.check(self.L.clrs_mw_sync_status(self.h))

    def sync_status_cholesky(self) -> int:
        return _lib.check(self.L.clrs_mw_sync_status_cholesky(self.h))

    def stream(self) -> int:
        return int(self.L.clrs_mw_stream(self.h) or 0)

    def set_stream(self, hip_stream: int):
        _lib.check(self.L.clrs_mw_set_stream(self.h, C.c_void_p(hip_stream)))


class LocalGroup:
    """clrs_mw_local_group: the in-process stand-in for the RCCL communicators (`world` contexts on one device, one thread each)."""

    def __init__(self, world: int, device: int = 0):
        self.L = _lib.load()
        self.world = int(world)
        h = C.c_void_p()
        _lib.check(self.L.clrs_mw_local_group_create(self.world, int(device), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.clrs_mw_local_group_destroy(self.h)
            self.h = None


def shard_problem(full: FlatSDP, rank: int, world: int, parts=None):
    """The sub-problem of rank `rank` (its clusters by `partition_clusters`, all free variables) and what `solvesdp_mw` must tell the
    library about the whole: (shard, shard_info)."""
    from .sdp import shard_clusters
    from .sharded import partition_clusters
    parts = partition_clusters(full, world) if parts is None else parts
    mine = parts[rank]
    if not mine:
        raise ValueError(f"rank {rank} of {world} holds no cluster: use at most {full.n_clusters} ranks")
    shard = shard_clusters(full, mine)
    blocks = [b for b in range(full.n_blocks) if int(full.block_cluster[b]) in set(mine)]
    info = dict(rows_global=int(np.sum(full.block_n)), clusters_global=int(full.n_clusters),
                cluster_ids=np.asarray(mine, dtype=np.int32), block_ids=np.asarray(blocks, dtype=np.int32))
    return shard, info


# ------------------------------------------------------------------------------------------------
# solvesdp at the reference's precision, device resident (clrs_mw_ipm_*)
# ------------------------------------------------------------------------------------------------

def solvesdp_mw(sdp, limbs: Optional[int] = None, prec: Optional[int] = None, ctx: Optional[MwSchurContext] = None, device: int = 0, data_limbs: int = 2,
                maxiterations: int = 500, beta_infeasible: float = 0.3, beta_feasible: float = 0.1, gamma: float = 0.9,
                omega_p: float = 1e10, omega_d: float = 1e10, duality_gap_threshold: float = 1e-15,
                dual_error_threshold: float = 1e-30, primal_error_threshold: float = 1e-30, max_complementary_gap: float = 1e100,
                need_dual_feasible: bool = False, need_primal_feasible: bool = False, verbose: bool = False,
                step_length_threshold: float = 1e-7, safe_step: bool = True, step_by_step: bool = False, shard_info: Optional[dict] = None,
                dualsol=None, primalsol=None, factor_limbs: Optional[int] = None, matmul_prec: Optional[int] = None, correctoronly: bool = False):
    """`solvesdp(sdp; prec, ...)` (src/solver.jl:71-127) with the whole loop body on the GPU in multi-word fp64.
    `correctoronly`: the reference's keyword (src/solver.jl:121, 370-374, 945): mu_p = mu, and the loop ends on `need_dual_feasible` / `need_primal_feasible`, an
    error or `maxiterations` only.
    `matmul_prec`: the reference's keyword (src/solver.jl:125): bits of the products that form the pairing matrices (rounded up to whole limbs; ignored when
    `ctx` is given -- create it with `matmul_limbs`).
    `factor_limbs`: see `MwSchurContext` (ignored when `ctx` is given); `timings["refine_bits"]` of the result lists, per iteration, the bits the first
    pass of the corrector's refined solve was good to.

    Keywords and DEFAULTS are the reference's (omega = 1e10, gap 1e-15, errors 1e-30: they assume its 256-bit arithmetic):
    `prec` bits select the limb count (`limbs_for_precision`), or pass `limbs` directly; the default is limbs = 5, which
    covers prec = 256.  The result's x, y, X, Y are planar limbs; objectives are fp64 heads plus `objectives_limbs`.
    `dualsol` / `primalsol`: the warm start of src/solver.jl:202-239 (applied, as there, only when BOTH are given): `dualsol` supplies x and X,
    `primalsol` y and Y -- a previous `SolveResult` (or anything with those attributes), fp64 or planar limbs; through `clrs_mw_ipm_set`.  The errors of
    the starting iterate are known after the first iteration (the reference computes them before its loop): a warm-started solve runs at least one.
    Termination (src/solver.jl:921-950): by the library and the device together in one call (`clrs_mw_ipm_solve_cb`; `verbose` prints the table rows
    from its callback), or -- `step_by_step` -- on the host from one record per call of `clrs_mw_ipm_iterate`."""
    import time
    from .solver import SolveResult
    f = sdp if isinstance(sdp, FlatSDP) else flatten(sdp)
    if limbs is None:
        limbs = limbs_for_precision(prec) if prec is not None else 5
    own_ctx = ctx is None
    if ctx is None:
        ctx = MwSchurContext(f, limbs=limbs, device=device, data_limbs=data_limbs, factor_limbs=factor_limbs,
                             matmul_limbs=None if matmul_prec is None else min(limbs, limbs_for_precision(matmul_prec)))
    K = ctx.limbs
    L = ctx.L
    keep = [ctx._data("C"), ctx._data("c"), ctx._data("b") if f.n_free else np.zeros((ctx.data_limbs, 1))]
    data = _lib.IpmData(_dp(keep[0]), _dp(keep[1]), _dp(keep[2]), int(f.maximize), 0, float(f.constant))
    _lib.check(L.clrs_mw_ipm_create_ex(ctx.h, C.byref(data), ctx.data_limbs))
    prm = _lib.IpmParams(beta_infeasible, beta_feasible, gamma, dual_error_threshold, primal_error_threshold, max_complementary_gap,
                         step_length_threshold, int(safe_step), int(bool(correctoronly)))
    _lib.check(L.clrs_mw_ipm_set_params(ctx.h, C.byref(prm)))
    if shard_info is not None:
        # `sdp` is this rank's share of a cluster-sharded problem (shard_problem) and `ctx` carries the rank's communicators: every call
        # below is collective; x, X, Y of the result are the shard's, y and every scalar are identical on all ranks
        cid, bid = np.ascontiguousarray(shard_info["cluster_ids"], np.int32), np.ascontiguousarray(shard_info["block_ids"], np.int32)
        _lib.check(L.clrs_mw_ipm_set_global(ctx.h, int(shard_info["rows_global"]), int(shard_info["clusters_global"]),
                                            cid.ctypes.data_as(_lib.p_i32), bid.ctypes.data_as(_lib.p_i32)))
    _lib.check(L.clrs_mw_ipm_init(ctx.h, float(omega_p), float(omega_d)))
    warm = dualsol is not None and primalsol is not None

    def _limbs_of(a, n):
        a = np.asarray(a, dtype=np.float64)
        a = a.reshape(1, -1) if a.ndim == 1 else a
        if a.shape[1] != n:
            raise ValueError(f"warm start: expected {n} numbers per limb plane, got {a.shape}")
        out = np.zeros((K, max(n, 1)))
        out[:min(K, a.shape[0]), :n] = a[:K]
        return out

    if warm:
        ws = [_limbs_of(dualsol.x, f.x_len), _limbs_of(primalsol.y, f.n_free), _limbs_of(dualsol.X, f.xy_len), _limbs_of(primalsol.Y, f.xy_len)]
        _lib.check(L.clrs_mw_ipm_set(ctx.h, _dp(ws[0]), _dp(ws[1]) if f.n_free else None, _dp(ws[2]), _dp(ws[3])))
    rec = _lib.IpmRecord()
    hist = []
    refine_bits = []
    t_start = time.time()
    error_code, it = 0, 1
    dual_error = primal_error = np.inf      # computed by the first iteration; no termination test can pass before
    gap = 0.0                               # x = 0, y = 0: both objectives equal the constant (src/solver.jl:319-321)
    d_obj = p_obj = f.constant
    if warm:                                # objectives of the starting iterate (src/solver.jl:319-321)
        o3 = np.zeros(3 * K)
        _lib.check(L.clrs_mw_ipm_objectives(ctx.h, _dp(o3)))
        d_obj, p_obj, gap = float(o3[0]), float(o3[K]), float(o3[2 * K])
    pd_feas = False

    def row(r):
        return [it, r.mu, d_obj, p_obj, gap, r.max_P, r.max_p, r.max_d, r.alpha_d, r.alpha_p, r.beta_c]

    if not step_by_step:
        # the whole loop in ONE call (clrs_mw_ipm_solve_cb): the library enqueues iterations one ahead of the record it waits for and the
        # device evaluates the termination test of src/solver.jl:921-950 itself; the records come back as the reference's table rows, and with
        # `verbose` a callback prints each row (:566-582) as the host reads it
        stop = _lib.IpmStop(float(duality_gap_threshold), int(bool(need_dual_feasible)), int(bool(need_primal_feasible)), int(maxiterations), 0)
        recs = (_lib.IpmRecord * max(int(maxiterations), 1))()
        n_it, err = C.c_int(0), C.c_int(0)
        shown = dict(it=1, d=d_obj, p=p_obj, g=gap)

        def _row(recp, _user):
            q = recp.contents
            print("%5d %8.1f %11.3e %11.3e %11.3e %10.2e %10.2e %10.2e %10.2e %10.2e %10.2e %10.2e" %
                  (shown["it"], time.time() - t_start, q.mu, shown["d"], shown["p"], shown["g"], q.max_P, q.max_p, q.max_d, q.alpha_d, q.alpha_p, q.beta_c), flush=True)
            if not q.error_code:
                shown.update(d=q.d_obj, p=q.p_obj, g=q.gap)
            shown["it"] += 1
        cb = _lib.RecordFn(_row) if verbose else _lib.RecordFn()
        _lib.check(L.clrs_mw_ipm_solve_cb(ctx.h, C.byref(stop), cb, None, recs, int(maxiterations), C.byref(n_it), C.byref(err)))
        error_code = err.value
        for i in range(n_it.value):
            r = recs[i]
            hist.append(row(r))
            refine_bits.append(int(r.refine_bits))
            dual_error, primal_error, pd_feas = r.dual_error, r.primal_error, bool(r.pd_feas)
            if r.error_code:
                break
            d_obj, p_obj, gap = r.d_obj, r.p_obj, r.gap
            it += 1
        if error_code == 2:
            it = n_it.value + 1
    else:
        while True:
            dual_feas, primal_feas = dual_error < dual_error_threshold, primal_error < primal_error_threshold
            if (need_dual_feasible and dual_feas) or (need_primal_feasible and primal_feas):          # src/solver.jl:921-950
                break
            if not correctoronly and dual_feas and primal_feas and gap < duality_gap_threshold:      # (:945)
                break
            if it > maxiterations:
                error_code = 2
                break
            _lib.check(L.clrs_mw_ipm_iterate(ctx.h, C.byref(rec)))
            hist.append(row(rec))
            refine_bits.append(int(rec.refine_bits))
            if verbose:
                print("%5d %8.1f %11.3e %11.3e %11.3e %10.2e %10.2e %10.2e %10.2e %10.2e %10.2e %10.2e" %
                      (it, time.time() - t_start, rec.mu, d_obj, p_obj, gap, rec.max_P, rec.max_p, rec.max_d, rec.alpha_d, rec.alpha_p, rec.beta_c))
            dual_error, primal_error, pd_feas = rec.dual_error, rec.primal_error, bool(rec.pd_feas)
            if rec.error_code:
                error_code = rec.error_code
                if verbose and rec.error_code == 1:
                    print("SolverFailure: factor status %d, Cholesky status %d" % (rec.factor_status, rec.cholesky_status))
                break
            d_obj, p_obj, gap = rec.d_obj, rec.p_obj, rec.gap
            it += 1
    t_total = time.time() - t_start
    x, y = np.zeros((K, f.x_len)), np.zeros((K, max(f.n_free, 1)))
    X, Y = np.zeros((K, f.xy_len)), np.zeros((K, f.xy_len))
    _lib.check(L.clrs_mw_ipm_get(ctx.h, _dp(x), _dp(y), _dp(X), _dp(Y)))
    obj = np.zeros(3 * K)
    _lib.check(L.clrs_mw_ipm_objectives(ctx.h, _dp(obj)))
    if pd_feas and gap < duality_gap_threshold:                                    # src/solver.jl:727-741
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
    res = SolveResult(status, x, X, y[:, :f.n_free], Y, t_total, error_code, it - 1, d_obj, p_obj, gap, dual_error, primal_error,
                      np.array(hist).reshape(-1, 11), dict(loop="device", limbs=K))
    res.timings["objectives_limbs"] = obj.reshape(3, K)
    res.timings["refine_bits"] = refine_bits
    return res
