"""Cluster-sharded hot path: one process per GPU, clusters partitioned over the ranks.

The clusters j are the reference's own outer parallel axis (`Threads.@threads for j in j_order`,
src/solver.jl:1245,1257,1537,1566): X_jl, Y_jl, S_j, L_j, LinvB_j, dx_j are cluster-local.  The only
couplings are sums over all clusters:

    Q  = sum_j LinvB_j^T LinvB_j          (src/solver.jl:1268-1269)   one all-reduce of N x N per factorisation
    u  = sum_j LinvB_j^T t_j              (src/solver.jl:1550-1553)   one all-reduce of N per solve

so each rank assembles / factors its own clusters, the partial Q (and u) is summed with ONE
collective (RCCL all-reduce over xGMI; messages are KB-sized, i.e. latency bound), and every rank
factors the N x N matrix Q redundantly.  dy is replicated, dx stays sharded.

`ShardedSchur` is the orchestration; the local compute object is injected: `HipLocal` (the product:
libclrs_hip.so through the device-pointer C ABI, buffers aliased as torch tensors so that
torch.distributed can reduce them in place) -- tests drive the same orchestration over `gloo` with
a numpy stand-in defined under tests/.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from .sdp import FlatSDP, shard_clusters


def cluster_weights(flat: FlatSDP) -> np.ndarray:
    """Cost model per cluster: sum_l (4 n^2 U + 4 n U^2) + P^3/3 + P^2 N  (replaces the
    n^3 / P^3 weights of ThreadingInfo, src/threadinginfo.jl:87-100).  U is bounded by the term count."""
    w = np.zeros(flat.n_clusters)
    for b in range(flat.n_blocks):
        j, n = int(flat.block_cluster[b]), float(flat.block_n[b])
        if flat.block_kind[b] == 0:
            U = float(min(flat.term_ptr[b + 1] - flat.term_ptr[b], flat.cluster_P[j] * flat.block_m[b]))
            w[j] += 4 * n * n * U + 4 * n * U * U
        else:
            cnt = float(flat.dense_ptr[b + 1] - flat.dense_ptr[b])
            w[j] += cnt * 6 * n ** 3 + cnt * cnt * n * n
    P = flat.cluster_P.astype(np.float64)
    return w + P ** 3 / 3 + P ** 2 * flat.n_free


def partition_clusters(flat: FlatSDP, world: int) -> List[List[int]]:
    """Greedy longest-processing-time assignment of clusters to ranks (replaces
    distribute_weights_swapping, src/threadinginfo.jl:4-57).  Deterministic; every rank gets its
    clusters in increasing order; ranks may be empty when world > n_clusters."""
    w = cluster_weights(flat)
    order = sorted(range(flat.n_clusters), key=lambda j: (-w[j], j))
    load = [0.0] * world
    parts: List[List[int]] = [[] for _ in range(world)]
    for j in order:
        r = min(range(world), key=lambda k: (load[k], k))
        parts[r].append(j)
        load[r] += w[j]
    return [sorted(p) for p in parts]


class _DevArray:
    """Alias of a raw device pointer for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


class HipLocal:
    """One rank's clusters on one MI355X through the device-pointer entry points of include/clrs_hip.h.
    All work is enqueued on torch's current stream, which is also what torch.distributed (RCCL)
    orders its collectives against; nothing here synchronises."""

    def __init__(self, shard: FlatSDP, device: int, graph: bool = False):
        import torch
        from .solver import SchurContext
        self.torch = torch
        self.flat = shard
        self.device = device
        torch.cuda.set_device(device)
        self.ctx = SchurContext(shard, device=device)
        if graph and torch.cuda.current_stream().cuda_stream == 0:
            # the legacy (null) stream cannot be captured: hipStreamBeginCapture fails on it
            raise ValueError("HipLocal(graph=True) needs a non-default current stream: wrap the calls in "
                             "`with torch.cuda.stream(torch.cuda.Stream()):` or call torch.cuda.set_stream first")
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        if graph:
            self.ctx.set_graph_mode(True)
        N = shard.n_free
        dev = f"cuda:{device}"
        # Q (N x N) and u (N) are one device buffer (include/clrs_hip.h): `qu` is what a merged exchange reduces
        self.qu = torch.as_tensor(_DevArray(self.ctx.q_buffer(), max(N * N + N, 1)), device=dev)[:N * N + N]
        self.q, self.u = self.qu[:N * N], self.qu[N * N:]
        assert N == 0 or self.ctx.u_buffer() == self.ctx.q_buffer() + 8 * N * N

    def cholesky_blocks(self, X, Xchol):
        self.ctx.cholesky_blocks_dev(X.data_ptr(), Xchol.data_ptr())

    def assemble(self, Xchol, Y):
        self.ctx.assemble_dev(Xchol.data_ptr(), Y.data_ptr())

    def factor_all(self):
        """Single-rank shortcut: no exchange needed, fewer launches."""
        self.ctx.factor_dev()

    def solve_all(self, rhs_x, rhs_y, dx, dy):
        self.ctx.solve_dev(rhs_x.data_ptr(), rhs_y.data_ptr() if rhs_y is not None and rhs_y.numel() else 0,
                           dx.data_ptr(), dy.data_ptr() if dy is not None and dy.numel() else 0)

    def factor_local(self):
        self.ctx.factor_local_dev()
        return self.q

    def factor_finish(self):
        self.ctx.factor_finish_dev()

    def solve_fwd(self, rhs_x):
        self.ctx.solve_fwd_dev(rhs_x.data_ptr())
        return self.u

    def solve_bwd(self, rhs_y, dx, dy):
        self.ctx.solve_bwd_dev(rhs_y.data_ptr() if rhs_y is not None and rhs_y.numel() else 0,
                               dx.data_ptr(), dy.data_ptr() if dy is not None and dy.numel() else 0)

    def status(self) -> int:
        return self.ctx.sync_status()

    def close(self):
        self.ctx.close()


class ShardedSchur:
    """Orchestration of the sharded path for one rank.

    `flat` is the FULL problem (every rank builds or loads the same description), `local_factory(shard)`
    creates the local compute object for this rank's clusters, `group` is a torch.distributed
    process group (None = default group; with world == 1 no collective is issued)."""

    def __init__(self, flat: FlatSDP, rank: int, world: int, local_factory, group=None,
                 parts: Optional[Sequence[Sequence[int]]] = None, force_split: bool = False, defer_q: bool = True):
        self.full = flat
        self.rank, self.world = rank, world
        self.force_split = force_split      # exercise the split-phase calls + collectives even with one rank
        self.parts = [list(p) for p in (parts if parts is not None else partition_clusters(flat, world))]
        assert sorted(j for p in self.parts for j in p) == list(range(flat.n_clusters)), "partition must cover every cluster once"
        self.clusters = self.parts[rank]
        self.shard = shard_clusters(flat, self.clusters)
        self.local = local_factory(self.shard)
        self.group = group
        self.N = flat.n_free
        self.defer_q = defer_q
        self._q = None                           # partial Q whose exchange is still pending

    # -- index helpers ---------------------------------------------------------------------------
    def xy_slices(self):
        """Slices of this rank's blocks in the full X/Y layout (in shard order)."""
        f = self.full
        mine = set(self.clusters)
        return [slice(int(f.block_off[b]), int(f.block_off[b + 1])) for b in range(f.n_blocks) if int(f.block_cluster[b]) in mine]

    def x_slices(self):
        f = self.full
        return [slice(int(f.cluster_off[j]), int(f.cluster_off[j + 1])) for j in self.clusters]

    def take_xy(self, full_vec: np.ndarray) -> np.ndarray:
        sl = self.xy_slices()
        return np.concatenate([full_vec[s] for s in sl]) if sl else np.zeros(0)

    def take_x(self, full_vec: np.ndarray) -> np.ndarray:
        sl = self.x_slices()
        return np.concatenate([full_vec[s] for s in sl]) if sl else np.zeros(0)

    # -- the path --------------------------------------------------------------------------------
    def _all_reduce(self, t):
        if (self.world > 1 or self.force_split) and t.numel():
            import torch.distributed as dist
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def decompose(self, Xchol, Y):
        """compute_T_decomposition! (src/solver.jl:1229-1287) on this rank's clusters + the one exchange."""
        self.local.assemble(Xchol, Y)
        if self.world == 1 and not self.force_split and hasattr(self.local, "factor_all"):
            self.local.factor_all()
            return
        self._q = self.local.factor_local()      # partial Q of this rank; its exchange is deferred to the first solve (or to status())
        if not self.defer_q:
            self._finish_factor()

    def _finish_factor(self):
        if self._q is not None:
            self._all_reduce(self._q)            # Q = sum over ranks of the partial Q
            self.local.factor_finish()
            self._q = None

    def solve(self, rhs_x, rhs_y, dx, dy):
        """Solve stage of compute_search_direction! (src/solver.jl:1527-1582): rhs_x, dx sharded; rhs_y, dy replicated.
        The first solve after a factorisation also carries the exchange of Q: both Q and u = sum_j LinvB_j^T t_j are needed at the
        same point (dy = Q^-1 (rhs_y - u)), t_j needs only the cluster-local factors, so ONE all-reduce of [Q | u] replaces two."""
        if self.world == 1 and not self.force_split and hasattr(self.local, "solve_all"):
            self.local.solve_all(rhs_x, rhs_y, dx, dy)
            return
        u = self.local.solve_fwd(rhs_x)
        if self._q is not None and getattr(self.local, "qu", None) is not None:
            self._all_reduce(self.local.qu)      # [Q | u] in one collective
            self.local.factor_finish()
            self._q = None
        else:
            self._finish_factor()
            self._all_reduce(u)                  # u = sum over ranks of LinvB_j^T t_j
        self.local.solve_bwd(rhs_y, dx, dy)

    def status(self) -> int:
        """Factorisation status over all ranks: 0, or the smallest failing code in GLOBAL cluster numbering
        (j+1 for S_j, n_clusters+1 for Q), like clrs_schur_factor."""
        self._finish_factor()                    # a pending Q is exchanged and factored now
        st = self.local.status()
        J_local, J = self.shard.n_clusters, self.full.n_clusters
        if st > 0:
            st = (self.clusters[st - 1] + 1) if st <= J_local else J + 1
        if self.world > 1:
            import torch
            import torch.distributed as dist
            big = 2 ** 30
            t = torch.tensor([st if st > 0 else big], dtype=torch.int64)
            backend = dist.get_backend(self.group)
            if backend == "nccl":
                t = t.cuda()
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
            st = int(t.item())
            st = 0 if st == big else st
        return st

    def close(self):
        if hasattr(self.local, "close"):
            self.local.close()


# ------------------------------------------------------------------------------------------------------------------------------------
# Host-side mirror of the scalar exchanges of the sharded interior-point iteration (csrc/clrs_mw_ipm.hip.h: MWG_*, k_mwi_gpack, mwi_gsum;
# csrc/clrs_mw_ipm_host.inc: mw_ipm_exchange).  The product runs them inside the C ABI on RCCL (or its in-process group); this mirror
# states the protocol -- slot layout, which stream's channel carries which stage, all-gather into rank-ordered slots, reduction in rank
# order -- in a form the CPU tests can drive over `gloo` and check against the C++ sources.
# ------------------------------------------------------------------------------------------------------------------------------------
IPM_EXCHANGE_SCHEDULE = (            # (scalar stage, stream) in issue order within one iteration of a running solve; "S" = side stream, "M" = the context's stream
    (15, "S"),                       # objectives of the iterate the last update produced (the tail of the previous iteration leads this one's side work) AND, in the
                                     # same record, <X,Y> of that iterate -> mu of this iteration, AND this rank's -B^T x and max|P| of that iterate -> p, errors
                                     # (stages 4 + 0 + 1: one exchange on the side stream since round 4)          src/solver.jl:793-804, 369, 899-916
    (2, "M"),                        # <X,dY> + <dX,Y> + <dX,dY>, max|d|, status words -> beta_c, errors           :429-434, 441-442
    (3, "M"),                        # smallest eigenvalues -> step lengths                         :1684-1686
)
# the first iteration of a solve has no tail in front of it: its <X,Y> travels alone (stage 0) and so do its residuals (stage 1: -B^T x, max|P|, max|d|);
# the last tail (mw_ipm_finish) has no iteration behind it: stage 4 alone
IPM_EXCHANGE_FIRST, IPM_EXCHANGE_LAST = ((0, "S"), (1, "S")), (4, "S")


def ipm_slot_layout(K: int, N: int) -> dict:
    """Offsets (in doubles) of one rank's record: S1, S2 K-limb sums, BX = K x N planar limbs, D = 8 plain doubles, XY = K limbs (<X,Y> of stage 15);
    LEN = slot length."""
    return dict(S1=0, S2=K, BX=2 * K, D=2 * K + K * N, XY=2 * K + K * N + 8, LEN=3 * K + K * N + 8)


class ScalarExchange:
    """One rank's view of the exchange: `gather(slot)` all-gathers the rank's record into rank-ordered slots (torch.distributed on the
    given group, any backend), the `reduce_*` functions fold the slots in RANK ORDER, so every rank computes identical bits."""

    def __init__(self, K: int, N: int, rank: int, world: int, group=None):
        self.K, self.N, self.rank, self.world, self.group = K, N, rank, world, group
        self.lay = ipm_slot_layout(K, N)

    def new_slot(self) -> np.ndarray:
        return np.zeros(self.lay["LEN"])

    def gather(self, slot: np.ndarray) -> np.ndarray:
        import torch
        import torch.distributed as dist
        mine = torch.from_numpy(np.ascontiguousarray(slot, dtype=np.float64))
        out = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(out, mine, group=self.group)
        return np.stack([t.numpy() for t in out])

    def reduce_sum(self, slots: np.ndarray, field: str, count: int = 1) -> np.ndarray:
        """Sum over the ranks, in rank order, of `count` K-limb numbers stored planar at `field` (limb l of number i at off + l * count + i);
        returned as exact sums of the limbs rounded once to fp64 per number (the device keeps K limbs; the order is what matters here)."""
        import math
        off, K = self.lay[field], self.K
        out = np.zeros(count)
        for i in range(count):
            terms = []
            for r in range(self.world):                      # rank order
                terms += [slots[r, off + l * count + i] for l in range(K)]
            out[i] = math.fsum(terms)
        return out

    def reduce_max(self, slots: np.ndarray, index: int) -> float:
        v = 0.0
        for r in range(self.world):
            v = max(v, float(slots[r, self.lay["D"] + index]))
        return v

    def reduce_min(self, slots: np.ndarray, index: int) -> float:
        v = float(slots[0, self.lay["D"] + index])
        for r in range(1, self.world):
            v = min(v, float(slots[r, self.lay["D"] + index]))
        return v
