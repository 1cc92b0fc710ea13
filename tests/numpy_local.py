"""CPU stand-in for the local compute object of clrs_amd.sharded.ShardedSchur (test infrastructure).

Same interface as `HipLocal`, numpy + the CPU oracle underneath, torch CPU tensors as buffers, so that the
orchestration (partition, shard extraction, the two all-reduces, status reduction) can be exercised over
`gloo` in a container without GPUs.  Never used by the product."""
import numpy as np
import scipy.linalg as sla
import torch


class NumpyLocal:
    def __init__(self, shard):
        from oracle.oracle import Oracle
        self.flat = shard
        self.o = Oracle(shard, quad=False)
        self.N = shard.n_free
        self.qu = torch.zeros(self.N * self.N + self.N, dtype=torch.float64)      # [Q | u] in one buffer, like HipLocal
        self.q, self.u = self.qu[:self.N * self.N], self.qu[self.N * self.N:]
        self.st = 0

    def _B(self, j):
        f, N = self.flat, self.N
        P = int(f.cluster_P[j])
        return f.B[int(f.cluster_off[j]) * N:int(f.cluster_off[j + 1]) * N].reshape(P, N, order="F")

    def assemble(self, Xchol, Y):
        self.S, _ = self.o.schur_assemble(Xchol.numpy(), Y.numpy())

    def factor_local(self):
        f = self.flat
        self.L, self.LB = [], []
        Q = np.zeros((self.N, self.N))
        self.st = 0
        for j in range(f.n_clusters):
            P = int(f.cluster_P[j])
            Sj = self.S[f.S_off[j]:f.S_off[j + 1]].reshape(P, P, order="F")
            try:
                L = np.linalg.cholesky(Sj)
            except np.linalg.LinAlgError:
                self.st = self.st or j + 1
                L = np.eye(P)
            LB = sla.solve_triangular(L, self._B(j), lower=True) if self.N else np.zeros((P, 0))
            self.L.append(L); self.LB.append(LB)
            Q += LB.T @ LB
        self.q.copy_(torch.from_numpy(Q.reshape(-1)))
        return self.q

    def factor_finish(self):
        if self.N:
            try:
                self.LQ = np.linalg.cholesky(self.q.numpy().reshape(self.N, self.N))
            except np.linalg.LinAlgError:
                self.st = self.st or self.flat.n_clusters + 1

    def solve_fwd(self, rhs_x):
        f = self.flat
        r = rhs_x.numpy()
        self.t = [sla.solve_triangular(self.L[j], r[f.cluster_off[j]:f.cluster_off[j + 1]], lower=True) for j in range(f.n_clusters)]
        u = np.zeros(self.N)
        for j in range(f.n_clusters):
            u += self.LB[j].T @ self.t[j]
        self.u.copy_(torch.from_numpy(u))
        return self.u

    def solve_bwd(self, rhs_y, dx, dy):
        f = self.flat
        if self.N:
            y = sla.cho_solve((self.LQ, True), rhs_y.numpy() - self.u.numpy())
            dy.copy_(torch.from_numpy(y))
        out = np.zeros(f.x_len)
        for j in range(f.n_clusters):
            t = self.t[j] + (self.LB[j] @ y if self.N else 0.0)
            out[f.cluster_off[j]:f.cluster_off[j + 1]] = sla.solve_triangular(self.L[j], t, lower=True, trans="T")
        dx.copy_(torch.from_numpy(out))

    def status(self):
        return self.st
