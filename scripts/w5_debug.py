"""k_cluster_assemble_w5 against the general kernel and against a numpy restatement of its formulas, pair-block by pair-block of S_j,
on Nsphere_packing(8, 15, [1/2, 1/2]): the diagnostic that found the kernel's first bug (S_off of the cluster's first block)."""
import sys; sys.path.insert(0,'.')
import numpy as np
from tests.util import flat, spd_iterates, chol_blocks_np
from clrs_amd.solver import SchurContext
f = flat("ns_8_15_2")
X, Y = spd_iterates(f, seed=1)
Xc = chol_blocks_np(f, X)
c0 = SchurContext(f, wave5=False); S0, AY0 = c0.compute_S_integrated(Xc, Y); c0.close()
c1 = SchurContext(f); print("w5 clusters", c1.wave5_clusters()); S1, AY1 = c1.compute_S_integrated(Xc, Y); c1.close()
j = 1
P = int(f.cluster_P[j]); sl = slice(int(f.S_off[j]), int(f.S_off[j+1]))
A = S0[sl].reshape(P, P, order="F"); B = S1[sl].reshape(P, P, order="F")
U = P // 3
print("scale", np.max(np.abs(A)))
for i in range(3):
    for jj in range(3):
        a = A[i*U:(i+1)*U, jj*U:(jj+1)*U]; b = B[i*U:(i+1)*U, jj*U:(jj+1)*U]
        print("pair-block", i, jj, "max err %.3e" % np.max(np.abs(a-b)), "ref max %.3e" % np.max(np.abs(a)), "ratio sample", (b[3,5]/a[3,5]), (b[20,5]/a[20,5]), (b[5,20]/a[5,20]))
print("sym err", np.max(np.abs(B-B.T)))
t0, t1 = int(f.term_ptr[1]), int(f.term_ptr[3])
print("AY err", np.max(np.abs(AY0[t0:t1]-AY1[t0:t1])), "ref", np.max(np.abs(AY0[t0:t1])))
d = np.abs(AY0[t0:t1]-AY1[t0:t1]); print("AY bad idx", np.nonzero(d > 1e-9*np.max(np.abs(AY0[t0:t1])))[0][:20])
a = A[0:U, 0:U]; b = B[0:U, 0:U]
bad = np.abs(a-b) > 1e-9*np.max(np.abs(a))
print("bad count", bad.sum(), "of", bad.size)
print("bad rows", sorted(set(np.nonzero(bad)[0])))
print("bad cols", sorted(set(np.nonzero(bad)[1])))
np.set_printoptions(linewidth=250, precision=3)
print((bad[:, :]).astype(int))
# single-block check: contribution of each block separately
# numpy restatement of the w5 formulas, block by block
contrib = {}
for b in (1, 2):
    n = int(f.block_n[b]); off = int(f.block_off[b])
    Xb = X[off:off + n * n].reshape(n, n, order="F"); Yb = Y[off:off + n * n].reshape(n, n, order="F")
    t0, t1 = int(f.term_ptr[b]), int(f.term_ptr[b + 1])
    V = np.zeros((16, U)); lam = np.zeros((3, U))
    for t in range(t0, t1):
        r, s_ = int(f.term_r[t]), int(f.term_s[t]); p_ = int(f.term_p[t])
        lam[r + s_, p_ % U] = f.term_lambda[t]
        if r == 0 and s_ == 0:
            vp = int(f.term_vec_ptr[t]); V[:, p_] = f.term_vs[vp:vp + 16]
    Xi = np.linalg.inv(Xb)
    G = lambda M, a, c: V.T @ M[16 * a:16 * a + 16, 16 * c:16 * c + 16] @ V
    AAx, ABx, BAx, BBx = G(Xi, 0, 0), G(Xi, 0, 1), G(Xi, 1, 0), G(Xi, 1, 1)
    AAy, ABy, BAy, BBy = G(Yb, 0, 0), G(Yb, 0, 1), G(Yb, 1, 0), G(Yb, 1, 1)
    Sb = np.zeros((P, P))
    L = lambda i, j: np.outer(lam[i], lam[j])
    Sb[0:U, 0:U] = L(0, 0) * AAx * AAy
    Sb[U:2*U, 0:U] = L(1, 0) * (AAx * BAy + BAx * AAy)
    Sb[2*U:, 0:U] = L(2, 0) * BAx * BAy
    Sb[U:2*U, U:2*U] = L(1, 1) * (AAx * BBy + BBx * AAy + ABx * BAy + BAx * ABy)
    Sb[2*U:, U:2*U] = L(2, 1) * (BAx * BBy + BBx * BAy)
    Sb[2*U:, 2*U:] = L(2, 2) * BBx * BBy
    contrib[b] = np.tril(Sb) + np.tril(Sb, -1).T
Sref = contrib[1] + contrib[2]
print("numpy formulas vs general kernel: max err %.3e (scale %.3e)" % (np.max(np.abs(Sref - A)), np.max(np.abs(A))))
print("numpy formulas vs w5: max err %.3e" % np.max(np.abs(Sref - B)))
print("w5 - block1 only: %.3e;  w5 - block2 only: %.3e" % (np.max(np.abs(B - contrib[1])), np.max(np.abs(B - contrib[2]))))
E = B - Sref
for i in range(3):
    for jj in range(i + 1):
        e = E[i*U:(i+1)*U, jj*U:(jj+1)*U]; c1 = contrib[1][i*U:(i+1)*U, jj*U:(jj+1)*U]; c2 = contrib[2][i*U:(i+1)*U, jj*U:(jj+1)*U]
        print("pair-block", i, jj, "err %.3e" % np.max(np.abs(e)), "| err vs -c1 %.3e" % np.max(np.abs(e + c1)), "| err vs -c2 %.3e" % np.max(np.abs(e + c2)))
E00 = E[0:U, 0:U]
for ti in range(2):
    for tj in range(2):
        e = E00[16*ti:16*ti+16, 16*tj:16*tj+16]; r2 = contrib[2][16*ti:16*ti+16, 16*tj:16*tj+16]
        print("tile", ti, tj, "max|E| %.3e  max|c2| %.3e  max rel %.3e" % (np.max(np.abs(e)), np.max(np.abs(r2)), np.max(np.abs(e) / (np.abs(r2) + 1e-30))))
print("E00 diag", np.diag(E00)[:8], np.diag(E00)[24:])
print("c2  diag", np.diag(contrib[2])[:8], np.diag(contrib[2])[24:32])
print("c1  diag", np.diag(contrib[1])[:8])
for (r_, c_) in ((0, 0), (5, 3), (3, 5), (20, 4), (40, 2), (40, 36), (70, 3), (70, 40), (90, 80)):
    print((r_, c_), "w5 %.6e  general %.6e  c1 %.6e  c2 %.6e" % (B[r_, c_], A[r_, c_], contrib[1][r_, c_], contrib[2][r_, c_]))
