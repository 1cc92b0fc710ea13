// clrs_fused.hip.h -- fused small-problem kernels (gfx950): one workgroup per cluster, everything LDS resident.
//
// The named BASELINE configurations have PSD blocks of side n <= 32 and clusters of P <= 96 constraints: a whole
// cluster's Schur assembly fits in one CU's 160 KiB LDS, and the work is bounded by launch latency and by the
// HBM bytes of the iterates, not by flops.  So instead of one launch per BLAS-like stage (clrs_kernels.hip.h,
// kept for blocks that do not fit), ONE launch does, per cluster j and for each of its blocks l in order:
//
//   load L_X, Y, V (coalesced)                                        HBM -> LDS, each byte read once
//   T_Y = Y V             MFMA f64 16x16x4                            (src/solver.jl:1125)
//   G_Y = W^T T_Y         MFMA                                        (src/solver.jl:1131)  bilinear_pairings_Y
//   Z   = L_X^-1 V        register forward substitution, one column per lane
//   G_X = Z_L^T Z_R       MFMA       = W^T X^-1 V                     (src/solver.jl:1117,1137-1143) bilinear_pairings_Xinv
//   A_Y[t] = G_Y[l_t, r_t]                                            (src/solver.jl:1152-1170)
//   S_j[p,q] += sum_{t1 in p, t2 in q} lam1 lam2 G_X[L1,R2] G_Y[L2,R1]  (src/solver.jl:1176-1212)
//   dense blocks: S_j[p,q] += <A_q, X^-1 A_p Y>                       (src/solver.jl:1089-1104)
//
// and finally writes S_j once, mirrored (symmetric!, src/tools.jl:43-57).  S_j is accumulated in LDS when it
// fits beside the pairing matrices, otherwise in global memory by the one thread that owns the entry.
// Deterministic: fixed block order, fixed term order, no atomics.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "clrs_wave.hip.h"

namespace clrs {

struct FBlock {
    int kind;             // 0 low rank, 1 dense
    int n;                // block side
    int URt, ULt;         // expanded unique right / left vectors (low rank)
    int sym;              // left table == right table
    int T;                // number of terms (low rank) / number of matrices cnt (dense)
    int ldn;              // LDS leading dimension of n-row buffers (>= ceil4(n), ldn % 4 == 2)
    int ldg;              // LDS leading dimension of the pairing matrices
    long long xyoff;      // offset of the block in the X/Y layout
    long long v_off;      // low rank: offset of V (n x URt, ld n) in the static arena; dense: offset of the A stack (n*n x cnt)
    long long w_off;      // low rank, !sym: offset of W (n x ULt, ld n)
    long long t0;         // first term of the block (sorted-by-p term arrays / original term order share the range)
    const int *tptr;      // low rank: [P+1] positions in the sorted term arrays; dense: [cnt] constraint index
};

struct FCluster {
    double *S;            // P x P output
    int P, b0, b1;        // blocks b0..b1-1 of the FBlock table
    int s_in_lds;         // accumulate S in LDS (1) or in global memory (0)
    // LDS offsets in doubles
    int oL, oY, oV, oTY, oZL, oGX, oGY, oS, oTab;
    int lds_doubles;      // total
};

struct FTables {
    const double *Xc, *Y;        // iterates (xy layout)
    const double *stat;          // static arena (vectors / dense matrices)
    const int *tL, *tR;          // per sorted term: left / right expanded unique index
    const double *tlam;          // per sorted term: lambda
    const int *ayL, *ayR;        // per original term: index of A_Y in G_Y
    double *AY;                  // [T] output
    unsigned long long *stamps;  // diagnostic builds only (CLRS_FUSED_STAMPS): s_memtime per phase of workgroup 0
};

#ifdef CLRS_FUSED_STAMPS
#define CLRS_STAMP(i)                                                                                           \
    do {                                                                                                        \
        if (tb.stamps && blockIdx.x == 0 && threadIdx.x == 0) {                                                 \
            unsigned long long t_;                                                                              \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                          \
            tb.stamps[(i)] = t_;                                                                                \
        }                                                                                                       \
    } while (0)
#else
#define CLRS_STAMP(i) do {} while (0)
#endif

__global__ __launch_bounds__(256) void k_cluster_assemble(const FCluster *__restrict__ clusters, const FBlock *__restrict__ blocks,
                                                          const FTables tb) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const FCluster cl = clusters[blockIdx.x];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int P = cl.P;
    double *Ls = lds + cl.oL, *Ys = lds + cl.oY, *Vs = lds + cl.oV, *TYs = lds + cl.oTY, *ZLs = lds + cl.oZL;
    double *GX = lds + cl.oGX, *GY = lds + cl.oGY, *Ss = lds + cl.oS;
    int *tab = (int *)(lds + cl.oTab);
    // S_j starts at zero, in LDS or (when it does not fit beside the pairing matrices) in global memory; an entry of
    // the global copy is touched by different threads in different phases, hence the workgroup-scope fences.
    if (cl.s_in_lds)
        for (int e = tid; e < P * P; e += 256) Ss[e] = 0.0;
    else
        for (int e = tid; e < P * P; e += 256) cl.S[e] = 0.0;

    CLRS_STAMP(0);
    for (int b = cl.b0; b < cl.b1; b++) {
        const FBlock k = blocks[b];
        const int sb = 1 + 10 * (b - cl.b0);
        (void)sb;
        const int n = k.n, ldn = k.ldn;
        const double *Lg = tb.Xc + k.xyoff, *Yg = tb.Y + k.xyoff;
        __threadfence_block();
        __syncthreads();   // previous block's accumulation has finished (LDS buffers free, global S visible)
        CLRS_STAMP(sb + 0);
        if (k.kind == 0) {
            const int UR = k.URt, UL = k.ULt, ldg = k.ldg;
            const int n4 = (n + 3) & ~3, UR16 = (UR + 15) & ~15, UL16 = (UL + 15) & ~15, n16 = (n + 15) & ~15;
            // ---- stage L (lower), Y, V [, W] zero padded; term tables ----
            // Loads in batches of eight per thread, all of a batch issued (clamped address, select after the load) before the first
            // store: a matrix is one trip to memory, not one per 256 entries (a select around a load is a branch)
            for (int e0 = 0; e0 < n16 * ldn; e0 += 4 * 256) {
                double vy[4], vl[4];
                bool in[4], lo[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int e = e0 + u * 256 + tid, i = e % ldn, j = e / ldn;
                    in[u] = e < n16 * ldn && i < n && j < n;
                    lo[u] = in[u] && i >= j;
                    const long long g = in[u] ? i + (long long)j * n : 0;
                    vy[u] = Yg[g];
                    vl[u] = Lg[g];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int e = e0 + u * 256 + tid;
                    if (e < n16 * ldn) {
                        Ys[e] = in[u] ? vy[u] : 0.0;
                        Ls[e] = lo[u] ? vl[u] : 0.0;
                    }
                }
            }
            auto stage_vectors = [&](double *dst, const double *src, int U, int U16, bool zero_ty) {
                for (int e0 = 0; e0 < U16 * ldn; e0 += 8 * 256) {
                    double vv[8];
                    bool in[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int e = e0 + u * 256 + tid, i = e % ldn, j = e / ldn;
                        in[u] = e < U16 * ldn && i < n && j < U;
                        vv[u] = src[in[u] ? i + (long long)j * n : 0];
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int e = e0 + u * 256 + tid;
                        if (e < U16 * ldn) {
                            dst[e] = in[u] ? vv[u] : 0.0;
                            if (zero_ty) TYs[e] = 0.0;   // the padding rows / columns of T_Y are read by the next contraction
                        }
                    }
                }
            };
            stage_vectors(Vs, tb.stat + k.v_off, UR, UR16, true);
            if (!k.sym) stage_vectors(ZLs, tb.stat + k.w_off, UL, UL16, false);
            int *tp = tab, *sL = tab + (P + 1), *sR = sL + k.T;
            double *slam = (double *)(tab + ((P + 1 + 2 * k.T + 1) & ~1));
            for (int e = tid; e <= P; e += 256) tp[e] = k.tptr[e] - (int)k.t0;
            for (int e = tid; e < k.T; e += 256) {
                sL[e] = tb.tL[k.t0 + e];
                sR[e] = tb.tR[k.t0 + e];
                slam[e] = tb.tlam[k.t0 + e];
            }
            __syncthreads();
            CLRS_STAMP(sb + 1);
            const double *Ws = k.sym ? Vs : ZLs;
            // ---- T_Y = Y V  (Y symmetric: Y^T V) ----
            lds_gemm_tn(Ys, ldn, Vs, ldn, TYs, ldn, n, UR, n, wave, 4, lane);
            // rows n..n4 of T_Y must be zero for the next contraction: they are (Y columns >= n are zero)
            __syncthreads();
            CLRS_STAMP(sb + 2);
            // ---- G_Y = W^T T_Y ----
            lds_gemm_tn(Ws, ldn, TYs, ldn, GY, ldg, UL, UR, n, wave, 4, lane);
            __syncthreads();
            CLRS_STAMP(sb + 3);
            // ---- Z = L^-1 V in place (and W when the tables differ); reciprocal diagonal in the T_Y buffer ----
            double *dinv = TYs;
            if (tid < n16) dinv[tid] = (tid < n) ? 1.0 / Ls[tid + tid * ldn] : 0.0;
            __syncthreads();
            lds_trsm<false>(Ls, ldn, dinv, Vs, 1, ldn, n, UR, wave, 4, lane);
            if (!k.sym) {
                __syncthreads();
                lds_trsm<false>(Ls, ldn, dinv, ZLs, 1, ldn, n, UL, wave, 4, lane);
            }
            __syncthreads();
            CLRS_STAMP(sb + 4);
            // ---- G_X = Z_L^T Z_R ----
            lds_gemm_tn(Ws, ldn, Vs, ldn, GX, ldg, UL, UR, n, wave, 4, lane);
            __syncthreads();
            CLRS_STAMP(sb + 5);
            // ---- A_Y ----
            for (int e = tid; e < k.T; e += 256) tb.AY[k.t0 + e] = GY[tb.ayL[k.t0 + e] + tb.ayR[k.t0 + e] * ldg];
            // ---- S accumulation: 16 x 16 tiles of (p, q), p <= q ----
            const int nt = (P + 15) >> 4;
            for (int tj = 0; tj < nt; tj++)
                for (int ti = 0; ti <= tj; ti++) {
                    const int p = ti * 16 + (tid & 15), q = tj * 16 + (tid >> 4);
                    if (p < P && q < P && p <= q) {
                        double acc = 0.0;
                        const int a0 = tp[p], a1 = tp[p + 1], b0 = tp[q], b1 = tp[q + 1];
                        for (int t1 = a0; t1 < a1; t1++) {
                            const int L1 = sL[t1], R1 = sR[t1];
                            const double l1 = slam[t1];
                            for (int t2 = b0; t2 < b1; t2++)
                                acc += (l1 * slam[t2]) * (GX[L1 + sR[t2] * ldg] * GY[sL[t2] + R1 * ldg]);
                        }
                        if (cl.s_in_lds) Ss[p + q * P] += acc;
                        else cl.S[p + (long long)q * P] += acc;
                    }
                }
            CLRS_STAMP(sb + 6);
        } else {
            // ---- dense block: T_a = X^-1 A_a Y for every matrix of the block, S[p_a, p_b] += <A_b, T_a> ----
            const int cnt = k.T, nn = n * n;
            const double *Ag = tb.stat + k.v_off;
            double *As = Vs, *Ts = GX;   // cnt * nn doubles each (sized by the host plan)
            for (int e = tid; e < nn; e += 256) {
                const int i = e % n, j = e / n;
                Ls[i + j * ldn] = (i >= j) ? Lg[e] : 0.0;
                Ys[i + j * ldn] = Yg[e];
            }
            for (int e = tid; e < cnt * nn; e += 256) As[e] = Ag[e];
            int *dp = tab;
            for (int e = tid; e < cnt; e += 256) dp[e] = k.tptr[e];
            __syncthreads();
            // W_a = X^-1 A_a: one thread per column (a, j): forward then backward substitution
            for (int c = tid; c < cnt * n; c += 256) {
                const double *src = As + c * n;
                double *w = Ts + c * n;
                for (int i = 0; i < n; i++) {
                    double s = src[i];
                    for (int kk = 0; kk < i; kk++) s -= Ls[i + kk * ldn] * w[kk];
                    w[i] = s / Ls[i + i * ldn];
                }
                for (int i = n - 1; i >= 0; i--) {
                    double s = w[i];
                    for (int kk = i + 1; kk < n; kk++) s -= Ls[kk + i * ldn] * w[kk];
                    w[i] = s / Ls[i + i * ldn];
                }
            }
            __syncthreads();
            // T_a = W_a Y in place row by row is not possible; use TYs as the destination stack
            double *Tt = TYs;
            for (int e = tid; e < cnt * nn; e += 256) {
                const int a = e / nn, r = e % nn, i = r % n, j = r / n;
                const double *w = Ts + a * nn;
                double s = 0.0;
                for (int kk = 0; kk < n; kk++) s += w[i + kk * n] * Ys[kk + j * ldn];
                Tt[e] = s;
            }
            __syncthreads();
            for (int e = tid; e < cnt * cnt; e += 256) {
                const int a = e % cnt, bb = e / cnt;
                if (a > bb) continue;
                const double *A2 = As + bb * nn, *T1 = Tt + a * nn;
                double s = 0.0;
                for (int kk = 0; kk < nn; kk++) s += A2[kk] * T1[kk];
                int p = dp[a], q = dp[bb];
                if (p > q) { const int t = p; p = q; q = t; }
                if (cl.s_in_lds) Ss[p + q * P] += s;
                else cl.S[p + (long long)q * P] += s;
            }
        }
    }
    __syncthreads();
    CLRS_STAMP(60);
    // ---- write S_j: full symmetric matrix ----
    if (cl.s_in_lds) {
        for (int e = tid; e < P * P; e += 256) {
            const int p = e % P, q = e / P;
            cl.S[e] = (p <= q) ? Ss[p + q * P] : Ss[q + p * P];
        }
    } else {
        __threadfence_block();
        __syncthreads();
        for (int e = tid; e < P * P; e += 256) {
            const int p = e % P, q = e / P;
            if (p > q) cl.S[e] = cl.S[q + (long long)p * P];
        }
    }
    CLRS_STAMP(61);
}


// S_j = sum of the per-block slabs written by k_cluster_assemble when the blocks of a cluster run in workgroups of their own
// (host plan: "split_blocks"), added in block order starting from zero: bit-identical to the accumulation in one workgroup.
struct SSlabSum {
    double *out;              // S_j (P x P)
    const double *slabs;      // nslabs slabs of len doubles
    long long len;
    int nslabs, pad;
};

// grid = (chunks of 1024 entries, clusters); the loads of eight slabs are in flight together, the additions stay in slab order
__global__ __launch_bounds__(256) void k_sum_S_slabs(const SSlabSum *__restrict__ descs) {
    const SSlabSum d = descs[blockIdx.y];
    const long long e0 = (long long)blockIdx.x * 1024;
    if (e0 >= d.len) return;
    long long idx[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const long long e = e0 + u * 256 + threadIdx.x;
        idx[u] = e < d.len ? e : 0;
    }
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int s0 = 0; s0 < d.nslabs; s0 += 8) {
        double v[8][4];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int sl = s0 + k < d.nslabs ? s0 + k : d.nslabs - 1;       // clamped: loaded, not added
#pragma unroll
            for (int u = 0; u < 4; u++) v[k][u] = d.slabs[idx[u] + sl * d.len];
        }
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (s0 + k < d.nslabs) {
#pragma unroll
                for (int u = 0; u < 4; u++) acc[u] += v[k][u];
            }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const long long e = e0 + u * 256 + threadIdx.x;
        if (e < d.len) d.out[e] = acc[u];
    }
}

// =====================================================================================================================
// fused factor / solve kernels for clusters with P <= 128 and N <= 128 (LDS resident; clrs_wave.hip.h does the work)
// =====================================================================================================================

// Cholesky of small matrices, one workgroup each: out = chol(in) with zeros above the diagonal.
// Used for the X blocks (approx_cholesky!(X_inv_blk, X_blk), src/solver.jl:388-399) and for Q (src/solver.jl:1274).
struct SmallPotrf {
    long long in_off, out_off;   // offsets from the base pointers given at launch
    double *dinv;                // optional: 1 / diag(L), n doubles
    int n, ldin, ldout, code;
    int nslabs;                  // > 1: the input is the sum of nslabs matrices, slab_stride apart (partial Q per cluster)
    int pad;
    long long slab_stride;
};

// Loads are issued in batches of eight 16 x 16 tiles (two tile columns x four tile rows) before any of them is stored: the
// whole matrix of the named configurations (n <= 32) is ONE round trip to memory instead of one per tile (these kernels are a
// chain of dependent steps on a few KB: every serial round trip is 0.5-2 us of a 5-15 us kernel).
__device__ __forceinline__ void lds_load_lower_identity_padded(double *A, int lda, const double *G, int ldg, int n, int n16, int tid, int nthr,
                                                               int nslabs = 1, long long slab_stride = 0) {
    // 16 x 16 element tiles: (tid & 15) walks down a column (contiguous in memory), (tid >> 4) across columns; nthr == 256
    const int i16 = tid & 15, j16 = tid >> 4, nt = n16 >> 4;
    (void)nthr;
    for (int tj0 = 0; tj0 < nt; tj0 += 2)
        for (int ti0 = 0; ti0 < nt; ti0 += 4) {
            double v[8];
            bool in[8];
            long long off[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int ti = ti0 + (e & 3), tj = tj0 + (e >> 2);
                const int i = ti * 16 + i16, j = tj * 16 + j16;
                in[e] = ti < nt && tj < nt && i < n && j < n && i >= j;
                off[e] = in[e] ? i + (long long)j * ldg : 0;
                v[e] = 0.0;
            }
            for (int sl = 0; sl < nslabs; sl += 2) {                 // fixed order: deterministic
                const bool two = sl + 1 < nslabs;
                double a[8], b[8];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    a[e] = G[off[e] + sl * slab_stride];
                    b[e] = G[off[e] + (two ? sl + 1 : sl) * slab_stride];
                }
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    v[e] += a[e];
                    if (two) v[e] += b[e];
                }
            }
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int ti = ti0 + (e & 3), tj = tj0 + (e >> 2);
                const int i = ti * 16 + i16, j = tj * 16 + j16;
                if (ti < nt && tj < nt) A[i + j * lda] = in[e] ? v[e] : ((i == j) ? 1.0 : 0.0);
            }
        }
}

// Z[i + j * ldz] = (i < rows) ? G[i + j * ldg] : 0 for i < rows16, j < cols: the same batching for a rectangular block (B_j)
__device__ __forceinline__ void lds_load_rect_padded(double *Z, int ldz, const double *G, long long ldg, int rows, int rows16, int cols, int tid) {
    const int i16 = tid & 15, j16 = tid >> 4, nti = rows16 >> 4, ntj = (cols + 15) >> 4;
    for (int tj0 = 0; tj0 < ntj; tj0 += 2)
        for (int ti0 = 0; ti0 < nti; ti0 += 4) {
            double v[8];
            bool in[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int ti = ti0 + (e & 3), tj = tj0 + (e >> 2);
                const int i = ti * 16 + i16, j = tj * 16 + j16;
                in[e] = ti < nti && tj < ntj && i < rows && j < cols;
                v[e] = G[in[e] ? i + (long long)j * ldg : 0];
            }
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int ti = ti0 + (e & 3), tj = tj0 + (e >> 2);
                const int i = ti * 16 + i16, j = tj * 16 + j16;
                if (ti < nti && tj < ntj && j < cols) Z[i + j * ldz] = in[e] ? v[e] : 0.0;
            }
        }
}

__global__ __launch_bounds__(256) void k_small_potrf(const SmallPotrf *__restrict__ descs, const double *in_base, double *out_base,
                                                     int *__restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const SmallPotrf d = descs[blockIdx.x];
    const double *din = in_base + d.in_off;
    double *dout = out_base + d.out_off;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n = d.n, n16 = (n + 15) & ~15, lda = n16 + 2;
    double *A = lds, *dinv = lds + lda * n16;
    lds_load_lower_identity_padded(A, lda, din, d.ldin, n, n16, tid, 256, d.nslabs, d.slab_stride);
    __syncthreads();
    const bool bad = lds_potrf(A, lda, dinv, n, wave, 4, lane);
    if (bad && lane == 0) atomicMin(info, d.code);
    __syncthreads();
    const int i16 = tid & 15, j16 = tid >> 4;
    for (int j0 = 0; j0 < n; j0 += 16)
        for (int i0 = 0; i0 < n; i0 += 16) {
            const int i = i0 + i16, j = j0 + j16;
            if (i < n && j < n) dout[i + (long long)j * d.ldout] = (i >= j) ? A[i + j * lda] : 0.0;
        }
    if (d.dinv && tid < n) d.dinv[tid] = dinv[tid];
}

// Per cluster: L_j = chol(S_j) in place (zero upper), LinvB_j = L_j^-1 B_j   (src/solver.jl:1245-1261)
struct CFactor {
    double *S;         // P x P, in: S_j, out: L_j
    const double *B;   // rows of the cluster in the stacked B (ld ldb)
    double *LB;        // same layout, output
    double *dinv;      // P doubles: 1 / diag(L_j), kept for the solves
    int P, N, ldb, code;
    int nc;            // columns of B processed per pass (LDS budget)
    int pad;
    double *Qslab;     // N x N partial Q_j = LinvB_j^T LinvB_j (only when nc >= N), or nullptr
};

#ifdef CLRS_W3_STAMPS
__device__ unsigned long long g_cf_stamps[16];
#define CF_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_cf_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CF_STAMP(i) do { } while (0)
#endif

__global__ __launch_bounds__(256) void k_cluster_factor(const CFactor *__restrict__ descs, int *__restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    CF_STAMP(0);
    const CFactor d = descs[blockIdx.x];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int P = d.P, P16 = (P + 15) & ~15, lda = P16 + 2;
    double *A = lds, *dinv = lds + lda * P16, *Z = dinv + P16;
    lds_load_lower_identity_padded(A, lda, d.S, P, P, P16, tid, 256);
    if (d.N > 0) lds_load_rect_padded(Z, lda, d.B, d.ldb, P, P16, min(d.nc, d.N), tid);      // the first (usually only) pass of B_j: in flight with S_j
    __syncthreads();
    CF_STAMP(1);
    const bool bad = lds_potrf(A, lda, dinv, P, wave, 4, lane);
    if (bad && lane == 0) atomicMin(info, d.code);
    __syncthreads();
    CF_STAMP(2);
    const int i16 = tid & 15, j16 = tid >> 4;
    for (int j0 = 0; j0 < P; j0 += 16)
        for (int i0 = 0; i0 < P; i0 += 16) {
            const int i = i0 + i16, j = j0 + j16;
            if (i < P && j < P) d.S[i + (long long)j * P] = (i >= j) ? A[i + j * lda] : 0.0;
        }
    if (tid < P) d.dinv[tid] = dinv[tid];
    CF_STAMP(3);
    for (int c0 = 0; c0 < d.N; c0 += d.nc) {
        const int nc = min(d.nc, d.N - c0);
        __syncthreads();
        if (c0 > 0) lds_load_rect_padded(Z, lda, d.B + (long long)c0 * d.ldb, d.ldb, P, P16, nc, tid);
        __syncthreads();
        lds_trsm<false>(A, lda, dinv, Z, 1, lda, P, nc, wave, 4, lane);
        __syncthreads();
        CF_STAMP(4);
        for (int j0 = 0; j0 < nc; j0 += 16)
            for (int i0 = 0; i0 < P; i0 += 16) {
                const int i = i0 + i16, j = j0 + j16;
                if (i < P && j < nc) d.LB[i + (long long)(c0 + j) * d.ldb] = Z[i + j * lda];
            }
        CF_STAMP(5);
        if (d.Qslab) {   // the whole LinvB_j is resident: its Gram matrix is this cluster's share of Q (src/solver.jl:1268-1269)
            const int nc16 = (nc + 15) & ~15;
            for (int e = tid; e < (nc16 - nc) * lda; e += 256) Z[nc * lda + e] = 0.0;   // zero the padding columns read by the MFMA tiles
            __syncthreads();
            lds_gemm_tn(Z, lda, Z, lda, d.Qslab, d.N, d.N, d.N, P, wave, 4, lane);
        }
    }
    CF_STAMP(6);
}

// Q = sum of the per-cluster slabs (fixed order), for the split-phase path where Q is exchanged between ranks
__global__ __launch_bounds__(256) void k_sum_slabs(const double *__restrict__ slabs, long long stride, int nslabs, long long len, double *__restrict__ out) {
    const long long e = blockIdx.x * 256ll + threadIdx.x;
    if (e >= len) return;
    double v = 0.0;
    for (int sl = 0; sl < nslabs; sl++) v += slabs[e + sl * stride];
    out[e] = v;
}

// u[k] = sum_i LB[i,k] t[i]  (approx_mul_transpose!, src/solver.jl:1546, summed over the clusters): one wave per column
__global__ __launch_bounds__(256) void k_gemv_t(const double *__restrict__ LB, int ld, int rows, int N, const double *__restrict__ t,
                                                double *__restrict__ u) {
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= N) return;
    const double *col = LB + (long long)k * ld;
    double s = 0.0;
    for (int i = lane; i < rows; i += 64) s += col[i] * t[i];
    s = wave_sum(s);
    if (lane == 0) u[k] = s;
}

// Per cluster: t_j = L_j^-1 rhs_x[j]   (src/solver.jl:1538)
struct CSolve {
    const double *L;      // P x P (ld P), lower
    const double *dinv;   // P
    const double *LB;     // rows of the cluster in the stacked LinvB (ld ldb)
    long long off;        // offset of the cluster in x-like vectors (rhs_x, t, dx)
    int P, N, ldb, pad;
};

// (batched: eight tiles in flight per trip to memory; a tile-by-tile loop with the select around the load is one trip per tile --
// 36 trips for P = 96.  The padding carries an identity diagonal, which the solves never use: their 1 / diag entries are zero there.)
__device__ __forceinline__ void lds_load_L_for_solve(double *A, int lda, double *dv, const CSolve &d, int P16, int tid) {
    lds_load_lower_identity_padded(A, lda, d.L, d.P, d.P, P16, tid, 256);
    if (tid < P16) dv[tid] = d.dinv[tid < d.P ? tid : 0] * ((tid < d.P) ? 1.0 : 0.0);
}

// partial[k] = sum_i LB[i, k] t[i] for the columns k0 .. k0 + 7 of one wave: the loads of eight columns are in flight together
__device__ __forceinline__ void gemv_t_8cols(const double *__restrict__ LB, int ldb, int rows, int N, int k0, const double *__restrict__ t, int lane,
                                             double *__restrict__ out) {
    double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int i = lane; i < rows; i += 64) {
        const double ti = t[i];
        double v[8];
#pragma unroll
        for (int c = 0; c < 8; c++) v[c] = LB[i + (long long)min(k0 + c, N - 1) * ldb];
#pragma unroll
        for (int c = 0; c < 8; c++) acc[c] = __builtin_fma(v[c], ti, acc[c]);
    }
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const double sacc = wave_sum(acc[c]);
        if (lane == 0 && k0 + c < N) out[k0 + c] = sacc;
    }
}

__global__ __launch_bounds__(256) void k_cluster_solve_fwd(const CSolve *__restrict__ descs, const double *__restrict__ rhs_x, double *__restrict__ t) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const CSolve d = descs[blockIdx.x];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int P = d.P, P16 = (P + 15) & ~15, lda = P16 + 2;
    double *A = lds, *dv = lds + lda * P16, *z = dv + P16;
    lds_load_L_for_solve(A, lda, dv, d, P16, tid);
    if (tid < P16) z[tid] = (tid < P) ? rhs_x[d.off + tid] : 0.0;
    __syncthreads();
    if (wave == 0) wave_trsv_fwd(A, lda, dv, z, P16, lane);      // one right-hand side: a single-wave job, no barriers between its panels
    __syncthreads();
    if (tid < P) t[d.off + tid] = z[tid];
}

// Per cluster: dx_j = L_j^-T (t_j + LinvB_j dy)   (src/solver.jl:1566-1573)
__global__ __launch_bounds__(256) void k_cluster_solve_bwd(const CSolve *__restrict__ descs, const double *__restrict__ dy, const double *__restrict__ t,
                                                           double *__restrict__ dx) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const CSolve d = descs[blockIdx.x];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int P = d.P, P16 = (P + 15) & ~15, lda = P16 + 2;
    double *A = lds, *dv = lds + lda * P16, *z = dv + P16;
    lds_load_L_for_solve(A, lda, dv, d, P16, tid);
    if (tid < P16) {
        double s = 0.0;
        if (tid < P) {                     // t_j + LinvB_j dy: four partial sums, eight loads in flight
            double s1 = 0.0, s2 = 0.0, s3 = 0.0;
            s = t[d.off + tid];
            int k = 0;
            for (; k + 8 <= d.N; k += 8) {
                double v[8];
#pragma unroll
                for (int c = 0; c < 8; c++) v[c] = d.LB[tid + (long long)(k + c) * d.ldb];
                s = __builtin_fma(v[0], dy[k], s); s1 = __builtin_fma(v[1], dy[k + 1], s1); s2 = __builtin_fma(v[2], dy[k + 2], s2); s3 = __builtin_fma(v[3], dy[k + 3], s3);
                s = __builtin_fma(v[4], dy[k + 4], s); s1 = __builtin_fma(v[5], dy[k + 5], s1); s2 = __builtin_fma(v[6], dy[k + 6], s2); s3 = __builtin_fma(v[7], dy[k + 7], s3);
            }
            for (; k < d.N; k++) s = __builtin_fma(d.LB[tid + (long long)k * d.ldb], dy[k], s);
            s = (s + s1) + (s2 + s3);
        }
        z[tid] = s;
    }
    __syncthreads();
    if (wave == 0) wave_trsv_bwd(A, lda, dv, z, P16, lane);
    __syncthreads();
    if (tid < P) dx[d.off + tid] = z[tid];
}

// dy = Q^-1 (rhs_y - u) with Q = L_Q L_Q^T   (src/solver.jl:1550-1558)
// With LB != nullptr the kernel first forms u = LinvB^T t itself (single-GPU path: saves the k_gemv_t launch).
__global__ __launch_bounds__(256) void k_q_solve(const double *__restrict__ LQ, const double *__restrict__ dinvQ, int N, const double *__restrict__ rhs_y,
                                                 double *__restrict__ u, double *__restrict__ dy, const double *__restrict__ LB, int ldb, int rows,
                                                 const double *__restrict__ t) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int N16 = (N + 15) & ~15, lda = N16 + 2;
    double *A = lds, *dv = lds + lda * N16, *z = dv + N16;
    lds_load_lower_identity_padded(A, lda, LQ, N, N, N16, tid, 256);
    if (LB) {
        for (int k0 = wave * 8; k0 < N; k0 += 32) gemv_t_8cols(LB, ldb, rows, N, k0, t, lane, u);
        __threadfence_block();
        __syncthreads();
    }
    if (tid < N16) {
        dv[tid] = (tid < N) ? dinvQ[tid] : 0.0;
        z[tid] = (tid < N) ? rhs_y[tid] - u[tid] : 0.0;
    }
    __syncthreads();
    if (wave == 0) {
        wave_trsv_fwd(A, lda, dv, z, N16, lane);
        wave_trsv_bwd(A, lda, dv, z, N16, lane);
    }
    __syncthreads();
    if (tid < N) dy[tid] = z[tid];
}

}  // namespace clrs

namespace clrs {

// =====================================================================================================================
// k_dense_block: the dense ("high rank") branch of the assembly for one PSD block with n <= 64, LDS resident
// =====================================================================================================================
// Sd[x,y] = <A_x, X^-1 A_y Y> (src/solver.jl:1089-1104) for the cnt constraint matrices of the block:
//   U_x = A_x Y (MFMA), W_y = X^-1 A_y (blocked DPP/MFMA triangular solves on all matrices at once, in place),
//   Sd = <U_x, W_y> as one Gram contraction over the n^2 entries (MFMA, K split over the 4 waves).
// One workgroup per block; every MOI/JuMP problem and the SDPA import land here (all their matrices are dense).
struct DBlock {
    int n, cnt;
    long long xyoff;     // block in the X/Y layout
    long long a_off;     // stack of cnt n x n matrices in the static arena
    double *Sd;          // cnt x cnt output (column-major, ld cnt)
};

__global__ __launch_bounds__(256) void k_dense_block(const DBlock *__restrict__ blocks, const FTables tb) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const DBlock k = blocks[blockIdx.x];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = k.n, n16 = (n + 15) & ~15, lda = n16 + 2, cnt = k.cnt;
    const int msz = lda * n16;                           // one matrix (zero padded columns up to n16)
    double *Ls = lds, *Ys = Ls + msz, *dinv = Ys + msz, *Ws = dinv + n16, *Us = Ws + cnt * msz, *part = Us + cnt * msz;   // part: 4 x 256
    const int i16 = tid & 15, j16 = tid >> 4;
    for (int j0 = 0; j0 < n16; j0 += 16)
        for (int i0 = 0; i0 < n16; i0 += 16) {
            const int i = i0 + i16, j = j0 + j16;
            const bool in = i < n && j < n;
            Ls[i + j * lda] = (in && i >= j) ? tb.Xc[k.xyoff + i + (long long)j * n] : 0.0;
            Ys[i + j * lda] = in ? tb.Y[k.xyoff + i + (long long)j * n] : 0.0;
        }
    if (tid < n16) dinv[tid] = (tid < n) ? 1.0 / tb.Xc[k.xyoff + tid + (long long)tid * n] : 0.0;
    const double *Ag = tb.stat + k.a_off;
    for (int a = 0; a < cnt; a++)
        for (int j0 = 0; j0 < n16; j0 += 16)
            for (int i0 = 0; i0 < n16; i0 += 16) {
                const int i = i0 + i16, j = j0 + j16;
                Ws[a * msz + i + j * lda] = (i < n && j < n) ? Ag[(long long)a * n * n + i + (long long)j * n] : 0.0;
            }
    // the two padding rows of every column (rows n16, n16+1 of the leading dimension) take part in the Gram contraction: zero them
    for (int e = tid; e < cnt * n16; e += 256) {
        Ws[e * lda + n16] = 0.0; Ws[e * lda + n16 + 1] = 0.0;
        Us[e * lda + n16] = 0.0; Us[e * lda + n16 + 1] = 0.0;
    }
    __syncthreads();
    // U_a = A_a Y  (A_a symmetric: A_a^T Y)
    for (int a = 0; a < cnt; a++) lds_gemm_tn(Ws + a * msz, lda, Ys, lda, Us + a * msz, lda, n16, n16, n, wave, 4, lane);
    __syncthreads();
    // W = X^-1 [A_1 ... A_cnt]: all columns at once
    lds_trsm<false>(Ls, lda, dinv, Ws, 1, lda, n, cnt * n16, wave, 4, lane);
    __syncthreads();
    lds_trsm<true>(Ls, lda, dinv, Ws, 1, lda, n, cnt * n16, wave, 4, lane);
    __syncthreads();
    // Sd[x,y] = sum_e U_x[e] W_y[e]: tiles of 16 x 16 outputs, the contraction (msz long) split over the 4 waves
    const int ct = (cnt + 15) >> 4, l15 = lane & 15, l4 = lane >> 4;
    const int kq = ((msz / 4 + 3) / 4) * 4;               // k range per wave, multiple of 4
    for (int ty = 0; ty < ct; ty++)
        for (int tx = 0; tx < ct; tx++) {
            v4d_f acc = {0.0, 0.0, 0.0, 0.0};
            const int k0 = wave * kq, k1 = min(msz, k0 + kq);
            const int xa = min(tx * 16 + l15, cnt - 1), ya = min(ty * 16 + l15, cnt - 1);     // clamped: out-of-range columns are discarded below
            const double *ux = Us + xa * msz + l4, *wy = Ws + ya * msz + l4;
            for (int kk = k0; kk < k1; kk += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wy[kk], ux[kk], acc, 0, 0, 0);   // D[r][c]: c -> x, r -> y
            __syncthreads();
#pragma unroll
            for (int reg = 0; reg < 4; reg++) part[wave * 256 + l15 + 16 * (l4 + 4 * reg)] = acc[reg];
            __syncthreads();
            const int x = tx * 16 + (tid & 15), y = ty * 16 + (tid >> 4);
            if (x < cnt && y < cnt) k.Sd[x + (long long)y * cnt] = (part[tid] + part[256 + tid]) + (part[512 + tid] + part[768 + tid]);
        }
}

// Whole solve stage in ONE workgroup, for problems with a handful of small clusters (the named configurations): the three
// phases of src/solver.jl:1537-1573 are separated by workgroup barriers instead of kernel boundaries, t and u never leave LDS.
__global__ __launch_bounds__(256) void k_solve_small(const CSolve *__restrict__ descs, int J, const double *__restrict__ LQ, const double *__restrict__ dinvQ,
                                                     int N, int xlen, const double *__restrict__ rhs_x, const double *__restrict__ rhs_y,
                                                     const double *__restrict__ LBall, double *__restrict__ dx, double *__restrict__ dy, int maxP16) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lda = maxP16 + 2, xl16 = (xlen + 15) & ~15, N16 = (N + 15) & ~15;
    double *A = lds, *dv = A + lda * maxP16, *z = dv + maxP16, *tt = z + maxP16 + 2, *uu = tt + xl16, *yy = uu + N16;
    const int i16 = tid & 15, j16 = tid >> 4;
    // ---- t_j = L_j^-1 rhs_x[j] ----
    for (int j = 0; j < J; j++) {
        const CSolve d = descs[j];
        const int P16 = (d.P + 15) & ~15;
        __syncthreads();
        lds_load_L_for_solve(A, lda, dv, d, P16, tid);
        if (tid < P16) z[tid] = (tid < d.P) ? rhs_x[d.off + tid] : 0.0;
        __syncthreads();
        lds_trsm<false>(A, lda, dv, z, 1, lda, d.P, 1, wave, 4, lane);
        __syncthreads();
        if (tid < d.P) tt[d.off + tid] = z[tid];
    }
    __syncthreads();
    if (N > 0) {
        // ---- u = LinvB^T t, dy = Q^-1 (rhs_y - u) ----
        for (int k = wave; k < N; k += 4) {
            const double *col = LBall + (long long)k * xlen;
            double sacc = 0.0;
            for (int i = lane; i < xlen; i += 64) sacc += col[i] * tt[i];
            sacc = wave_sum(sacc);
            if (lane == 0) uu[k] = sacc;
        }
        for (int j0 = 0; j0 < N16; j0 += 16)
            for (int i0 = 0; i0 < N16; i0 += 16) {
                const int i = i0 + i16, j = j0 + j16;
                A[i + j * lda] = (i < N && j < N && i >= j) ? LQ[i + (long long)j * N] : 0.0;
            }
        if (tid < N16) dv[tid] = (tid < N) ? dinvQ[tid] : 0.0;
        __syncthreads();
        if (tid < N16) yy[tid] = (tid < N) ? rhs_y[tid] - uu[tid] : 0.0;
        __syncthreads();
        lds_trsm<false>(A, lda, dv, yy, 1, lda, N, 1, wave, 4, lane);
        __syncthreads();
        lds_trsm<true>(A, lda, dv, yy, 1, lda, N, 1, wave, 4, lane);
        __syncthreads();
        if (tid < N) dy[tid] = yy[tid];
    }
    // ---- dx_j = L_j^-T (t_j + LinvB_j dy) ----
    for (int j = 0; j < J; j++) {
        const CSolve d = descs[j];
        const int P16 = (d.P + 15) & ~15;
        __syncthreads();
        lds_load_L_for_solve(A, lda, dv, d, P16, tid);
        if (tid < P16) {
            double sacc = 0.0;
            if (tid < d.P) {
                sacc = tt[d.off + tid];
                for (int k = 0; k < N; k++) sacc += d.LB[tid + (long long)k * d.ldb] * yy[k];
            }
            z[tid] = sacc;
        }
        __syncthreads();
        lds_trsm<true>(A, lda, dv, z, 1, lda, d.P, 1, wave, 4, lane);
        __syncthreads();
        if (tid < d.P) dx[d.off + tid] = z[tid];
    }
}

// =====================================================================================================================
// k_cluster_assemble_w1: Schur assembly with ONE WAVE PER PSD BLOCK (no workgroup barriers inside a block)
// =====================================================================================================================
// For the dominant block shape of the named configurations -- n <= 16, rank-1 terms, one term per constraint, left and
// right vectors identical and all distinct (Delsarte, PolyOpt with 2d <= 30, Cohn-Elkies) -- the pairing matrices never
// need to exist in memory: with U = number of vectors of the block and pmap[u] the constraint of vector u,
//
//     S[pmap[u], pmap[v]] += lam_u lam_v * (Z^T Z)[u,v] * (V^T Y V)[u,v]          Z = L_X^-1 V
//
// is a Hadamard product of two MFMA output tiles that live in the same lanes.  A wave stages V (n x U) in LDS, keeps
// Y and the row of L_X it needs in registers, computes T_Y = Y V, the lower tiles of G_Y = V^T T_Y (registers),
// Z (DPP forward substitution in place of V), then tile by tile G_X = Z^T Z and the product, which it adds into its
// own P x P slab in LDS.  The waves of a workgroup take the blocks of one cluster round-robin (dense blocks included);
// one barrier, then the slabs are summed in a fixed order into S_j (symmetric by construction: only u >= v is
// computed and mirrored).  Independent waves overlap each other's global-memory latency, so many clusters per launch
// stream at HBM rate instead of paying every round trip in lock step.
struct WBlock {
    int kind;            // 0: simple low-rank block, 1: dense block
    int n, U;            // side; number of vectors (= terms) / dense: number of matrices
    int pad;
    long long xyoff;     // offset of the block in the X/Y layout
    long long v_off;     // static arena: vectors n x U (ld n) / dense: stack of n x n matrices
    const int *pmap;     // [U] constraint index of vector u / dense: constraint index of matrix a
    const double *lam;   // [U] lambda of the term of vector u
    const int *ay;       // [U] position of the term in the A_Y output
};
struct WCluster {
    double *S;
    int P, nblk;         // blocks of the cluster: slots 0..nblk-1 of its row of the block table
    int work_doubles;    // per-wave work area (after the slab)
    int pad;
};

template <int UT>   // U <= 16 * UT
__global__ __launch_bounds__(512) void k_cluster_assemble_w1(const WCluster *__restrict__ clusters, const WBlock *__restrict__ blocks, const FTables tb,
                                                             int nwaves, int maxblk) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
    // The block table has a fixed number of slots per cluster, so the descriptor of this wave's first block is fetched
    // together with the cluster descriptor (one round trip to memory instead of two before the first data load).
    const WBlock *myblocks = blocks + (size_t)blockIdx.x * maxblk;
    WBlock k = myblocks[wave < maxblk ? wave : 0];
    const WCluster cl = clusters[blockIdx.x];
    const int P = cl.P, PS = P | 1, PP = P * PS;   // slab leading dimension odd: the mirrored (strided) accesses spread over the banks
    constexpr int LD = 17;                       // leading dimension of the 16-row LDS buffers (one 2-way bank conflict per operand read)
    double *slab = lds + (size_t)wave * (PP + cl.work_doubles);
    double *work = slab + PP;
    for (int e = lane; e < PP; e += 64) slab[e] = 0.0;
    CLRS_STAMP(0);

    for (int b = wave; b < cl.nblk; b += nwaves) {
        wave_sync();
        if (b != wave) k = myblocks[b];
        const int sb = 1 + 10 * b;
        (void)sb;
        CLRS_STAMP(sb + 0);
        const int n = k.n;
        const double *Lg = tb.Xc + k.xyoff, *Yg = tb.Y + k.xyoff;
        if (k.kind == 0) {
            const int U = k.U;
            double *Vs = work, *TYs = work + LD * 16 * UT;
            const double *Vg = tb.stat + k.v_off;
            // ---- every global load of the block is issued here, before any use ----
            // (each element of L, Y, V and of the per-vector tables is fetched by exactly one lane; L and the tables are
            // redistributed through LDS, which costs LDS latency once instead of 40 more global load instructions per lane)
            double yop[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int kk = 4 * q + l4;
                yop[q] = (kk < n && l15 < n) ? Yg[kk + l15 * n] : 0.0;          // MFMA operand Y[k, i]: k = 4q + (lane >> 4), i = lane & 15
            }
            double *Lt = TYs;                                                   // 16 x 16 staging of L_X (the T_Y buffer is free until T_Y is formed)
            double *tl = TYs + LD * 16;                                         // lam[u], then pmap / ay as ints
            int *tpm = (int *)(tl + 16 * UT), *tay = tpm + 16 * UT;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int col = 4 * q + l4;
                Lt[l15 + col * LD] = (l15 < n && col < l15) ? Lg[l15 + col * n] : 0.0;            // strictly lower part
            }
            const double dg = (l15 < n) ? Lg[l15 * (n + 1)] : 1.0;
            for (int u = lane; u < 16 * UT; u += 64) {
                tl[u] = (u < U) ? k.lam[u] : 0.0;
                tpm[u] = (u < U) ? k.pmap[u] : 0;
                tay[u] = (u < U) ? k.ay[u] : 0;
            }
#pragma unroll
            for (int c0 = 0; c0 < 16 * UT; c0 += 4) {
                const int col = c0 + l4;
                Vs[l15 + col * LD] = (l15 < n && col < U) ? Vg[l15 + col * n] : 0.0;
            }
            const double di = (l15 < n) ? 1.0 / dg : 0.0;
            wave_sync();
            double Lr[16], lam_r[UT], lam_c[UT * 4];
            int pm_r[UT], pm_c[UT * 4], ay_r[UT];
#pragma unroll
            for (int c = 0; c < 16; c++) Lr[c] = Lt[l15 + c * LD];              // row (lane & 15) of L_X, zero on and above the diagonal
#pragma unroll
            for (int t = 0; t < UT; t++) {
                const int u = t * 16 + l15;                                     // row index of the tiles in tile row t
                lam_r[t] = tl[u];
                pm_r[t] = tpm[u];
                ay_r[t] = tay[u];
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int v = t * 16 + l4 + 4 * reg;                        // column index of the tiles in tile column t
                    lam_c[t * 4 + reg] = tl[v];
                    pm_c[t * 4 + reg] = tpm[v];
                }
            }
            wave_sync();
            CLRS_STAMP(sb + 1);
            // ---- T_Y = Y V ----
#pragma unroll
            for (int tj = 0; tj < UT; tj++) {
                if (tj * 16 < U) {
                    v4d_f acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Vs[(4 * q + l4) + (tj * 16 + l15) * LD], yop[q], acc, 0, 0, 0);
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) TYs[l15 + (tj * 16 + l4 + 4 * reg) * LD] = acc[reg];   // T_Y[i, j]
                }
            }
            wave_sync();
            CLRS_STAMP(sb + 2);
            // ---- lower tiles of G_Y = V^T T_Y, kept in registers: lane holds G_Y[ti*16 + (lane&15), tj*16 + (lane>>4) + 4 reg] ----
            v4d_f gy[UT * (UT + 1) / 2];
#pragma unroll
            for (int ti = 0; ti < UT; ti++)
#pragma unroll
                for (int tj = 0; tj <= ti; tj++) {
                    v4d_f acc = {0.0, 0.0, 0.0, 0.0};
                    if (ti * 16 < U) {
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(TYs[(4 * q + l4) + (tj * 16 + l15) * LD], Vs[(4 * q + l4) + (ti * 16 + l15) * LD], acc, 0, 0, 0);
                    }
                    gy[ti * (ti + 1) / 2 + tj] = acc;
                }
            // A_Y[t] = w^T Y v of the term's own vector: the diagonal of G_Y
#pragma unroll
            for (int t = 0; t < UT; t++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int u = t * 16 + l15;
                    if (l15 == l4 + 4 * reg && u < U) tb.AY[ay_r[t]] = gy[t * (t + 1) / 2 + t][reg];
                }
            wave_sync();
            CLRS_STAMP(sb + 3);
            // ---- Z = L_X^-1 V in place: 4 columns per 16-lane group pair, two groups interleaved ----
#pragma unroll
            for (int c0 = 0; c0 < 16 * UT; c0 += 8) {
                if (c0 < U) {
                    double x0 = Vs[l15 + (c0 + l4) * LD], x1 = Vs[l15 + (c0 + 4 + l4) * LD];
                    Trsm16<0>::run(x0, x1, Lr, di);
                    Vs[l15 + (c0 + l4) * LD] = x0 * di;
                    Vs[l15 + (c0 + 4 + l4) * LD] = x1 * di;
                }
            }
            wave_sync();
            CLRS_STAMP(sb + 4);
            // ---- G_X = Z^T Z for all lower tiles (independent MFMA chains back to back), Hadamard with G_Y, scale ----
#pragma unroll
            for (int ti = 0; ti < UT; ti++)
#pragma unroll
                for (int tj = 0; tj <= ti; tj++) {
                    v4d_f acc = {0.0, 0.0, 0.0, 0.0};
                    if (ti * 16 < U) {
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Vs[(4 * q + l4) + (tj * 16 + l15) * LD], Vs[(4 * q + l4) + (ti * 16 + l15) * LD], acc, 0, 0, 0);
                    }
                    v4d_f &g = gy[ti * (ti + 1) / 2 + tj];
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) g[reg] = (lam_r[ti] * lam_c[tj * 4 + reg]) * (acc[reg] * g[reg]);
                }
            // ---- add into the slab (u >= v only, mirrored).  The (p, q) of one block are all distinct: per tile the 8 slab
            // entries are read first and written afterwards, so the LDS latency is paid once per tile ----
#pragma unroll
            for (int ti = 0; ti < UT; ti++)
#pragma unroll
                for (int tj = 0; tj <= ti; tj++) {
                    if (ti * 16 < U) {
                        const v4d_f sv = gy[ti * (ti + 1) / 2 + tj];
                        double o1[4], o2[4];
                        bool on[4];
#pragma unroll
                        for (int reg = 0; reg < 4; reg++) {
                            const int u = ti * 16 + l15, v = tj * 16 + l4 + 4 * reg;
                            on[reg] = u < U && v < U && u >= v;
                            const int p = pm_r[ti], q2 = pm_c[tj * 4 + reg];
                            o1[reg] = slab[p + q2 * PS];        // always a valid address (p = q2 = 0 for padding lanes): unconditional reads
                            o2[reg] = slab[q2 + p * PS];
                        }
#pragma unroll
                        for (int reg = 0; reg < 4; reg++) {
                            const int u = ti * 16 + l15, v = tj * 16 + l4 + 4 * reg;
                            const int p = pm_r[ti], q2 = pm_c[tj * 4 + reg];
                            if (on[reg]) {
                                slab[p + q2 * PS] = o1[reg] + sv[reg];
                                if (u != v) slab[q2 + p * PS] = o2[reg] + sv[reg];
                            }
                        }
                    }
                }
            CLRS_STAMP(sb + 5);
        } else {
            // ---- dense block (n <= 16): T_a = X^-1 A_a Y for every matrix, S[p_a, p_b] += <A_b, T_a>, all within the wave ----
            const int cnt = k.U, nn = n * n;
            const double *Ag = tb.stat + k.v_off;
            double *Ls = work, *Ys = Ls + nn, *As = Ys + nn, *Ws = As + cnt * nn, *Tt = Ws + cnt * nn;
            for (int e = lane; e < nn; e += 64) {
                const int i = e % n, j = e / n;
                Ls[e] = (i >= j) ? Lg[e] : 0.0;
                Ys[e] = Yg[e];
            }
            for (int e = lane; e < cnt * nn; e += 64) As[e] = Ag[e];
            wave_sync();
            for (int c = lane; c < cnt * n; c += 64) {      // W_a = X^-1 A_a, one column per lane
                const double *src = As + c * n;
                double *w = Ws + c * n;
                for (int i = 0; i < n; i++) {
                    double s = src[i];
                    for (int kk = 0; kk < i; kk++) s -= Ls[i + kk * n] * w[kk];
                    w[i] = s / Ls[i + i * n];
                }
                for (int i = n - 1; i >= 0; i--) {
                    double s = w[i];
                    for (int kk = i + 1; kk < n; kk++) s -= Ls[kk + i * n] * w[kk];
                    w[i] = s / Ls[i + i * n];
                }
            }
            wave_sync();
            for (int e = lane; e < cnt * nn; e += 64) {     // T_a = W_a Y
                const int a = e / nn, r = e % nn, i = r % n, j = r / n;
                const double *w = Ws + a * nn;
                double s = 0.0;
                for (int kk = 0; kk < n; kk++) s += w[i + kk * n] * Ys[kk + j * n];
                Tt[e] = s;
            }
            wave_sync();
            for (int e = lane; e < cnt * cnt; e += 64) {
                const int a = e % cnt, bb = e / cnt;
                if (a > bb) continue;
                const double *A2 = As + bb * nn, *T1 = Tt + a * nn;
                double s = 0.0;
                for (int kk = 0; kk < nn; kk++) s += A2[kk] * T1[kk];
                const int p = k.pmap[a], q2 = k.pmap[bb];
                slab[p + q2 * PS] += s;
                if (p != q2) slab[q2 + p * PS] += s;
            }
        }
    }
    __syncthreads();
    CLRS_STAMP(60);
    // ---- S_j = sum of the wave slabs, fixed order ----
    const int stride = PP + cl.work_doubles;
    const int i16 = threadIdx.x & 15, j16 = threadIdx.x >> 4, jstep = blockDim.x >> 4;
    for (int j0 = 0; j0 < P; j0 += jstep)
        for (int i0 = 0; i0 < P; i0 += 64) {                         // 4 independent sums per thread per pass
            double s[4] = {0.0, 0.0, 0.0, 0.0};
            const int j = j0 + j16;
            for (int w = 0; w < nwaves; w++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int i = i0 + 16 * q + i16;
                    if (i < P && j < P) s[q] += lds[w * stride + i + j * PS];
                }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = i0 + 16 * q + i16;
                if (i < P && j < P) cl.S[i + (long long)j * P] = s[q];
            }
        }
    CLRS_STAMP(61);
}

}  // namespace clrs

namespace clrs {

// =====================================================================================================================
// k_cluster_assemble_w2: ONE WAVE PER CLUSTER, S accumulated in registers
// =====================================================================================================================
// When every low-rank block of a cluster is "simple" (see k_cluster_assemble_w1) AND they all use the same constraint
// order (vector u of every block belongs to constraint pmap[u], U = P) AND the dense blocks are 1 x 1 -- the Cohn-Elkies,
// Delsarte and univariate polynomial-optimisation clusters -- the Hadamard products of all blocks land on the same lanes:
// the S tiles stay in registers across the blocks, there is no slab, no barrier and no reduction, and LDS holds only V and
// T_Y of the block in flight (8.7 KB per wave for U <= 32), so 12-16 waves per CU overlap each other's memory latency.
// Z = L^-1 V is formed as (L^-1) V: the 16 x 16 inverse by the DPP substitution on the identity (two passes instead of
// U/8), then MFMA.  S_j is staged through the (then free) LDS area for a coalesced store when it fits.
struct W2Block {
    int kind;            // 0: simple low-rank block, 1: dense 1 x 1 block
    int n, pad0, pad1;
    long long xyoff;     // offset of the block in the X/Y layout
    long long v_off;     // static arena: vectors n x U (ld n)
    const double *lam;   // [U] lambda of the term of vector u / dense: the 1 x 1 matrix entry of constraint pmap[u] (0 when absent)
    const int *ay;       // [U] position of the term in the A_Y output (low rank only)
};
struct W2Cluster {
    double *S;
    const int *pmap;     // [U] constraint index of vector u, a permutation of 0..P-1
    int P, nblk;
    long long blk0;      // first block of the cluster in the block table
};

// FULL: every low-rank block has n == 16 and U == 16 * UT exactly (Cohn-Elkies 2d = 30: n = 16, U = 32): no tile is partial, so
// every bounds predicate folds away at compile time -- a third of the VALU / SALU instructions of the issue-bound loop.
template <int UT, bool FULL>   // P = U <= 16 * UT
__global__ __launch_bounds__(256) void k_cluster_assemble_w2(const W2Cluster *__restrict__ clusters, const W2Block *__restrict__ blocks, const FTables tb,
                                                             int nclusters) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
    const int ci = blockIdx.x * 4 + wave;
    if (ci >= nclusters) return;                 // no workgroup barrier anywhere in this kernel
    const W2Cluster cl = clusters[ci];
    constexpr int LD = 17, WORK = 2 * LD * 16 * UT;
    double *work = lds + (size_t)wave * WORK;
    double *Vs = work, *TYs = work + LD * 16 * UT;
    const int P = FULL ? 16 * UT : cl.P, U = P;
    constexpr int NT = UT * (UT + 1) / 2;
    v4d_f sacc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) sacc[t] = (v4d_f){0.0, 0.0, 0.0, 0.0};

    // Software pipeline over the blocks of the cluster: the global loads of block b+1 are issued (into registers) before the
    // arithmetic of block b and written to LDS when block b is done, so that the memory time of one block hides behind the
    // arithmetic of the other instead of every wave of the chip loading, then computing, in lock step.
    struct Pre {
        double yop[4], lt[4], dg, lam, v[4 * UT];
        int ay;
    };
    auto issue = [&](const W2Block &kb, Pre &r) {
        const int n = FULL ? 16 : kb.n;
        const double *Lg = tb.Xc + kb.xyoff, *Yg = tb.Y + kb.xyoff, *Vg = tb.stat + kb.v_off;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int kk = 4 * q + l4;
            r.yop[q] = (kk < n && l15 < n) ? Yg[kk + l15 * n] : 0.0;        // MFMA operand Y[k, i]
            r.lt[q] = (l15 < n && kk < l15) ? Lg[l15 + kk * n] : 0.0;       // strictly lower part of L_X
        }
        r.dg = (l15 < n) ? Lg[l15 * (n + 1)] : 1.0;
        r.lam = (lane < U && lane < 16 * UT) ? kb.lam[lane] : 0.0;
        r.ay = (lane < U && lane < 16 * UT) ? kb.ay[lane] : 0;
#pragma unroll
        for (int c = 0; c < 4 * UT; c++) {
            const int col = 4 * c + l4;
            r.v[c] = (l15 < n && col < U) ? Vg[l15 + col * n] : 0.0;
        }
    };
    static_assert(UT <= 4, "the per-vector tables are fetched by one lane each");
    W2Block k = blocks[cl.blk0];
    Pre cur, nxt;
    if (k.kind == 0) issue(k, cur);
    for (int b = 0; b < cl.nblk; b++) {
        wave_sync();
        W2Block knext = k;
        const bool have_next = b + 1 < cl.nblk;
        if (have_next) knext = blocks[cl.blk0 + b + 1];
        const double *Lg = tb.Xc + k.xyoff, *Yg = tb.Y + k.xyoff;
        double lam_r[UT], lam_c[UT * 4];
        if (k.kind == 1) {
            // 1 x 1 dense block: S[p_u, p_v] += a_u a_v Y / X, X = L^2
            const double Lx = Lg[0], ratio = Yg[0] / (Lx * Lx);
            if (have_next && knext.kind == 0) issue(knext, cur);
#pragma unroll
            for (int t = 0; t < UT; t++) {
                const int u = t * 16 + l15;
                lam_r[t] = (u < U) ? k.lam[u] : 0.0;
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int v = t * 16 + l4 + 4 * reg;
                    lam_c[t * 4 + reg] = (v < U) ? k.lam[v] : 0.0;
                }
            }
#pragma unroll
            for (int ti = 0; ti < UT; ti++)
#pragma unroll
                for (int tj = 0; tj <= ti; tj++)
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) sacc[ti * (ti + 1) / 2 + tj][reg] += (lam_r[ti] * lam_c[tj * 4 + reg]) * ratio;
            k = knext;
            continue;
        }
        const int n = FULL ? 16 : k.n;
        // ---- the block's data is in `cur` (registers): commit to LDS, then start the loads of the next block ----
        double yop[4];
        double *Lt = TYs;                       // strictly lower part of L_X, staged for the row reads
        double *tl = TYs + LD * 16;
        int *tay = (int *)(tl + 16 * UT);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            yop[q] = cur.yop[q];
            Lt[l15 + (4 * q + l4) * LD] = cur.lt[q];
        }
        if (lane < 16 * UT) { tl[lane] = cur.lam; tay[lane] = cur.ay; }
#pragma unroll
        for (int c = 0; c < 4 * UT; c++) Vs[l15 + (4 * c + l4) * LD] = cur.v[c];
        const double di = (l15 < n) ? 1.0 / cur.dg : 0.0;
        if (have_next && knext.kind == 0) issue(knext, nxt);
        wave_sync();
        double Lr[16];
        int ay_r[UT];
#pragma unroll
        for (int c = 0; c < 16; c++) Lr[c] = Lt[l15 + c * LD];
#pragma unroll
        for (int t = 0; t < UT; t++) {
            const int u = t * 16 + l15;
            lam_r[t] = tl[u];
            ay_r[t] = tay[u];
#pragma unroll
            for (int reg = 0; reg < 4; reg++) lam_c[t * 4 + reg] = tl[t * 16 + l4 + 4 * reg];
        }
        wave_sync();
        // ---- T_Y = Y V ----
#pragma unroll
        for (int tj = 0; tj < UT; tj++) {
            v4d_f acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Vs[(4 * q + l4) + (tj * 16 + l15) * LD], yop[q], acc, 0, 0, 0);
#pragma unroll
            for (int reg = 0; reg < 4; reg++) TYs[l15 + (tj * 16 + l4 + 4 * reg) * LD] = acc[reg];
        }
        wave_sync();
        // ---- lower tiles of G_Y = V^T T_Y in registers ----
        v4d_f gy[NT];
#pragma unroll
        for (int ti = 0; ti < UT; ti++)
#pragma unroll
            for (int tj = 0; tj <= ti; tj++) {
                v4d_f acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 4; q++)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(TYs[(4 * q + l4) + (tj * 16 + l15) * LD], Vs[(4 * q + l4) + (ti * 16 + l15) * LD], acc, 0, 0, 0);
                gy[ti * (ti + 1) / 2 + tj] = acc;
            }
#pragma unroll
        for (int t = 0; t < UT; t++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int u = t * 16 + l15;
                if (l15 == l4 + 4 * reg && u < U) tb.AY[ay_r[t]] = gy[t * (t + 1) / 2 + t][reg];
            }
        wave_sync();
        // ---- W = L^-1 by substitution on the identity (16 columns = 4 groups = 2 passes), stored TRANSPOSED over T_Y ----
        double *Wt = TYs;                        // Wt[k + i * LD] = W[i, k]
        {
            double x0 = (l15 == l4) ? 1.0 : 0.0, x1 = (l15 == 4 + l4) ? 1.0 : 0.0;       // columns l4 and 4 + l4 of I
            Trsm16<0>::run(x0, x1, Lr, di);
            Wt[l4 + l15 * LD] = x0 * di;
            Wt[(4 + l4) + l15 * LD] = x1 * di;
            x0 = (l15 == 8 + l4) ? 1.0 : 0.0;
            x1 = (l15 == 12 + l4) ? 1.0 : 0.0;
            Trsm16<0>::run(x0, x1, Lr, di);
            Wt[(8 + l4) + l15 * LD] = x0 * di;
            Wt[(12 + l4) + l15 * LD] = x1 * di;
        }
        wave_sync();
        // ---- Z = W V (MFMA), in registers first (V is both source and destination) ----
        v4d_f zt[UT];
#pragma unroll
        for (int tj = 0; tj < UT; tj++) {
            v4d_f acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; q++)   // Z[i, j] = sum_k Wt[k, i] V[k, j]
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Vs[(4 * q + l4) + (tj * 16 + l15) * LD], Wt[(4 * q + l4) + l15 * LD], acc, 0, 0, 0);
            zt[tj] = acc;
        }
        wave_sync();
#pragma unroll
        for (int tj = 0; tj < UT; tj++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) Vs[l15 + (tj * 16 + l4 + 4 * reg) * LD] = zt[tj][reg];
        wave_sync();
        // ---- G_X = Z^T Z tile by tile, Hadamard with G_Y, scale, accumulate in registers ----
#pragma unroll
        for (int ti = 0; ti < UT; ti++)
#pragma unroll
            for (int tj = 0; tj <= ti; tj++) {
                v4d_f acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 4; q++)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Vs[(4 * q + l4) + (tj * 16 + l15) * LD], Vs[(4 * q + l4) + (ti * 16 + l15) * LD], acc, 0, 0, 0);
                const v4d_f g = gy[ti * (ti + 1) / 2 + tj];
#pragma unroll
                for (int reg = 0; reg < 4; reg++) sacc[ti * (ti + 1) / 2 + tj][reg] += (lam_r[ti] * lam_c[tj * 4 + reg]) * (acc[reg] * g[reg]);
            }
        cur = nxt;
        k = knext;
    }
    wave_sync();
    // ---- S_j: u >= v computed, mirrored.  Through LDS for coalesced stores when P (P|1) fits in the work area. ----
    int pm_r[UT], pm_c[UT * 4];
#pragma unroll
    for (int t = 0; t < UT; t++) {
        const int u = t * 16 + l15;
        pm_r[t] = (u < U) ? cl.pmap[u] : 0;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int v = t * 16 + l4 + 4 * reg;
            pm_c[t * 4 + reg] = (v < U) ? cl.pmap[v] : 0;
        }
    }
    const int PS = P | 1;
    const bool via_lds = P * PS <= WORK;
#pragma unroll
    for (int ti = 0; ti < UT; ti++)
#pragma unroll
        for (int tj = 0; tj <= ti; tj++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int u = ti * 16 + l15, v = tj * 16 + l4 + 4 * reg;
                if (u < U && v < U && u >= v) {
                    const double s = sacc[ti * (ti + 1) / 2 + tj][reg];
                    const int p = pm_r[ti], q2 = pm_c[tj * 4 + reg];
                    if (via_lds) {
                        work[p + q2 * PS] = s;
                        work[q2 + p * PS] = s;
                    } else {
                        cl.S[p + (long long)q2 * P] = s;
                        cl.S[q2 + (long long)p * P] = s;
                    }
                }
            }
    if (via_lds) {
        wave_sync();
        for (int j = l4; j < P; j += 4)
            for (int i0 = 0; i0 < P; i0 += 16) {
                const int i = i0 + l15;
                if (i < P) cl.S[i + (long long)j * P] = work[i + j * PS];
            }
    }
}

}  // namespace clrs
