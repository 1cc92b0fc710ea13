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

namespace clrs {

typedef double v4d_f __attribute__((ext_vector_type(4)));

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

// ---- small MFMA GEMM on LDS operands:  C[i,j] = sum_k A[k,i] B[k,j]  (i < M, j < N, k < K) -------------------------
// A: K x M (ld lda), B: K x N (ld ldb), C: M x N (ld ldc), all column-major in LDS.  Rows k >= K of A/B up to
// ceil4(K) and columns up to ceil16(M)/ceil16(N) must be readable and ZERO (the buffers are zero padded once).
// Tiles of 16 x 16 are dealt to the waves of the workgroup; lower_only skips tiles strictly above the diagonal.
__device__ __forceinline__ void lds_gemm_tn(const double *A, int lda, const double *B, int ldb, double *C, int ldc, int M, int N, int K,
                                            int wave, int nwaves, int lane) {
    const int tm = (M + 15) >> 4, tn = (N + 15) >> 4, K4 = (K + 3) & ~3;
    const int l15 = lane & 15, l4 = lane >> 4;
    for (int t = wave; t < tm * tn; t += nwaves) {
        const int ti = t % tm, tj = t / tm;
        const double *a = A + l4 + (ti * 16 + l15) * lda;   // A[k, i0 + c]
        const double *b = B + l4 + (tj * 16 + l15) * ldb;   // B[k, j0 + r]
        v4d_f acc = {0.0, 0.0, 0.0, 0.0};
        int k = 0;
        for (; k + 16 <= K4; k += 16) {   // 8 LDS reads in flight per 4 MFMAs
            const double a0 = a[k], a1 = a[k + 4], a2 = a[k + 8], a3 = a[k + 12];
            const double b0 = b[k], b1 = b[k + 4], b2 = b[k + 8], b3 = b[k + 12];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b2, a2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b3, a3, acc, 0, 0, 0);
        }
        for (; k < K4; k += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b[k], a[k], acc, 0, 0, 0);
        // D[r][c] = C[i0 + c, j0 + r]; lane holds c = lane & 15, r = (lane >> 4) + 4 * reg
        const int i = ti * 16 + l15;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int j = tj * 16 + l4 + 4 * reg;
            if (i < M && j < N) C[i + j * ldc] = acc[reg];
        }
    }
}

// ---- forward substitution Z <- L^-1 Z, blocked by 16 rows, wave-level -------------------------------------------------
// Diagonal 16 x 16 solve: a wave holds 4 columns x 16 rows, one element per lane (row = lane & 15).  Step k broadcasts
// the finished x_k across the 16 lanes of its column with a DPP row broadcast (no LDS round trip) and every lane
// below row k eliminates it with one FMA: a chain of 16 x (mul, dpp, fma) instead of n^2/2 dependent LDS reads.
template <int K>
__device__ __forceinline__ double bcast16(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x150 + K, 0xf, 0xf, false);   // row_newbcast:K
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x150 + K, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int K>
struct Trsm16 {
    static __device__ __forceinline__ void run(double &x0, double &x1, const double (&Lrow)[16], double dinv, int row) {
        const double b0 = bcast16<K>(x0 * dinv), b1 = bcast16<K>(x1 * dinv);
        x0 = (row == K) ? b0 : __builtin_fma(-Lrow[K], b0, x0);   // Lrow[K] == 0 above the diagonal: finished rows stay
        x1 = (row == K) ? b1 : __builtin_fma(-Lrow[K], b1, x1);
        Trsm16<K + 1>::run(x0, x1, Lrow, dinv, row);
    }
};
template <>
struct Trsm16<16> {
    static __device__ __forceinline__ void run(double &, double &, const double (&)[16], double, int) {}
};

// L: lower triangular in LDS (ld ldl, zero above the diagonal and beyond n up to ceil16(n)); dinv[i] = 1 / L[i,i]
// (0 for i >= n); Z: ceil16(n) x ncols in LDS (ld ldz), rows >= n zero.  Must be called by all `nwaves` waves.
__device__ __forceinline__ void lds_trsm_lower(const double *L, int ldl, const double *dinv, double *Z, int ldz, int n, int ncols, int wave,
                                               int nwaves, int lane) {
    const int npan = (n + 15) >> 4, row16 = lane & 15, cg4 = lane >> 4;
    const int ngroups = (ncols + 3) >> 2;
    for (int pb = 0; pb < npan; pb++) {
        const int r0 = pb * 16, row = r0 + row16;
        double Lrow[16];
#pragma unroll
        for (int k = 0; k < 16; k++) Lrow[k] = L[row + (r0 + k) * ldl];
        const double di = dinv[row];
        for (int g = wave; g < ngroups; g += 2 * nwaves) {   // two column groups per pass: independent chains interleave
            const int c0 = g * 4 + cg4, c1 = (g + nwaves) * 4 + cg4;
            const bool v0 = c0 < ncols, v1 = c1 < ncols;
            double x0 = v0 ? Z[row + c0 * ldz] : 0.0, x1 = v1 ? Z[row + c1 * ldz] : 0.0;
            Trsm16<0>::run(x0, x1, Lrow, di, row16);
            if (v0) Z[row + c0 * ldz] = x0;
            if (v1) Z[row + c1 * ldz] = x1;
        }
        if (pb + 1 < npan) {
            __syncthreads();
            // trailing update  Z[r0+16:, :] -= L[r0+16:, r0:r0+16] Z[r0:r0+16, :]   (MFMA, 16 x 16 tiles)
            const int tm = npan - pb - 1, tn = (ncols + 15) >> 4;
            for (int t = wave; t < tm * tn; t += nwaves) {
                const int i0 = r0 + 16 + (t % tm) * 16, j0 = (t / tm) * 16;
                v4d_f acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 16; kk += 4) {
                    const double zb = Z[(r0 + kk + cg4) + (j0 + row16) * ldz];    // a-operand: Z[k, j0 + r]
                    const double la = L[(i0 + row16) + (r0 + kk + cg4) * ldl];    // b-operand: L[i0 + c, k]
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(zb, la, acc, 0, 0, 0);
                }
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int j = j0 + cg4 + 4 * reg;
                    if (j < ncols) Z[(i0 + row16) + j * ldz] -= acc[reg];
                }
            }
            __syncthreads();
        }
    }
}

template <int NMAX>
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
            for (int e = tid; e < n16 * ldn; e += 256) {
                const int i = e % ldn, j = e / ldn;
                const bool in = i < n && j < n;
                Ys[e] = in ? Yg[i + (long long)j * n] : 0.0;
                Ls[e] = (in && i >= j) ? Lg[i + (long long)j * n] : 0.0;
            }
            const double *Vg = tb.stat + k.v_off;
            for (int e = tid; e < UR16 * ldn; e += 256) {
                const int i = e % ldn, j = e / ldn;
                Vs[e] = (i < n && j < UR) ? Vg[i + (long long)j * n] : 0.0;
                TYs[e] = 0.0;   // the padding rows / columns of T_Y are read by the next contraction
            }
            if (!k.sym) {
                const double *Wg = tb.stat + k.w_off;
                for (int e = tid; e < UL16 * ldn; e += 256) {
                    const int i = e % ldn, j = e / ldn;
                    ZLs[e] = (i < n && j < UL) ? Wg[i + (long long)j * n] : 0.0;
                }
            }
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
            lds_trsm_lower(Ls, ldn, dinv, Vs, ldn, n, UR, wave, 4, lane);
            if (!k.sym) {
                __syncthreads();
                lds_trsm_lower(Ls, ldn, dinv, ZLs, ldn, n, UL, wave, 4, lane);
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

}  // namespace clrs
