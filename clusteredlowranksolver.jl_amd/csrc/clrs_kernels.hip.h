// clrs_kernels.hip.h -- gfx950 (CDNA4) device kernels of the clustered low-rank SDP hot path.
//
// Everything here is written for MI355X only: 64-lane wavefronts, v_mfma_f64_16x16x4_f64 for the
// dense contractions, LDS-staged tiles, device-resident descriptor tables so that ONE launch
// processes the same step of every PSD block / cluster ("grouped" kernels).
//
// Kernel            replaces (reference, src/...)                         bound
// k_gemm_f64        matmul_threaded! tools.jl:175-266 (all GEMMs)         MFMA fp64
// k_trsm_diag       Arblib.approx_solve_tril!/triu! solver.jl:1258,1538   latency / LDS
// k_potrf_diag      approx_cholesky! tools.jl:75-107                      latency / LDS
// k_schur_gather    S accumulation loops solver.jl:1176-1212 + symmetric! HBM / L2 gather
// k_gather_scalar   A_Y extraction solver.jl:1152-1170                    HBM
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace clrs {

typedef double v4d __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// grouped GEMM:  C = alpha * op(A) op(B) + beta * C   (column-major, fp64, MFMA 16x16x4)
// ------------------------------------------------------------------------------------------------
struct GemmDesc {
    const double *A, *B;
    double *C;
    int M, N, K;
    int lda, ldb, ldc;
    int ta, tb;          // 0: as stored, 1: transposed
    int lower_only;      // skip tiles strictly above the diagonal (SYRK-style updates)
    int pad0;
    double alpha, beta;
    long long sA, sB, sC;  // strides of a strided batch
};
struct GemmTile { int desc, batch, tm, tn; };

constexpr int GEMM_BM = 64, GEMM_BN = 64, GEMM_BK = 16, GEMM_LDS = 80;  // 80: conflict-free ds_read_b64 across the 4 k-rows

__global__ __launch_bounds__(256) void k_gemm_f64(const GemmDesc *__restrict__ descs, const GemmTile *__restrict__ tiles) {
    const GemmTile t = tiles[blockIdx.x];
    const GemmDesc d = descs[t.desc];
    const double *__restrict__ A = d.A + (long long)t.batch * d.sA;
    const double *__restrict__ B = d.B + (long long)t.batch * d.sB;
    double *__restrict__ C = d.C + (long long)t.batch * d.sC;
    const int m0 = t.tm * GEMM_BM, n0 = t.tn * GEMM_BN;
    __shared__ double As[GEMM_BK][GEMM_LDS];
    __shared__ double Bs[GEMM_BK][GEMM_LDS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
    const int l15 = lane & 15, l4 = lane >> 4;
    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};

    for (int k0 = 0; k0 < d.K; k0 += GEMM_BK) {
        // stage op(A)[m0.., k0..] -> As[k][i] and op(B)[k0.., n0..] -> Bs[k][j]; 4 elements per thread each,
        // consecutive lanes walk the contiguous dimension of the source.
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int e = tid + 256 * q;
            int i, k;
            if (d.ta == 0) { i = e & 63; k = e >> 6; } else { k = e & 15; i = e >> 4; }
            const int gi = m0 + i, gk = k0 + k;
            double v = 0.0;
            if (gi < d.M && gk < d.K) v = d.ta == 0 ? A[gi + (long long)gk * d.lda] : A[gk + (long long)gi * d.lda];
            As[k][i] = v;
            int j, kb;
            if (d.tb == 0) { kb = e & 15; j = e >> 4; } else { j = e & 63; kb = e >> 6; }
            const int gj = n0 + j, gkb = k0 + kb;
            double w = 0.0;
            if (gj < d.N && gkb < d.K) w = d.tb == 0 ? B[gkb + (long long)gj * d.ldb] : B[gj + (long long)gkb * d.ldb];
            Bs[kb][j] = w;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GEMM_BK; kk += 4) {
            // The MFMA computes D[r][c] = sum_k Aop[r][k] Bop[k][c] with c on lane&15.  We feed Aop = op(B)^T and
            // Bop = op(A)^T so that c runs along the rows i of C (contiguous in memory) -> coalesced C stores.
            const double a0 = As[kk + l4][wm + l15], a1 = As[kk + l4][wm + 16 + l15];
            const double b0 = Bs[kk + l4][wn + l15], b1 = Bs[kk + l4][wn + 16 + l15];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a1, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a0, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    // D layout of v_mfma_f64_16x16x4_f64: c = lane & 15, r = (lane >> 4) + 4 * reg.  Here r indexes j, c indexes i.
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int ni = 0; ni < 2; ni++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int gi = m0 + wm + mi * 16 + l15;
                const int gj = n0 + wn + ni * 16 + l4 + 4 * reg;
                if (gi < d.M && gj < d.N) {
                    double *p = C + gi + (long long)gj * d.ldc;
                    const double v = d.alpha * acc[mi][ni][reg];
                    *p = d.beta == 0.0 ? v : v + d.beta * *p;
                }
            }
}

// ------------------------------------------------------------------------------------------------
// triangular solve with a diagonal block (n <= 64): one thread per right-hand-side vector.
//   trans == 0:  L x = b (forward)      trans == 1:  L^T x = b (backward)
// element i of vector v lives at B[v * vs + i * es]  (es = 1, vs = ldb: columns of B, "left" solve;
// es = ldb, vs = 1: rows of B, i.e. X L^T = B, the panel update of the blocked Cholesky).
// ------------------------------------------------------------------------------------------------
struct TrsmDesc {
    const double *L;
    double *B;
    int ldl, n, nvec, trans;
    long long es, vs;
};
struct TrsmWork { int desc, chunk; };
constexpr int TRSM_NB = 64;

__global__ __launch_bounds__(64) void k_trsm_diag(const TrsmDesc *__restrict__ descs, const TrsmWork *__restrict__ work) {
    const TrsmWork w = work[blockIdx.x];
    const TrsmDesc d = descs[w.desc];
    __shared__ double Ls[TRSM_NB][TRSM_NB + 1];
    __shared__ double xs[TRSM_NB][TRSM_NB];   // xs[i][v]
    const int tid = threadIdx.x, n = d.n;
    const int v0 = w.chunk * 64;
    const int nv = min(64, d.nvec - v0);
    for (int e = tid; e < n * n; e += 64) {
        const int i = e % n, k = e / n;
        Ls[i][k] = (k <= i) ? d.L[i + (long long)k * d.ldl] : 0.0;
    }
    if (d.es == 1) {
        for (int e = tid; e < n * nv; e += 64) { const int i = e % n, v = e / n; xs[i][v] = d.B[(long long)(v0 + v) * d.vs + i]; }
    } else {
        for (int e = tid; e < n * nv; e += 64) { const int v = e % nv, i = e / nv; xs[i][v] = d.B[(long long)(v0 + v) * d.vs + (long long)i * d.es]; }
    }
    __syncthreads();
    if (tid < nv) {
        if (d.trans == 0) {
            for (int i = 0; i < n; i++) {
                double s = xs[i][tid];
                for (int k = 0; k < i; k++) s -= Ls[i][k] * xs[k][tid];
                xs[i][tid] = s / Ls[i][i];
            }
        } else {
            for (int i = n - 1; i >= 0; i--) {
                double s = xs[i][tid];
                for (int k = i + 1; k < n; k++) s -= Ls[k][i] * xs[k][tid];
                xs[i][tid] = s / Ls[i][i];
            }
        }
    }
    __syncthreads();
    if (d.es == 1) {
        for (int e = tid; e < n * nv; e += 64) { const int i = e % n, v = e / n; d.B[(long long)(v0 + v) * d.vs + i] = xs[i][v]; }
    } else {
        for (int e = tid; e < n * nv; e += 64) { const int v = e % nv, i = e / nv; d.B[(long long)(v0 + v) * d.vs + (long long)i * d.es] = xs[i][v]; }
    }
}

// ------------------------------------------------------------------------------------------------
// Cholesky of a diagonal block (n <= 64), one workgroup per matrix, LDS resident, right-looking.
// On a non-positive pivot records `code` (atomicMin) in *info; the lower triangle is overwritten by L.
// ------------------------------------------------------------------------------------------------
struct PotrfDesc {
    double *A;
    int lda, n, code, pad;
};
constexpr int POTRF_NB = 64;

__global__ __launch_bounds__(256) void k_potrf_diag(const PotrfDesc *__restrict__ descs, int *__restrict__ info) {
    const PotrfDesc d = descs[blockIdx.x];
    __shared__ double As[POTRF_NB][POTRF_NB + 1];
    __shared__ int failed;
    const int tid = threadIdx.x, n = d.n;
    if (tid == 0) failed = 0;
    for (int e = tid; e < n * n; e += 256) {
        const int i = e % n, j = e / n;
        if (i >= j) As[i][j] = d.A[i + (long long)j * d.lda];
    }
    __syncthreads();
    for (int k = 0; k < n; k++) {
        if (tid == 0) {
            const double dk = As[k][k];
            if (!(dk > 0.0)) failed = 1;
            As[k][k] = sqrt(dk);
        }
        __syncthreads();
        if (tid > k && tid < n) As[tid][k] /= As[k][k];
        __syncthreads();
        const int m = n - k - 1;
        for (int e = tid; e < m * m; e += 256) {
            const int i = k + 1 + e % m, j = k + 1 + e / m;
            if (i >= j) As[i][j] -= As[i][k] * As[j][k];
        }
        __syncthreads();
    }
    if (tid == 0 && failed) atomicMin(info, d.code);
    for (int e = tid; e < n * n; e += 256) {
        const int i = e % n, j = e / n;
        if (i >= j) d.A[i + (long long)j * d.lda] = As[i][j];
    }
}

// ------------------------------------------------------------------------------------------------
// Schur gather:  S_j[p,q] = sum_{l low rank} sum_{t1 in p, t2 in q} lam1 lam2 GX_l[L1,R2] GY_l[L2,R1]
//                         + sum_{l dense} Sd_l[inv_l[q], inv_l[p]]              for p <= q, mirrored.
// One thread per (p,q); each S entry is written exactly once (deterministic, no atomics).
// ------------------------------------------------------------------------------------------------
struct SBlockDesc {
    int kind, ldg, cnt, pad;
    const double *GX, *GY;     // low rank: UL x UR pairing matrices
    const int *tptr;           // [P+1] CSR of this block's terms over the cluster's constraint index
    const int *tL, *tR;        // per term: global left / right unique-vector index
    const double *tlam;        // per term: lambda
    const double *Sd;          // dense: cnt x cnt, Sd[i,k] = <A_i, X^-1 A_k Y>
    const int *inv;            // dense: [P] constraint -> index in the block's list or -1
};
struct SClusterDesc {
    double *S;
    int P, b0, b1, pad;
};
struct STile { int cluster, ti, tj, pad; };

__global__ __launch_bounds__(256) void k_schur_gather(const SClusterDesc *__restrict__ cl, const SBlockDesc *__restrict__ bl,
                                                      const STile *__restrict__ tiles) {
    const STile t = tiles[blockIdx.x];
    const SClusterDesc c = cl[t.cluster];
    const int p = t.ti * 16 + (threadIdx.x & 15), q = t.tj * 16 + (threadIdx.x >> 4);
    if (p >= c.P || q >= c.P || p > q) return;
    double acc = 0.0;
    for (int b = c.b0; b < c.b1; b++) {
        const SBlockDesc d = bl[b];
        if (d.kind == 0) {
            const int a0 = d.tptr[p], a1 = d.tptr[p + 1], b0 = d.tptr[q], b1 = d.tptr[q + 1];
            for (int t1 = a0; t1 < a1; t1++) {
                const int L1 = d.tL[t1], R1 = d.tR[t1];
                const double l1 = d.tlam[t1];
                for (int t2 = b0; t2 < b1; t2++) {
                    const int L2 = d.tL[t2], R2 = d.tR[t2];
                    acc += (l1 * d.tlam[t2]) * (d.GX[L1 + (long long)R2 * d.ldg] * d.GY[L2 + (long long)R1 * d.ldg]);
                }
            }
        } else {
            const int i = d.inv[p], k = d.inv[q];
            if (i >= 0 && k >= 0) acc += d.Sd[k + (long long)i * d.cnt];
        }
    }
    c.S[p + (long long)q * c.P] = acc;
    c.S[q + (long long)p * c.P] = acc;
}

// out[i] = src[idx[i]]
__global__ void k_gather_scalar(double *__restrict__ out, const double *__restrict__ src, const long long *__restrict__ idx, long long n) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i < n && idx[i] >= 0) out[i] = src[idx[i]];   // negative: the entry is produced by the fused kernel
}

// y = a - b   (dy right-hand side: rhs_y - sum_j u_j, solver.jl:1550-1553)
__global__ void k_sub(double *__restrict__ y, const double *__restrict__ a, const double *__restrict__ b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = a[i] - b[i];
}

// zero the strict upper triangles of the matrices listed in descs (output formatting of L, tools.jl:100-105)
__global__ void k_zero_upper(const PotrfDesc *__restrict__ descs) {
    const PotrfDesc d = descs[blockIdx.y];
    const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (e >= (long long)d.n * d.n) return;
    const int i = (int)(e % d.n), j = (int)(e / d.n);
    if (i < j) d.A[i + (long long)j * d.lda] = 0.0;
}

}  // namespace clrs
