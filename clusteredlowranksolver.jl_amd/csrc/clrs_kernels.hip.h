// clrs_kernels.hip.h -- gfx950 (CDNA4) device kernels of the clustered low-rank SDP hot path.
//
// Everything here is written for MI355X only: 64-lane wavefronts, v_mfma_f64_16x16x4_f64 for the
// dense contractions, LDS-staged tiles, device-resident descriptor tables so that ONE launch
// processes the same step of every PSD block / cluster ("grouped" kernels).
//
// Kernel            replaces (reference, src/...)                         bound
// k_gemm_f64_t<BM,BN,TA,TB> matmul_threaded! tools.jl:175-266 (all GEMMs)  MFMA fp64
// k_trsm_diag       Arblib.approx_solve_tril!/triu! solver.jl:1258,1538   latency / LDS   (triangles within one 64-wide block)
// k_trtri_diag      the same for n > 512: inverses of the 64 x 64 leaves, joined and applied by GEMMs (plan_trsm_blockinv)
// k_potrf_diag      approx_cholesky! tools.jl:75-107                      latency / LDS   (matrices within one 64-wide block)
// k_chol_level      the same beyond one block: ONE launch per block column (column workgroups + 128 x 128 trailing tiles)
// k_chol_pack / k_chol_unpack   solver.jl:1245-1269 as one factorisation of [S .; B^T 0]: L, L^-1 B and Q together
// k_trtri32 / k_dense_T32       X^-1 A Y of the dense branch, solver.jl:1089-1097, one wave per matrix    MFMA fp64 / HBM
// k_schur_gather    S accumulation loops solver.jl:1176-1212 + symmetric! HBM / L2 gather
// k_gather_scalar   A_Y extraction solver.jl:1152-1170                    HBM
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "clrs_wave.hip.h"

namespace clrs {

typedef double v4d __attribute__((ext_vector_type(4)));

// A pointer read from a descriptor table in memory is a GENERIC pointer: its loads are flat_load, which the hardware also counts as
// LDS operations -- a wait for an LDS read then waits for every global prefetch in flight as well.  as_global() states what the
// host knows (every table entry points into device memory) and turns them into global_load / global_store.
template <class T> using gptr = T __attribute__((address_space(1))) *;
template <class T> __device__ __forceinline__ gptr<T> as_global(T *p) { return (gptr<T>)p; }

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

constexpr int GEMM_BM = 64, GEMM_BN = 64, GEMM_BK = 16;   // small-tile kernel; the large-tile kernel is 128 x 128 x 16
constexpr int gemm_padk(int BM) { return BM <= 64 ? 8 : 16; }

// Workgroup tile BM x BN, 4 waves in a 2 x 2 arrangement, each wave (BM/2) x (BN/2) = (BM/32) x (BN/32) MFMA tiles.
// LDS rows are padded by 16 doubles: the four k-rows an MFMA operand read touches then fall into different banks.
// Double buffered: the global loads of step k+1 are issued before the MFMAs of step k and written to the other
// buffer after them, one barrier per step.  With 128 x 128 tiles a step is 64 MFMAs per wave (4096 cycles of the matrix
// pipe) against 16 global loads per thread, which hides the load latency even with one workgroup per CU.
//
// The transposes are template parameters and the k-loop is unrolled over the two buffers, so that every LDS address of the loop is a
// per-lane base plus an immediate and every global address a per-lane offset (computed once) plus a uniform base that advances with
// k.  With run-time transposes the SQ counters of this kernel showed SEVEN vector ALU instructions per MFMA (index arithmetic and
// selects of the staging maps) and the matrix pipe busy 48 % of the time: one wave per SIMD issues in order, so whatever it spends on
// address arithmetic the matrix pipe waits.  Rows / columns beyond M / N are read from a clamped address and never stored; only the
// last, partial chunk of k is zero filled.
// OFF = unsigned: the 32-bit per-lane byte offsets described below (every product whose operands they reach: gemm_offsets_reach);
// OFF = unsigned long long: the same kernel with 64-bit offsets (a vector add per load) for leading dimensions beyond that -- the Gram
// product of a dense block with n >= 2048 has lda = ldb = n^2.
template <int BM, int BN, int TA, int TB, typename OFF = unsigned>
__global__ __launch_bounds__(256, (BM <= 64 ? 4 : 2)) void k_gemm_f64_t(const GemmDesc *__restrict__ descs, const GemmTile *__restrict__ tiles) {
    // small tiles: a [k][i] stride of BM + 8 keeps a workgroup at 36 KB of LDS, so that FOUR of them share a compute unit: a grouped
    // launch of 1024 tiles (the Gram contractions of the dense branch) then runs in one round instead of one and a third
    constexpr int BK = GEMM_BK, PADK = gemm_padk(BM), LDA_S = BM + PADK, LDB_S = BN + PADK, MT = BM / 32, NT = BN / 32, EA = BM * BK / 256, EB = BN * BK / 256;
    // Two LDS layouts per operand, chosen so that BOTH the staging writes and the MFMA operand reads are conflict free:
    //   source contiguous along the tile row/column index (ta == 0 / tb == 1):  [k][i], row stride BM + 16
    //   source contiguous along k               (ta == 1 / tb == 0):            [i][k], row stride BK + 2
    constexpr int LDT = BK + 2, ABUF = BK * LDA_S, BBUF = BK * LDB_S;
    static_assert(BK * LDA_S >= BM * LDT && BK * LDB_S >= BN * LDT, "both layouts must fit the same buffer");
    const GemmTile t = tiles[blockIdx.x];
    const GemmDesc d = descs[t.desc];
    const double *__restrict__ A = d.A + (long long)t.batch * d.sA;
    const double *__restrict__ B = d.B + (long long)t.batch * d.sB;
    double *__restrict__ C = d.C + (long long)t.batch * d.sC;
    const int m0 = t.tm * BM, n0 = t.tn * BN;
    extern __shared__ __attribute__((aligned(16))) double gsm[];
    double *As = gsm, *Bs = gsm + 2 * ABUF;                   // As[buf][k][i], Bs[buf][k][j]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = (wave & 1) * (BM / 2), wn = (wave >> 1) * (BN / 2);
    const int l15 = lane & 15, l4 = lane >> 4;
    v4d acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
        for (int b = 0; b < NT; b++) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
    // staging map: consecutive lanes walk the contiguous dimension of the source.  Element q of a thread: e = tid + 256 q,
    //   TA == 0: (i, k) = (e % BM, e / BM) -> LDS [k][i];   TA == 1: (k, i) = (e % BK, e / BK) -> LDS [i][k]       (B alike with j)
    // Byte offsets from the tile's first row / column, 32 bits (at most 128 lda doubles), on top of a uniform base: the loads take the
    // scalar-base + vector-offset form and cost no vector ALU instruction.  The bases are cast to the GLOBAL address space: a pointer
    // read from a descriptor in memory is a generic one, its loads are flat_load -- which also count as LDS operations, so that every
    // wait for an MFMA operand read waited for the prefetch of the next chunk as well.
    typedef const char __attribute__((address_space(1))) *gbytes_t;
    typedef const double __attribute__((address_space(1))) *gdouble_t;
    OFF oa[EA], ob[EB];
    const long long lda = d.lda, ldb = d.ldb;
#pragma unroll
    for (int q = 0; q < EA; q++) {
        const int e = tid + 256 * q, i = TA == 0 ? e % BM : e / BK, k = TA == 0 ? e / BM : e % BK, ic = min(i, d.M - 1 - m0);
        oa[q] = (OFF)(TA == 0 ? ic + k * lda : k + ic * lda) * (OFF)8;
    }
#pragma unroll
    for (int q = 0; q < EB; q++) {
        const int e = tid + 256 * q, j = TB == 0 ? e / BK : e % BN, k = TB == 0 ? e % BK : e / BN, jc = min(j, d.N - 1 - n0);
        ob[q] = (OFF)(TB == 0 ? k + jc * ldb : jc + k * ldb) * (OFF)8;
    }
    const gbytes_t Ag = (gbytes_t)(A + (TA == 0 ? (long long)m0 : m0 * lda)), Bg = (gbytes_t)(B + (TB == 0 ? n0 * ldb : (long long)n0));
    double *const sa = As + (TA == 0 ? (tid / BM) * LDA_S + tid % BM : (tid / BK) * LDT + tid % BK);     // element q: + q * SQA
    double *const sb = Bs + (TB == 0 ? (tid / BK) * LDT + tid % BK : (tid / BN) * LDB_S + tid % BN);
    constexpr int SQA = TA == 0 ? (256 / BM) * LDA_S : (256 / BK) * LDT, SQB = TB == 0 ? (256 / BK) * LDT : (256 / BN) * LDB_S;
    const double *const ra_ = As + (TA == 0 ? l4 * LDA_S + wm + l15 : (wm + l15) * LDT + l4);              // operand reads: + immediates
    const double *const rb_ = Bs + (TB == 0 ? (wn + l15) * LDT + l4 : l4 * LDB_S + wn + l15);
    double ra[EA], rb[EB];
    auto fetch = [&](int k0, auto tail) {
        const gbytes_t Ak = Ag + (TA == 0 ? k0 * lda : (long long)k0) * 8, Bk = Bg + (TB == 0 ? (long long)k0 : k0 * ldb) * 8;
        if constexpr (!decltype(tail)::value) {
#pragma unroll
            for (int q = 0; q < EA; q++) ra[q] = *(gdouble_t)(Ak + oa[q]);
#pragma unroll
            for (int q = 0; q < EB; q++) rb[q] = *(gdouble_t)(Bk + ob[q]);
        } else {                                              // the last chunk of k: rows k >= K are zero
#pragma unroll
            for (int q = 0; q < EA; q++) {
                const int e = tid + 256 * q, k = TA == 0 ? e / BM : e % BK;
                ra[q] = k0 + k < d.K ? *(gdouble_t)(Ak + oa[q]) : 0.0;
            }
#pragma unroll
            for (int q = 0; q < EB; q++) {
                const int e = tid + 256 * q, k = TB == 0 ? e % BK : e / BN;
                rb[q] = k0 + k < d.K ? *(gdouble_t)(Bk + ob[q]) : 0.0;
            }
        }
    };
    // `cur` (0 / ABUF doubles): the buffer the MFMAs read; the staging writes go to the other one
    auto stash = [&](int wo) {
#pragma unroll
        for (int q = 0; q < EA; q++) sa[wo + q * SQA] = ra[q];
#pragma unroll
        for (int q = 0; q < EB; q++) sb[wo + q * SQB] = rb[q];
    };
    // 16-row / 16-column pieces of this wave's quarter that lie inside C: a tile on the edge of C (the roofline instances are one row
    // and column beyond a multiple of the tile: 65 of 1089 tiles) skips the MFMAs of the pieces outside, instead of costing a full tile
    const int na = min(MT, max(0, (d.M - m0 - wm + 15) >> 4)), nb = min(NT, max(0, (d.N - n0 - wn + 15) >> 4));
    const bool inner = na == MT && nb == NT;
    auto mma = [&](int ro) {
        if (inner) {
#pragma unroll
            for (int kk = 0; kk < BK; kk += 4) {
                // The MFMA computes D[r][c] = sum_k Aop[r][k] Bop[k][c] with c on lane&15.  We feed Aop = op(B)^T and
                // Bop = op(A)^T so that c runs along the rows i of C (contiguous in memory) -> coalesced C stores.
                double av[MT], bv[NT];
#pragma unroll
                for (int a = 0; a < MT; a++) av[a] = ra_[ro + (TA == 0 ? kk * LDA_S + a * 16 : a * 16 * LDT + kk)];
#pragma unroll
                for (int b = 0; b < NT; b++) bv[b] = rb_[ro + (TB == 0 ? b * 16 * LDT + kk : kk * LDB_S + b * 16)];
#pragma unroll
                for (int a = 0; a < MT; a++)
#pragma unroll
                    for (int b = 0; b < NT; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[b], av[a], acc[a][b], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < BK; kk += 4) {
                double av[MT], bv[NT];
#pragma unroll
                for (int a = 0; a < MT; a++) av[a] = ra_[ro + (TA == 0 ? kk * LDA_S + a * 16 : a * 16 * LDT + kk)];
#pragma unroll
                for (int b = 0; b < NT; b++) bv[b] = rb_[ro + (TB == 0 ? b * 16 * LDT + kk : kk * LDB_S + b * 16)];
#pragma unroll
                for (int a = 0; a < MT; a++)
#pragma unroll
                    for (int b = 0; b < NT; b++)
                        if (a < na && b < nb) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[b], av[a], acc[a][b], 0, 0, 0);
            }
        }
    };
    static_assert(ABUF == BBUF, "one buffer offset for both operands");
    // chunks of BK: nfull whole ones, then a partial one (zero filled) if K is not a multiple.  The loop over the whole chunks has
    // no branch in it: loads of chunk c + 1 in flight during the MFMAs of chunk c, written to the other buffer after them.
    const int nfull = d.K / BK, rem = d.K % BK;
    int cur = 0;
    if (nfull > 0) fetch(0, std::false_type{});
    else fetch(0, std::true_type{});
    stash(0);
    __syncthreads();
    for (int c = 0; c + 1 < nfull; c++) {
        fetch((c + 1) * BK, std::false_type{});
        mma(cur);
        stash(cur ^ ABUF);
        __syncthreads();
        cur ^= ABUF;
    }
    if (nfull > 0 && rem > 0) {
        fetch(nfull * BK, std::true_type{});
        mma(cur);
        stash(cur ^ ABUF);
        __syncthreads();
        cur ^= ABUF;
    }
    mma(cur);
    // D layout of v_mfma_f64_16x16x4_f64: c = lane & 15, r = (lane >> 4) + 4 * reg.  Here r indexes j, c indexes i.
#pragma unroll
    for (int mi = 0; mi < MT; mi++)
#pragma unroll
        for (int ni = 0; ni < NT; ni++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int gi = m0 + wm + mi * 16 + l15;
                const int gj = n0 + wn + ni * 16 + l4 + 4 * reg;
                if (gi < d.M && gj < d.N) {
                    double __attribute__((address_space(1))) *p = (double __attribute__((address_space(1))) *)(C + gi + (long long)gj * d.ldc);
                    const double v = d.alpha * acc[mi][ni][reg];
                    *p = d.beta == 0.0 ? v : v + d.beta * *p;
                }
            }
}
// true if the 32-bit byte offsets of a BM x BN tile reach every element of both operands: an operand walked along k inside the tile spans
// BK leading dimensions, one walked along the tile's rows / columns BM (BN) of them
static inline bool gemm_offsets_reach(const GemmDesc &d, int BMN) {
    const long long ra = (d.ta == 0 ? (long long)GEMM_BK : (long long)BMN) * d.lda, rb = (d.tb == 0 ? (long long)BMN : (long long)GEMM_BK) * d.ldb;
    return (ra + BMN) * 8 < (1ll << 32) && (rb + BMN) * 8 < (1ll << 32);
}
constexpr size_t gemm_lds_bytes(int BM, int BN) { return (size_t)2 * GEMM_BK * ((BM + gemm_padk(BM)) + (BN + gemm_padk(BM))) * sizeof(double); }

// ------------------------------------------------------------------------------------------------
// Dense ("high rank") branch for blocks with n <= 32 and MANY matrices (SDPA-type problems; src/solver.jl:1089-1097):
// T_e = X^-1 A_e Y = Linv^T (Linv (A_e Y)) with Linv = chol(X_b)^-1, three 32 x 32 x 32 products per (block, matrix) through
// v_mfma_f64_16x16x4 by ONE WAVE: A_e, Y, Linv staged in LDS (zero padded to 32 x 32), each result chained into the next product
// as right operand in its accumulator registers, the last one stored from them.  The staged
// form spent two launches of one-thread-per-column substitutions (k_trsm_diag) and a batched GEMM with 32 x 32 tiles on it.
//   k_trtri32     Linv per block (one wave: lane c owns column c of the inverse), padded to 32 x 32 with leading dimension 32
//   k_dense_T32   DT32_WAVES consecutive matrices of one block per workgroup; Linv and Y of the block shared in LDS; T_e is written
//                 transposed (its only reader pairs it with a symmetric A_i)
// ------------------------------------------------------------------------------------------------
struct DenseTBlock {
    const double *L, *Y, *A;     // chol(X_b) and Y_b (n x n, column-major), the stack of cnt matrices A_e (n x n each)
    double *Linv, *TT;           // 32 x 32 scratch of the block; output stack T_e (n x n each)
    int n, cnt;
};
struct DenseTPair { int blk, e0; };
#define DT32_WAVES 4
#define DT32_ITER 2          // matrices per wave
#define DT32_LD 34
#define DT32_MS (32 * DT32_LD)
__global__ __launch_bounds__(64) void k_trtri32(const DenseTBlock *__restrict__ blocks) {
    __shared__ double Ls[32 * 33];
    const DenseTBlock b = blocks[blockIdx.x];
    const int n = b.n, lane = threadIdx.x;
    const gptr<const double> gL = as_global(b.L);
    const gptr<double> gLinv = as_global(b.Linv);
    {
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int e = lane + 64 * q, i = e % 32, c = e / 32;
            v[q] = gL[min(i, n - 1) + (long long)min(c, n - 1) * n];
        }
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int e = lane + 64 * q, i = e % 32, c = e / 32;
            Ls[i + 33 * c] = (i < n && c < n) ? v[q] : (i == c ? 1.0 : 0.0);
        }
    }
    __syncthreads();
    // column c = lane & 31 of the inverse by forward substitution, the column in registers: the reads of L are broadcasts that do
    // not depend on the chain, which leaves one FMA per (row, k) on it
    const int c = lane & 31;
    double x[32];
#pragma unroll
    for (int i = 0; i < 32; i++) {
        double s = i == c ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; k++) s -= Ls[i + 33 * k] * x[k];         // x[k] = 0 for k < c
        x[i] = i < c ? 0.0 : s / Ls[i + 33 * i];
    }
    if (lane < 32) {
#pragma unroll
        for (int i = 0; i < 32; i++) gLinv[i + 32 * c] = (i < n && c < n) ? x[i] : 0.0;
    }
}
// D = X Y for 32 x 32 operands in LDS (leading dimension DT32_LD).  acc[ti][tj]: lane holds column tj*16 + (lane & 15), rows
// ti*16 + (lane >> 4) + 4 reg -- which is exactly the B-operand layout of the next product's k-step (tile ti, reg), so that a
// result is chained into the next MFMA as RIGHT operand without leaving the registers (dt32_mm_chain).
__device__ __forceinline__ void dt32_mm(const double *X, const double *Yb, v4d (&acc)[2][2], int l15, int l4) {
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 32; kk += 4) {
        double xv[2], yv[2];
#pragma unroll
        for (int a = 0; a < 2; a++) xv[a] = X[(a * 16 + l15) + DT32_LD * (kk + l4)];
#pragma unroll
        for (int b = 0; b < 2; b++) yv[b] = Yb[(kk + l4) + DT32_LD * (b * 16 + l15)];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[a], yv[b], acc[a][b], 0, 0, 0);
    }
}
// out = op(Li) yin with Li lower triangular in LDS and yin in accumulator layout; XT: op = transpose.  The zero tile of the
// triangle (k > row for Li, k < row for Li^T) is skipped.
template <bool XT>
__device__ __forceinline__ void dt32_mm_chain(const double *Li, const v4d (&yin)[2][2], v4d (&out)[2][2], int l15, int l4) {
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) out[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kt = 0; kt < 2; kt++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int k = kt * 16 + 4 * reg + l4;
#pragma unroll
            for (int a = 0; a < 2; a++) {
                if ((!XT && kt > a) || (XT && kt < a)) continue;
                const double xv = XT ? Li[k + DT32_LD * (a * 16 + l15)] : Li[(a * 16 + l15) + DT32_LD * k];
#pragma unroll
                for (int b = 0; b < 2; b++) out[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv, yin[kt][b][reg], out[a][b], 0, 0, 0);
            }
        }
}
// FULL: n == 32, the usual case -- no guards on the loads, shifts instead of divisions by n in the copy-out
template <bool FULL>
__device__ __forceinline__ void dense_T32_body(const DenseTBlock &b, const DenseTPair &pr, double *dts) {
    const int n = FULL ? 32 : b.n, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
    double *Li = dts, *Ys = dts + DT32_MS, *B0 = dts + 2 * DT32_MS + wave * DT32_MS;     // B0: this wave's matrix, then its result
    const gptr<const double> gLinv = as_global((const double *)b.Linv), gY = as_global(b.Y), gA = as_global(b.A);
    const gptr<double> gTT = as_global(b.TT);
    {
        double vl[4], vy[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int e = tid + 64 * DT32_WAVES * q, i = e % 32, c = e / 32;
            vl[q] = gLinv[e];
            vy[q] = gY[min(i, n - 1) + (long long)min(c, n - 1) * n];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int e = tid + 64 * DT32_WAVES * q, i = e % 32, c = e / 32;
            Li[i + DT32_LD * c] = vl[q];
            Ys[i + DT32_LD * c] = (FULL || (i < n && c < n)) ? vy[q] : 0.0;
        }
    }
    // DT32_ITER matrices per wave, the next one fetched into registers while the products of the current one run.  The addresses are
    // clamped into the matrix (and a wave beyond the last matrix re-reads the first): no branch around any load
    double areg[16];
    auto fetch = [&](int e) {
        const bool live = e < b.cnt;
        const gptr<const double> A = gA + (long long)(live ? e : 0) * n * n;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int o = lane + 64 * r, i = o % 32, c = o / 32;
            areg[r] = FULL ? A[o] : A[min(i, n - 1) + (long long)min(c, n - 1) * n];
        }
        if (!FULL) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int o = lane + 64 * r, i = o % 32, c = o / 32;
                areg[r] = (i < n && c < n) ? areg[r] : 0.0;
            }
        }
    };
    fetch(pr.e0 + wave);
    __syncthreads();                                        // Linv, Y staged
    for (int it = 0; it < DT32_ITER; it++) {
        const int e = pr.e0 + it * DT32_WAVES + wave;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int o = lane + 64 * r;
            B0[(o % 32) + DT32_LD * (o / 32)] = areg[r];
        }
        if (it + 1 < DT32_ITER) fetch(e + DT32_WAVES);
        v4d q1[2][2], q2[2][2];
        dt32_mm(B0, Ys, q1, l15, l4);                       // A Y
        dt32_mm_chain<false>(Li, q1, q2, l15, l4);          // Linv (A Y)
        dt32_mm_chain<true>(Li, q2, q1, l15, l4);           // Linv^T Linv A Y = X^-1 A Y
        // T_e goes out TRANSPOSED, straight from the accumulators (lane & 15 runs along a row of T: 128-byte pieces): its only reader is
        // the Gram product <A_i, T_e>, and A_i is symmetric -- <A_i, T_e> = <A_i^T, T_e^T> = <A_i, T_e^T>.  No trip through LDS.
        if (e < b.cnt) {
            const gptr<double> T = gTT + (long long)e * n * n;
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int bb = 0; bb < 2; bb++)
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) {
                        const int r = a * 16 + l4 + 4 * reg, cc = bb * 16 + l15;
                        if (FULL || (r < n && cc < n)) T[cc + n * r] = q1[a][bb][reg];
                    }
        }
    }
}
__global__ __launch_bounds__(64 * DT32_WAVES) void k_dense_T32(const DenseTBlock *__restrict__ blocks, const DenseTPair *__restrict__ pairs) {
    extern __shared__ __attribute__((aligned(16))) double dts[];
    const DenseTPair pr = pairs[blockIdx.x];
    const DenseTBlock b = blocks[pr.blk];
    if (b.n == 32) dense_T32_body<true>(b, pr, dts);
    else dense_T32_body<false>(b, pr, dts);
}
constexpr size_t dense_T32_lds_bytes() { return (size_t)(2 + DT32_WAVES) * DT32_MS * sizeof(double); }

// Up to 64 x 64 elements G[i si + j sj] (i < rows, j < cols) into Z[i + j ldz] over the frame i < fr, j < fc, zero outside the
// valid part (MODE 1: and above the diagonal; MODE 2: and ones on the diagonal of the padding).  All sixteen loads of a thread are
// issued before the first store (addresses clamped into the valid part, not guarded): one round trip to memory instead of one per
// 16 x 16 tile -- these kernels are links of a chain of dependent launches.  JFAST: consecutive threads walk j (sj == 1).
template <int MODE, bool JFAST = false>
__device__ __forceinline__ void tile64_load(double *Z, int ldz, gptr<const double> G, long long si, long long sj, int rows, int cols, int fr, int fc, int tid) {
    double v[16];
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int e = tid + 256 * q, i = JFAST ? e >> 6 : e & 63, j = JFAST ? e & 63 : e >> 6;
        v[q] = G[min(i, rows - 1) * si + min(j, cols - 1) * sj];
    }
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int e = tid + 256 * q, i = JFAST ? e >> 6 : e & 63, j = JFAST ? e & 63 : e >> 6;
        const bool in = i < rows && j < cols && (MODE == 0 || i >= j);
        if (i < fr && j < fc) Z[i + j * ldz] = in ? v[q] : ((MODE == 2 && i == j && i >= rows) ? 1.0 : 0.0);
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

__global__ __launch_bounds__(256) void k_trsm_diag(const TrsmDesc *__restrict__ descs, const TrsmWork *__restrict__ work) {
    // One workgroup per (problem, chunk of 64 right-hand sides): L and the chunk live in LDS, the substitution itself is
    // the wave-level DPP / MFMA routine of clrs_wave.hip.h (16-row panels, 4 right-hand sides per 16-lane group).
    const TrsmWork w = work[blockIdx.x];
    const TrsmDesc d = descs[w.desc];
    constexpr int LDL = TRSM_NB + 2;
    __shared__ double Ls[LDL * TRSM_NB];
    __shared__ double xs[LDL * 64];       // xs[i + v * LDL]
    __shared__ double dinv[TRSM_NB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = d.n, n16 = (n + 15) & ~15;
    const int v0 = w.chunk * 64;
    const int nv = min(64, d.nvec - v0);
    const int i16 = tid & 15, j16 = tid >> 4;
    const gptr<const double> gL = as_global(d.L);
    const gptr<double> gB = as_global(d.B);
    tile64_load<1>(Ls, LDL, gL, 1, d.ldl, n, n, n16, n16, tid);
    if (tid < n16) dinv[tid] = (tid < n) ? 1.0 / gL[tid + (long long)tid * d.ldl] : 0.0;
    if (d.es == 1) tile64_load<0>(xs, LDL, (gptr<const double>)gB + (long long)v0 * d.vs, 1, d.vs, n, nv, n16, nv, tid);            // vectors are columns
    else tile64_load<0, true>(xs, LDL, (gptr<const double>)gB + (long long)v0 * d.vs, d.es, d.vs, n, nv, n16, nv, tid);              // vectors are rows (vs == 1)
    __syncthreads();
    if (d.trans == 0) lds_trsm<false>(Ls, LDL, dinv, xs, 1, LDL, n, nv, wave, 4, lane);
    else lds_trsm<true>(Ls, LDL, dinv, xs, 1, LDL, n, nv, wave, 4, lane);
    __syncthreads();
    if (d.es == 1) {
        for (int j0 = 0; j0 < nv; j0 += 16)
            for (int i0 = 0; i0 < n; i0 += 16) {
                const int i = i0 + i16, v = j0 + j16;
                if (i < n && v < nv) gB[(long long)(v0 + v) * d.vs + i] = xs[i + v * LDL];
            }
    } else {
        for (int i0 = 0; i0 < n; i0 += 4) {
            const int v = tid & 63, i = i0 + (tid >> 6);
            if (i < n && v < nv) gB[(long long)(v0 + v) * d.vs + (long long)i * d.es] = xs[i + v * LDL];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Inverse of a lower triangular diagonal block (n <= 64): out = L^-1 (lower triangle; the strict upper triangle is written as zero).
// Every 64 x 64 leaf of a large triangular factor is independent of the others, so ONE launch inverts them all; the host then joins
// pairs of inverses by two GEMMs per doubling ([[A,0],[B,C]]^-1 = [[A^-1,0],[-C^-1 B A^-1, C^-1]]) up to blocks of TRSM_IB, and a
// triangular solve with many right-hand sides becomes n / TRSM_IB levels of GEMMs instead of n / 64 levels of substitutions.
// ------------------------------------------------------------------------------------------------
struct TrtriDesc {
    const double *L;
    double *out;
    int ldl, ldo, n, pad;
};
constexpr int TRSM_IB = 512;
constexpr size_t trtri_diag_lds_bytes() { return (size_t)(2 * (TRSM_NB + 2) * TRSM_NB + TRSM_NB) * sizeof(double); }
__global__ __launch_bounds__(256) void k_trtri_diag(const TrtriDesc *__restrict__ descs) {
    extern __shared__ __attribute__((aligned(16))) double tts[];
    const TrtriDesc d = descs[blockIdx.x];
    constexpr int LDL = TRSM_NB + 2;
    double *Ls = tts, *Zs = tts + LDL * TRSM_NB, *dinv = Zs + LDL * TRSM_NB;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = d.n, n16 = (n + 15) & ~15;
    const int i16 = tid & 15, j16 = tid >> 4;
    const gptr<const double> gL = as_global(d.L);
    const gptr<double> gout = as_global(d.out);
    tile64_load<1>(Ls, LDL, gL, 1, d.ldl, n, n, n16, n16, tid);
    for (int j0 = 0; j0 < n16; j0 += 16)
        for (int i0 = 0; i0 < n16; i0 += 16) {
            const int i = i0 + i16, j = j0 + j16;
            Zs[i + j * LDL] = (i == j && i < n) ? 1.0 : 0.0;
        }
    if (tid < n16) dinv[tid] = (tid < n) ? 1.0 / gL[tid + (long long)tid * d.ldl] : 0.0;
    __syncthreads();
    lds_trsm<false>(Ls, LDL, dinv, Zs, 1, LDL, n, n, wave, 4, lane);
    __syncthreads();
    for (int j0 = 0; j0 < n; j0 += 16)
        for (int i0 = 0; i0 < n; i0 += 16) {
            const int i = i0 + i16, j = j0 + j16;
            if (i < n && j < n) gout[i + (long long)j * d.ldo] = (i >= j) ? Zs[i + j * LDL] : 0.0;
        }
}
// rows x cols copy between two column-major arrays (the solved right-hand sides back into the caller's array)
struct Copy2dDesc {
    const double *src;
    double *dst;
    long long lds, ldd;
    int rows, cols;
};
__global__ __launch_bounds__(256) void k_copy2d(const Copy2dDesc *__restrict__ descs) {
    const Copy2dDesc d = descs[blockIdx.y];
    const long long total = (long long)d.rows * d.cols;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long i = e % d.rows, j = e / d.rows;
        d.dst[i + j * d.ldd] = d.src[i + j * d.lds];
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
    constexpr int LDA = POTRF_NB + 2;
    __shared__ double As[LDA * POTRF_NB];
    __shared__ double dinv[POTRF_NB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = d.n, n16 = (n + 15) & ~15;
    const int i16 = tid & 15, j16 = tid >> 4;
    const gptr<double> gA = as_global(d.A);
    tile64_load<2>(As, LDA, (gptr<const double>)gA, 1, d.lda, n, n, n16, n16, tid);
    __syncthreads();
    const bool bad = lds_potrf(As, LDA, dinv, n, wave, 4, lane);
    if (bad && lane == 0) atomicMin(info, d.code);
    __syncthreads();
    for (int j0 = 0; j0 < n; j0 += 16)
        for (int i0 = 0; i0 < n; i0 += 16) {
            const int i = i0 + i16, j = j0 + j16;
            if (i < n && j < n && i >= j) gA[i + (long long)j * d.lda] = As[i + j * LDA];
        }
}

// ------------------------------------------------------------------------------------------------
// Schur gather:  S_j[p,q] = sum_{l low rank} sum_{t1 in p, t2 in q} lam1 lam2 GX_l[L1,R2] GY_l[L2,R1]
//                         + sum_{l dense} Sd_l[inv_l[q], inv_l[p]]              for p <= q, mirrored.
// One thread per (p,q); each S entry is written exactly once (deterministic, no atomics).
// ------------------------------------------------------------------------------------------------
struct SBlockDesc {
    int kind, ldg, cnt, tri;   // tri: GX and GY (low rank: W = V, one sub-block) / Sd (dense) are symmetric and only their lower tiles were formed
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

// SG_W lanes per entry, each walking every SG_W-th block of the cluster (a cluster of an SDPA-type problem has tens of blocks, and a
// thread's walk over them is a chain of dependent loads), joined in a fixed order by shuffles; SG_W = 1 when no cluster has four blocks
template <int SG_W>
__global__ __launch_bounds__(256 * SG_W) void k_schur_gather(const SClusterDesc *__restrict__ cl, const SBlockDesc *__restrict__ bl,
                                                             const STile *__restrict__ tiles) {
    const STile t = tiles[blockIdx.x];
    const SClusterDesc c = cl[t.cluster];
    const int ent = threadIdx.x / SG_W, sub = threadIdx.x % SG_W;
    const int p = t.ti * 16 + (ent & 15), q = t.tj * 16 + (ent >> 4);
    if (p >= c.P || q >= c.P || p > q) return;             // uniform over the SG_W lanes of an entry
    double acc = 0.0;
    // a cluster of dense blocks only (an SDPA-type problem: tens of blocks per entry): four blocks of a lane per pass, the four
    // descriptors, then the eight index look-ups, then the four values in flight together instead of three dependent loads per block
    // one block after the other.  Same order of the additions.
    int b = c.b0 + sub;
    for (; b + 3 * SG_W < c.b1; b += 4 * SG_W) {
        SBlockDesc d4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) d4[u] = bl[b + u * SG_W];
        if (d4[0].kind == 0 || d4[1].kind == 0 || d4[2].kind == 0 || d4[3].kind == 0) break;          // a low-rank block: one by one below
        int i4[4], k4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { i4[u] = as_global(d4[u].inv)[p]; k4[u] = as_global(d4[u].inv)[q]; }
        double v4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = max(i4[u], 0), k = max(k4[u], 0);
            const gptr<const double> Sd = as_global(d4[u].Sd);
            v4[u] = d4[u].tri && k < i ? Sd[i + (long long)k * d4[u].cnt] : Sd[k + (long long)i * d4[u].cnt];
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (i4[u] >= 0 && k4[u] >= 0) acc += v4[u];
    }
    for (; b < c.b1; b += SG_W) {
        const SBlockDesc d = bl[b];
        if (d.kind == 0) {
            const int a0 = d.tptr[p], a1 = d.tptr[p + 1], b0 = d.tptr[q], b1 = d.tptr[q + 1];
            for (int t1 = a0; t1 < a1; t1++) {
                const int L1 = d.tL[t1], R1 = d.tR[t1];
                const double l1 = d.tlam[t1];
                for (int t2 = b0; t2 < b1; t2++) {
                    const int L2 = d.tL[t2], R2 = d.tR[t2];
                    const bool sx = d.tri && L1 < R2, sy = d.tri && L2 < R1;
                    acc += (l1 * d.tlam[t2]) * (d.GX[(sx ? R2 : L1) + (long long)(sx ? L1 : R2) * d.ldg] * d.GY[(sy ? R1 : L2) + (long long)(sy ? L2 : R1) * d.ldg]);
                }
            }
        } else {
            const int i = d.inv[p], k = d.inv[q];
            if (i >= 0 && k >= 0) acc += d.tri && k < i ? d.Sd[i + (long long)k * d.cnt] : d.Sd[k + (long long)i * d.cnt];
        }
    }
#pragma unroll
    for (int off = SG_W / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (sub == 0) {
        const gptr<double> S = as_global(c.S);
        S[p + (long long)q * c.P] = acc;
        S[q + (long long)p * c.P] = acc;
    }
}

// Q[k,l] = sum_i LB[i,k] LB[i,l] for a handful of free variables (N <= 16) and many rows: one workgroup per entry,
// fixed-shape tree reduction (deterministic).  Replaces a 1-tile GEMM whose K loop would run serially in one workgroup.
__global__ __launch_bounds__(256) void k_gram_small(const double *__restrict__ LB, int ld, int rows, int N, double *__restrict__ Q) {
    const int k = blockIdx.x % N, l = blockIdx.x / N;
    if (k < l) return;                                   // lower triangle, mirrored below
    const double *a = LB + (long long)k * ld, *b = LB + (long long)l * ld;
    double s = 0.0;
    for (int i = threadIdx.x; i < rows; i += 256) s += a[i] * b[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double v = (part[0] + part[1]) + (part[2] + part[3]);
        Q[k + (long long)l * N] = v;
        Q[l + (long long)k * N] = v;
    }
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

// ------------------------------------------------------------------------------------------------
// Blocked Cholesky of a large matrix (n > 64), ONE launch per block column instead of three (k_potrf_diag, k_trsm_diag, k_gemm_f64_t): what
// bounds the staged regime is the chain of dependent launches, n / 64 levels long (profiles/r02/g_staged_polyopt2048*), not its flops.
// Launch k (k = -1 .. np - 2) finds the panel of block column k final (L_ik, i > k) and does, with c = k + 1:
//   kind 0, one workgroup per block row i >= c ("column" workgroups, the critical path):
//       D = A_cc - L_ck L_ck^T;  T = A_ic - L_ik L_ck^T;  D = chol(D) (every workgroup for itself: no workgroup waits for another);
//       i == c: L_cc -> the side buffer J.D (the others still read the raw A_cc; copied into A after the last launch);
//       i >  c: L_ic = T L_cc^-T as a product with the inverse of L_cc (lds_trsm on the identity) -> A_ic;
//   kind 1, one workgroup per 128 x 128 tile of the rest of the trailing matrix (block columns >= c + 1, lower tiles only):
//       A_ij -= L_ik L_jk^T, both panels of the tile loaded at once (K = 64: one round trip to memory, no k-loop).
// All operand tiles are column-major in LDS with leading dimension CL_LD.
// ------------------------------------------------------------------------------------------------
struct CholLevelJob {
    double *A, *D;               // matrix (lower triangle), side buffer of np diagonal factors (64 x 64 each, ld 64)
    int lda, n, k, code;
    int bulk0, pad;              // first block row / column of the 128 x 128 trailing tiles: k + 2, or k + 1 in the closing launch of a
                                 // factorisation that stops early (no column workgroups left to take block column k + 1)
};
struct CholLevelWork { int job, kind, ti, tj; };
constexpr int CL_LD = 136, CL_TS = CL_LD * 64;       // one 128 x 64 panel in LDS
constexpr int CL_NT = 1024, CL_WAVES = CL_NT / 64;   // sixteen waves: the update, the panel solves and the trailing tiles of the diagonal block spread over all of them
constexpr int CL_WR = CL_WAVES / 4, CL_RPW = 128 / CL_WR, CL_NA = CL_RPW / 16;   // waves along the 128 rows of a panel (four groups along its columns), rows and 16-row pieces per wave
constexpr size_t chol_level_lds_bytes() { return (size_t)(2 * CL_TS + 64) * sizeof(double); }

// acc[a][b] = sum_k X[i][k] Y[j][k], k < 64, over NA x NB subtiles of 16 x 16: rows i of X from X[0], rows j of Y from Y[0]; the lane
// holds i = 16 a + (lane & 15), j = 16 b + (lane >> 4) + 4 reg
template <int NA, int NB>
__device__ __forceinline__ void cl_mm(const double *X, const double *Y, v4d (&acc)[NA][NB], int l15, int l4) {
#pragma unroll
    for (int a = 0; a < NA; a++)
#pragma unroll
        for (int b = 0; b < NB; b++) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 64; kk += 4) {
        double av[NA], bv[NB];
#pragma unroll
        for (int a = 0; a < NA; a++) av[a] = X[(a * 16 + l15) + (kk + l4) * CL_LD];
#pragma unroll
        for (int b = 0; b < NB; b++) bv[b] = Y[(b * 16 + l15) + (kk + l4) * CL_LD];
#pragma unroll
        for (int a = 0; a < NA; a++)
#pragma unroll
            for (int b = 0; b < NB; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[b], av[a], acc[a][b], 0, 0, 0);
    }
}
// rows x cols (<= 128 x 64) of a column-major array into a panel, zero filled.  The addresses are clamped into the valid part instead
// of guarded: a guarded load is a branch, and the loads of a thread then wait for each other (15 us for four 64 x 64 tiles against 3).
__device__ __forceinline__ void cl_load(double *dst, gptr<const double> src, long long ld, int rows, int cols, int tid) {
    if (rows <= 0 || cols <= 0) {
        for (int e = tid; e < 128 * 64; e += CL_NT) dst[(e & 127) + (e >> 7) * CL_LD] = 0.0;
        return;
    }
    constexpr int Q = 128 * 64 / CL_NT;
    double v[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const int e = tid + CL_NT * q, i = e & 127, j = e >> 7;
        v[q] = src[min(i, rows - 1) + min(j, cols - 1) * ld];
    }
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const int e = tid + CL_NT * q, i = e & 127, j = e >> 7;
        dst[i + j * CL_LD] = (i < rows && j < cols) ? v[q] : 0.0;
    }
}
#ifdef CL_STAMPS
#define CL_STAMP(i) do { if (blockIdx.x == 1 && threadIdx.x == 0) g_cl_stamps[i] = wall_clock64(); } while (0)
#else
#define CL_STAMP(i)
#endif
// One matrix (the usual case): the job rides in the kernel arguments and the work item follows from blockIdx.x (column workgroups
// first, then the lower 128 x 128 tiles column by column) -- two dependent trips to memory less on every level; several matrices: tables.
__global__ __launch_bounds__(CL_NT) void k_chol_level(const CholLevelJob J0, const int ncol, const CholLevelJob *__restrict__ jobs,
                                                    const CholLevelWork *__restrict__ work, int *__restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) double cls[];
    CholLevelWork w;
    if (jobs) w = work[blockIdx.x];
    else if ((int)blockIdx.x < ncol) w = CholLevelWork{0, 0, J0.k + 1 + (int)blockIdx.x, 0};
    else {
        const int nt = (J0.n - J0.bulk0 * 64 + 127) / 128;
        int t = (int)blockIdx.x - ncol, tj = 0;
        while (t >= nt - tj) { t -= nt - tj; tj++; }
        w = CholLevelWork{0, 1, tj + t, tj};
    }
    const CholLevelJob J = jobs ? jobs[w.job] : J0;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
    double *P0 = cls, *P1 = cls + CL_TS, *dinv = cls + 2 * CL_TS;
    const long long lda = J.lda;
    const gptr<double> GA = as_global(J.A);
    const gptr<double> GD = as_global(J.D);
    const int rk = J.k * 64, wi = wave % CL_WR, wj = wave / CL_WR;      // wave (wi, wj): rows CL_RPW wi .., one of four column groups
    if (w.kind == 1) {
        const int rb = J.bulk0 * 64, r0 = rb + 128 * w.ti, c0 = rb + 128 * w.tj;
        cl_load(P0, (gptr<const double>)GA + r0 + rk * lda, lda, min(128, J.n - r0), 64, tid);
        if (w.ti != w.tj) cl_load(P1, (gptr<const double>)GA + c0 + rk * lda, lda, min(128, J.n - c0), 64, tid);
        __syncthreads();
        if (w.ti == w.tj && (wi + 1) * CL_RPW <= wj * 32) return;          // strictly above the diagonal of a diagonal tile
        v4d acc[CL_NA][2];
        cl_mm<CL_NA, 2>(P0 + wi * CL_RPW, (w.ti != w.tj ? P1 : P0) + wj * 32, acc, l15, l4);
#pragma unroll
        for (int a = 0; a < CL_NA; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int gi = r0 + wi * CL_RPW + a * 16 + l15, gj = c0 + wj * 32 + b * 16 + l4 + 4 * reg;
                    if (gi < J.n && gj < J.n) GA[gi + gj * lda] -= acc[a][b][reg];
                }
        return;
    }
    // column workgroup: the panel [D; T] = [A_cc; A_ic] in P0 (128 x 64), [L_ck; L_ik] in P1
    const int c = J.k + 1, rc = c * 64, mc = min(64, J.n - rc), ri = w.ti * 64, mi = min(64, J.n - ri);
    const bool diag = w.ti == c;
    CL_STAMP(0);
    {
        double v[4096 / CL_NT];
#pragma unroll
        for (int q = 0; q < 4096 / CL_NT; q++) {
            const int e = tid + CL_NT * q, i = e & 63, j = e >> 6;
            v[q] = GA[(rc + min(i, mc - 1)) + (rc + min(j, mc - 1)) * lda];
        }
        double vt[4096 / CL_NT];
        if (!diag) {
#pragma unroll
            for (int q = 0; q < 4096 / CL_NT; q++) {
                const int e = tid + CL_NT * q, i = e & 63, j = e >> 6;
                vt[q] = GA[(ri + min(i, mi - 1)) + (rc + min(j, mc - 1)) * lda];
            }
        }
#pragma unroll
        for (int q = 0; q < 4096 / CL_NT; q++) {
            const int e = tid + CL_NT * q, i = e & 63, j = e >> 6;
            P0[i + j * CL_LD] = (i < mc && j < mc) ? (i >= j ? v[q] : 0.0) : (i == j ? 1.0 : 0.0);
            P0[64 + i + j * CL_LD] = (!diag && i < mi && j < mc) ? vt[q] : 0.0;
        }
    }
    if (J.k >= 0) {
        double v[4096 / CL_NT], vt[4096 / CL_NT];
#pragma unroll
        for (int q = 0; q < 4096 / CL_NT; q++) {
            const int e = tid + CL_NT * q, i = e & 63, j = e >> 6;
            v[q] = GA[(rc + min(i, mc - 1)) + (rk + j) * lda];
        }
        if (!diag) {
#pragma unroll
            for (int q = 0; q < 4096 / CL_NT; q++) {
                const int e = tid + CL_NT * q, i = e & 63, j = e >> 6;
                vt[q] = GA[(ri + min(i, mi - 1)) + (rk + j) * lda];
            }
        }
#pragma unroll
        for (int q = 0; q < 4096 / CL_NT; q++) {
            const int e = tid + CL_NT * q, i = e & 63, j = e >> 6;
            P1[i + j * CL_LD] = i < mc ? v[q] : 0.0;
            P1[64 + i + j * CL_LD] = (!diag && i < mi) ? vt[q] : 0.0;
        }
    }
    __syncthreads();
    CL_STAMP(1);
    if (J.k >= 0 && !(diag && wi * CL_RPW >= 64)) {
        // [D; T] -= [L_ck; L_ik] L_ck^T: wave (wi, wj) rows CL_RPW wi .., columns 16 wj ..
        v4d acc[CL_NA][1];
        cl_mm<CL_NA, 1>(P1 + wi * CL_RPW, P1 + wj * 16, acc, l15, l4);
#pragma unroll
        for (int a = 0; a < CL_NA; a++)
#pragma unroll
            for (int b = 0; b < 1; b++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int i = wi * CL_RPW + a * 16 + l15, j = wj * 16 + b * 16 + l4 + 4 * reg;
                    if (i >= j) P0[i + j * CL_LD] -= acc[a][b][reg];        // rows >= 64 (T) always, D in its lower triangle
                }
    }
    __syncthreads();
    CL_STAMP(2);
    // chol(D), and T L^-T in the same sweep (the rows of T ride along as rows below every diagonal block)
    const bool bad = lds_potrf(P0, CL_LD, dinv, mc, wave, CL_WAVES, lane, diag ? 0 : 64);
    if (diag && bad && lane == 0) atomicMin(info, J.code);
    __syncthreads();
    CL_STAMP(3);
    if (diag) {
        const gptr<double> out = GD + (long long)c * 4096;
        for (int e = tid; e < 64 * 64; e += CL_NT) {
            const int i = e & 63, j = e >> 6;
            out[e] = (i >= j && i < mc) ? P0[i + j * CL_LD] : 0.0;
        }
        return;
    }
    for (int e = tid; e < 64 * 64; e += CL_NT) {
        const int i = e & 63, j = e >> 6;
        if (i < mi && j < mc) GA[(ri + i) + (rc + j) * lda] = P0[64 + i + j * CL_LD];
    }
    CL_STAMP(4);
}

// ------------------------------------------------------------------------------------------------
// Factorisation stage of ONE large cluster with free variables as ONE blocked factorisation (src/solver.jl:1245-1269): with B^T
// appended to S as one more block row,
//        [ S    .  ]   [ L        .  ] [ L^T  L^-1 B ]
//        [ B^T  0  ] = [ B^T L^-T  I ] [ .    -Q     ],      Q = (L^-1 B)^T (L^-1 B),
// the column workgroups of k_chol_level return B^T L^-T = (L^-1 B)^T as that row's panels and its trailing tiles leave -Q in the
// corner: L^-1 B (a chain of its own of 18 launches through inverted diagonal blocks) and the Gram product cost nothing.  The
// factorisation stops before the corner (PotrfJob::stop).  k_chol_pack lays the augmented matrix out (rows P .. P64 - 1 pad S to a
// multiple of the block size with an identity), k_chol_unpack hands L, L^-1 B and Q back in the layouts of the rest of the library.
// ------------------------------------------------------------------------------------------------
struct CholAugDesc {
    double *S, *LB, *Q, *Aug;     // S: P x P (ld P), in: S_j, out: L_j;  LB: P x N (ld ldb);  Q: N x N;  Aug: (P64 + N)^2 (ld P64 + N)
    const double *B;              // P x N (ld ldb)
    int P, P64, N, ldb;
};
__global__ __launch_bounds__(256) void k_chol_pack(const CholAugDesc d) {
    const gptr<const double> S = as_global((const double *)d.S), B = as_global(d.B);
    const gptr<double> A = as_global(d.Aug);
    const long long na = d.P64 + d.N, total = na * na;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int i = (int)(e % na), j = (int)(e / na);
        if (i < j) continue;
        double v = 0.0;
        if (j < d.P) v = i < d.P ? S[i + (long long)j * d.P] : (i >= d.P64 ? B[j + (long long)(i - d.P64) * d.ldb] : 0.0);
        else if (j < d.P64) v = i == j ? 1.0 : 0.0;
        A[e] = v;
    }
}
__global__ __launch_bounds__(256) void k_chol_unpack(const CholAugDesc d) {
    const gptr<const double> A = as_global((const double *)d.Aug);
    const gptr<double> S = as_global(d.S), LB = as_global(d.LB), Q = as_global(d.Q);
    const long long na = d.P64 + d.N, PP = (long long)d.P * d.P, total = PP + (long long)d.P * d.N + (long long)d.N * d.N;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        if (e < PP) {
            const int i = (int)(e % d.P), j = (int)(e / d.P);
            S[e] = i >= j ? A[i + j * na] : 0.0;
        } else if (e < PP + (long long)d.P * d.N) {
            const long long f = e - PP;
            const int c = (int)(f % d.P), r = (int)(f / d.P);
            LB[c + (long long)r * d.ldb] = A[(d.P64 + r) + c * na];
        } else {
            const long long f = e - PP - (long long)d.P * d.N;
            const int a = (int)(f % d.N), b = (int)(f / d.N), hi = max(a, b), lo = min(a, b);
            Q[f] = -A[(d.P64 + hi) + (d.P64 + lo) * na];
        }
    }
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
