// clrs_wave.hip.h -- wave-level (64 lanes) dense building blocks on LDS-resident fp64 matrices, gfx950.
//
// The matrices of the named configurations are 16..128 wide: too small for tiled BLAS-3 kernels, too serial for
// one-thread-per-column loops through LDS.  These routines keep the dependent chains of the triangular solves and
// of the Cholesky factorisation inside ONE wave, with
//   * one matrix element (or one matrix row) per lane,
//   * DPP row broadcasts (row_newbcast, 16 lanes) to pass the pivot / the finished unknown -- no LDS round trip,
//   * v_mfma_f64_16x16x4_f64 for every 16 x 16 x 16 trailing update,
// and are blocked by 16 so that any n <= 128 is handled by the same code.  All matrices are column-major in LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace clrs {

typedef double v4d_f __attribute__((ext_vector_type(4)));

// broadcast lane K of every 16-lane row to the whole row
template <int K>
__device__ __forceinline__ double bcast16(double v) {
    long long x = __double_as_longlong(v);
    x = __builtin_amdgcn_update_dpp(x, x, 0x150 + K, 0xf, 0xf, false);   // v_mov_b64_dpp row_newbcast:K (one instruction on gfx950)
    return __longlong_as_double(x);
}

// ordering point for the LDS traffic of ONE wave (no other wave involved): LDS operations of a wave complete in issue
// order, this only stops the compiler from moving them across
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// The blocked routines below synchronise the waves that share a matrix: the whole workgroup (WG = true, __syncthreads), or -- when ONE
// wave runs the routine on its own matrix (wave = 0, nwaves = 1) -- only the compiler's ordering of that wave's LDS accesses.
template <bool WG>
__device__ __forceinline__ void lds_sync() {
    if (WG) __syncthreads();
    else wave_sync();
}
// Sum over the 64 lanes, the same value (bitwise) in every lane.  DPP row shifts inside the 16-lane rows (lanes shifted in
// from outside a row read 0), then the four row totals through v_readlane: ~25 short instructions.  (__shfl_xor on a
// double lowers to two ds_bpermute per step: ~700 cycles for the same reduction.)
template <int CTRL>
__device__ __forceinline__ double dpp_mov_zero(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_mov_zero<0x111>(v);   // row_shr:1
    v += dpp_mov_zero<0x112>(v);   // row_shr:2
    v += dpp_mov_zero<0x114>(v);   // row_shr:4
    v += dpp_mov_zero<0x118>(v);   // row_shr:8   -> lane 15 of every row holds the row total
    return (readlane_f64(v, 15) + readlane_f64(v, 31)) + (readlane_f64(v, 47) + readlane_f64(v, 63));
}

// ---- small MFMA GEMM on LDS operands:  C[i,j] = sum_k A[k,i] B[k,j]  (i < M, j < N, k < K) -------------------------
// A: K x M (ld lda), B: K x N (ld ldb), C: M x N (ld ldc).  Rows k >= K of A/B up to ceil4(K) and columns up to
// ceil16(M)/ceil16(N) must be readable and ZERO.  Tiles of 16 x 16 are dealt to the waves of the workgroup.
// lower_only: only tiles with ti >= tj (C symmetric, the caller mirrors).
__device__ __forceinline__ void lds_gemm_tn(const double *A, int lda, const double *B, int ldb, double *C, int ldc, int M, int N, int K,
                                            int wave, int nwaves, int lane, bool lower_only = false) {
    const int tm = (M + 15) >> 4, tn = (N + 15) >> 4, K4 = (K + 3) & ~3;
    const int l15 = lane & 15, l4 = lane >> 4;
    int t = wave, ti = wave, tj = 0;
    while (ti >= tm) { ti -= tm; tj++; }
    for (; t < tm * tn; t += nwaves) {
        if (!lower_only || ti >= tj) {
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
        ti += nwaves;
        while (ti >= tm) { ti -= tm; tj++; }
    }
}

// ---- 16 x 16 triangular solves inside a wave ----------------------------------------------------------------------
// A wave holds 4 right-hand sides x 16 unknowns, one element per lane (row = lane & 15).  Forward (L x = b): step K
// broadcasts the finished x_K and every lane below eliminates it with one FMA.  Lrow[k] = L[row, k], zero above the
// diagonal; dinv = 1 / L[row,row] (0 for padding rows).  Two independent right-hand-side sets are interleaved.
// The lanes carry the UNSCALED partial results y (x = y * dinv): step K broadcasts x_K = y_K * dinv_K and every lane
// subtracts L[row,K] x_K.  Lrow must be STRICTLY lower (the diagonal entry and everything above it zero), so that the
// lanes whose unknown is finished are left alone without a compare/select; the caller multiplies by dinv at the end.
template <int K>
struct Trsm16 {
    static __device__ __forceinline__ void run(double &y0, double &y1, const double (&Lrow)[16], double dinv) {
        const double b0 = bcast16<K>(y0 * dinv), b1 = bcast16<K>(y1 * dinv);
        y0 = __builtin_fma(-Lrow[K], b0, y0);
        y1 = __builtin_fma(-Lrow[K], b1, y1);
        Trsm16<K + 1>::run(y0, y1, Lrow, dinv);
    }
};
template <>
struct Trsm16<15> {     // the last unknown has nothing below it
    static __device__ __forceinline__ void run(double &, double &, const double (&)[16], double) {}
};
// Backward (L^T x = b): Lcol[k] = L[k, row] for k > row (zero otherwise), steps K = 15 .. 1.
template <int K>
struct Trsm16T {
    static __device__ __forceinline__ void run(double &y0, double &y1, const double (&Lcol)[16], double dinv) {
        const double b0 = bcast16<K>(y0 * dinv), b1 = bcast16<K>(y1 * dinv);
        y0 = __builtin_fma(-Lcol[K], b0, y0);
        y1 = __builtin_fma(-Lcol[K], b1, y1);
        Trsm16T<K - 1>::run(y0, y1, Lcol, dinv);
    }
};
template <>
struct Trsm16T<0> {
    static __device__ __forceinline__ void run(double &, double &, const double (&)[16], double) {}
};

// The same substitutions with ONE instruction per elimination: on the row-scaled system (m[k] = -L[row,k] / L[row,row], right-hand
// side scaled by 1 / L[row,row]) a step is y += y_K m[K], and v_fmac_f64_dpp reads y_K from lane K of the 16-lane row inside the
// multiply-add (row_newbcast is the one DPP control the 64-bit ALU has).  Two right-hand-side sets are interleaved; with the s_nop
// an instruction reads a register written three issue slots earlier (a DPP read needs two wait states after the VALU write).
// 15 steps x 2 chains: 230-250 cycles against the 470 of Trsm16 (scripts/micro/micro_dppchain.hip).
#define CLRS_FMAC2(K) "v_fmac_f64_dpp %0, %0, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
__device__ __forceinline__ void trsm16_fmac_fwd(double &y0, double &y1, const double (&m)[16]) {
// (the compiler's hazard recogniser does not look inside inline assembly: every DPP read here is kept two wait states behind the write
// of its source by the s_nop INSIDE the same asm statement, which the scheduler cannot separate from it)
#define CLRS_STEP(K, R) asm volatile("v_fmac_f64_dpp %0, %0, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\ts_nop 0" : "+v"(y0), "+v"(y1) : "v"(m[R]))
#define CLRS_STEP0(K, R) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\ts_nop 0" : "+v"(y0), "+v"(y1) : "v"(m[R]))
    CLRS_STEP0(0, 0); CLRS_STEP(1, 1); CLRS_STEP(2, 2); CLRS_STEP(3, 3); CLRS_STEP(4, 4); CLRS_STEP(5, 5); CLRS_STEP(6, 6); CLRS_STEP(7, 7);
    CLRS_STEP(8, 8); CLRS_STEP(9, 9); CLRS_STEP(10, 10); CLRS_STEP(11, 11); CLRS_STEP(12, 12); CLRS_STEP(13, 13); CLRS_STEP(14, 14);
}
__device__ __forceinline__ void trsm16_fmac_bwd(double &y0, double &y1, const double (&m)[16]) {
    CLRS_STEP0(15, 15); CLRS_STEP(14, 14); CLRS_STEP(13, 13); CLRS_STEP(12, 12); CLRS_STEP(11, 11); CLRS_STEP(10, 10); CLRS_STEP(9, 9); CLRS_STEP(8, 8);
    CLRS_STEP(7, 7); CLRS_STEP(6, 6); CLRS_STEP(5, 5); CLRS_STEP(4, 4); CLRS_STEP(3, 3); CLRS_STEP(2, 2); CLRS_STEP(1, 1);
#undef CLRS_STEP
#undef CLRS_STEP0
}
#undef CLRS_FMAC2

// Z <- L^-1 Z (TRANS = false) or Z <- L^-T Z (TRANS = true), blocked by 16.
//   L: lower triangular, ld ldl, ZERO above the diagonal and in rows/columns n..ceil16(n)-1;
//   dinv[i] = 1 / L[i,i] for i < n, 0 for n <= i < ceil16(n);
//   Z element (row, col) at Z[row * rs + col * cs]; rows n..ceil16(n)-1 must be readable (any finite value) and are
//   only written with values that do not matter.  Must be called by all `nwaves` waves of the workgroup (barriers).
template <bool TRANS, bool WG = true>
__device__ __forceinline__ void lds_trsm(const double *L, int ldl, const double *dinv, double *Z, int rs, int cs, int n, int ncols, int wave,
                                         int nwaves, int lane) {
    const int npan = (n + 15) >> 4, row16 = lane & 15, cg4 = lane >> 4;
    const int ngroups = (ncols + 3) >> 2;
    for (int step = 0; step < npan; step++) {
        const int pb = TRANS ? npan - 1 - step : step;
        const int r0 = pb * 16, row = r0 + row16;
        const double di = dinv[row];
        double Lr[16];
#pragma unroll
        for (int k = 0; k < 16; k++) Lr[k] = TRANS ? L[(r0 + k) + row * ldl] : L[row + (r0 + k) * ldl];
#pragma unroll
        for (int k = 0; k < 16; k++) Lr[k] = (k == row16) ? 0.0 : -(Lr[k] * di);      // strictly triangular (L is zero on the other side already), row scaled
        for (int g = wave; g < ngroups; g += 2 * nwaves) {   // two column groups per pass: independent chains interleave
            const int c0 = g * 4 + cg4, c1 = (g + nwaves) * 4 + cg4;
            const bool v0 = c0 < ncols, v1 = c1 < ncols;
            const double z0 = Z[row * rs + (v0 ? c0 : 0) * cs], z1 = Z[row * rs + (v1 ? c1 : 0) * cs];
            double x0 = v0 ? z0 * di : 0.0, x1 = v1 ? z1 * di : 0.0;
            if (TRANS) trsm16_fmac_bwd(x0, x1, Lr);
            else trsm16_fmac_fwd(x0, x1, Lr);
            if (v0) Z[row * rs + c0 * cs] = x0;
            if (v1) Z[row * rs + c1 * cs] = x1;
        }
        if (step + 1 < npan) {
            lds_sync<WG>();
            // remaining panels:  Z[i-panel, :] -= op(L)[i-panel, r0:r0+16] Z[r0:r0+16, :]   (MFMA, 16 x 16 tiles)
            const int tm = npan - step - 1, tn = (ncols + 15) >> 4;
            int ti = wave, tj = 0;
            while (ti >= tm) { ti -= tm; tj++; }
            for (int t = wave; t < tm * tn; t += nwaves) {
                const int i0 = TRANS ? ti * 16 : r0 + 16 + ti * 16, j0 = tj * 16;
                v4d_f acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 16; kk += 4) {
                    const double zb = (j0 + row16 < ncols) ? Z[(r0 + kk + cg4) * rs + (j0 + row16) * cs] : 0.0;    // a-operand: Z[k, j0 + r]
                    const double la = TRANS ? L[(r0 + kk + cg4) + (i0 + row16) * ldl]  // b-operand: L[k, i0 + c]  (L^T[i0 + c, k])
                                            : L[(i0 + row16) + (r0 + kk + cg4) * ldl]; //            L[i0 + c, k]
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(zb, la, acc, 0, 0, 0);
                }
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int j = j0 + cg4 + 4 * reg;
                    if (j < ncols) Z[(i0 + row16) * rs + j * cs] -= acc[reg];
                }
                ti += nwaves;
                while (ti >= tm) { ti -= tm; tj++; }
            }
            lds_sync<WG>();
        }
    }
}

// ---- 16 x 16 Cholesky inside a wave: lane r (of every 16-lane row, redundantly) holds row r of the block ---------------
// Right-looking: step K takes the square root of the pivot of lane K, scales column K, and eliminates it from the
// trailing columns; l_JK of lane J reaches the others by DPP broadcast.  IEEE sqrt and division (parity with LAPACK-style
// references).  Returns through `bad` whether a pivot of a row < nvalid was not positive (tools.jl:92-95).
// sqrt(a) and 1/sqrt(a) together: v_rsq_f64 seed, two coupled Goldschmidt iterations, one residual correction each.
// 13 dependent fp64 operations instead of the ~25 of an IEEE sqrt followed by an IEEE division; the results are
// within 1 ulp of the correctly rounded values (the parity tests compare against LAPACK at 1e-12 relative).
__device__ __forceinline__ void sqrt_rsqrt(double a, double &d, double &r) {
    const double y = __builtin_amdgcn_rsq(a);
    double g = a * y, h = 0.5 * y;
    double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    h = __builtin_fma(h, e, h);
    e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    h = __builtin_fma(h, e, h);
    const double res = __builtin_fma(-g, g, a);
    g = __builtin_fma(res, h, g);            // g = sqrt(a)
    double rr = h + h;
    const double e2 = __builtin_fma(-g, rr, 1.0);
    rr = __builtin_fma(rr, e2, rr);          // rr = 1 / sqrt(a)
    d = g;
    r = rr;
}

// Square-root-free elimination (L D L^T) with the scaling applied at the end: the dependent chain of a step is
//   broadcast pivot -> reciprocal (v_rcp_f64 + one Newton step) -> one multiply -> the FMA that finishes the next pivot,
// about 70 cycles instead of the 200 of "sqrt, divide, eliminate"; the 16 reciprocal square roots are then computed by the 16
// lanes in parallel (one chain in total) and broadcast.  Column K of the working matrix holds w_iK = l_iK * sqrt(d_K).
template <int K>
struct Potrf16 {
    static __device__ __forceinline__ void run(double (&a)[16], int row, int nvalid, bool &bad, double &dgn) {
        const double p = bcast16<K>(a[K]);               // pivot d_K: the diagonal entry of lane K after the previous eliminations
        if (K < nvalid && !(p > 0.0)) bad = true;
        dgn = (row == K) ? p : dgn;
        double r = __builtin_amdgcn_rcp(p);
        r = __builtin_fma(__builtin_fma(-p, r, 1.0), r, r);
        const double w = a[K];
        const double nt = -(w * r);
        // Entries above the diagonal are never read: the eliminations run unpredicated on all rows; whatever they leave above
        // the diagonal is zeroed by the final scaling pass.
        Elim<K + 1>::run(a, nt, w);
        // the next pivot broadcast is a DPP read of a[K + 1] that the COMPILER emits: it cannot see the VALU write inside the inline
        // assembly above, so the two wait states are put here, tied to that register so that they stay between the two
        if constexpr (K + 1 < 16) asm volatile("s_nop 1" : "+v"(a[K + 1 < 16 ? K + 1 : 15]));
        Potrf16<K + 1>::run(a, row, nvalid, bad, dgn);
    }
    template <int J, int DUMMY = 0>
    struct Elim {
        static __device__ __forceinline__ void run(double (&a)[16], double nt, double w) {
            // a_iJ -= w_iK w_JK / d_K as ONE instruction: w_JK is read from lane J of the row by DPP inside the multiply-add
            // (v_fmac_f64_dpp; row_newbcast is the one DPP control the 64-bit ALU has) instead of a v_mov_b64_dpp + v_fma pair
            asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a[J]) : "v"(w), "v"(nt), "n"(J));
            Elim<J + 1>::run(a, nt, w);
        }
    };
    template <int DUMMY>
    struct Elim<16, DUMMY> {
        static __device__ __forceinline__ void run(double (&)[16], double, double) {}
    };
};
template <>
struct Potrf16<16> {
    static __device__ __forceinline__ void run(double (&)[16], int, int, bool &, double &) {}
};
template <int K>
struct PotrfScale16 {
    static __device__ __forceinline__ void run(double (&a)[16], int row, double rs) {
        const double sK = bcast16<K>(rs);                // 1 / sqrt(d_K) lives in lane K
        a[K] = (row >= K) ? a[K] * sK : 0.0;
        PotrfScale16<K + 1>::run(a, row, rs);
    }
};
template <>
struct PotrfScale16<16> {
    static __device__ __forceinline__ void run(double (&)[16], int, double) {}
};
// lower Cholesky of the 16 x 16 matrix whose row `row` the lane holds in a[] (entries above the diagonal ignored);
// on return a[] is row `row` of L with zeros above the diagonal.
// rdiag receives 1 / L[row,row] (the reciprocal square root of the row's pivot: what the triangular solves need)
__device__ __forceinline__ void potrf16(double (&a)[16], int row, int nvalid, bool &bad, double &rdiag) {
    double dgn = 1.0;
    Potrf16<0>::run(a, row, nvalid, bad, dgn);
    double d, rs;
    sqrt_rsqrt(dgn, d, rs);                              // every lane: 1 / sqrt of its own pivot
    PotrfScale16<0>::run(a, row, rs);
    rdiag = rs;
}
__device__ __forceinline__ void potrf16(double (&a)[16], int row, int nvalid, bool &bad) {
    double rdiag;
    potrf16(a, row, nvalid, bad, rdiag);
}

// In-place lower Cholesky of the n x n matrix A in LDS (ld lda), blocked by 16.  Requirements: rows/columns
// n..ceil16(n)-1 of A hold the identity (diagonal 1, rest 0); only the lower triangle is read.  On return the lower
// triangle holds L, the diagonal blocks have zeros above the diagonal; the strictly upper off-diagonal blocks are
// NOT touched (callers mask them when storing).  dinv (ceil16(n) doubles of LDS) receives 1 / L[i,i] (0 for i >= n).
// Returns true in every thread of wave 0 .. (callers reduce) if a pivot was not positive.  All waves must call.
// extra (a multiple of 16): rows n16 .. n16 + extra - 1 of the same array hold a panel B below the square; on return they hold
// B L^-T (the panel of a blocked factorisation solved in the same sweep: its rows ride along as rows below every diagonal block).
template <bool WG = true>
__device__ __forceinline__ bool lds_potrf(double *A, int lda, double *dinv, int n, int wave, int nwaves, int lane, int extra = 0) {
    const int npan = (n + 15) >> 4, row16 = lane & 15, cg4 = lane >> 4;
    bool bad = false;
    for (int pb = 0; pb < npan; pb++) {
        const int r0 = pb * 16;
        if (wave == 0) {
            double a[16];
#pragma unroll
            for (int c = 0; c < 16; c++) a[c] = A[(r0 + row16) + (r0 + c) * lda];
#pragma unroll
            for (int c = 1; c < 16; c++)
                if (c > row16) a[c] = 0.0;
            double rdiag;
            potrf16(a, row16, n - r0, bad, rdiag);
            if (cg4 == 0) {
#pragma unroll
                for (int c = 0; c < 16; c++) A[(r0 + row16) + (r0 + c) * lda] = a[c];
                dinv[r0 + row16] = (r0 + row16 < n) ? rdiag : 0.0;      // 1 / L[i,i] = the reciprocal square root of the pivot, already at hand
            }
        }
        if (pb + 1 == npan && extra == 0) break;
        lds_sync<WG>();
        // panel: rows below the diagonal block, X L_kk^T = A_panel  <=>  L_kk X^T = A_panel^T: the unknown index runs along
        // the 16 columns of the panel (stride lda), the right-hand sides are the panel rows (stride 1)
        {
            const int M = (npan - pb - 1) * 16 + extra;
            const double di = dinv[r0 + row16];
            double Lr[16];
#pragma unroll
            for (int k = 0; k < 16; k++) Lr[k] = A[(r0 + row16) + (r0 + k) * lda];
#pragma unroll
            for (int k = 0; k < 16; k++) Lr[k] = (k == row16) ? 0.0 : -(Lr[k] * di);
            const int ngroups = M >> 2;
            for (int g = wave; g < ngroups; g += 2 * nwaves) {
                const int c0 = g * 4 + cg4, c1 = (g + nwaves) * 4 + cg4;
                const bool v1 = c1 < M;
                double *p0 = A + (r0 + 16 + c0) + (r0 + row16) * lda, *p1 = A + (r0 + 16 + (v1 ? c1 : c0)) + (r0 + row16) * lda;
                const double z1 = *p1;
                double x0 = *p0 * di, x1 = v1 ? z1 * di : 0.0;
                trsm16_fmac_fwd(x0, x1, Lr);
                *p0 = x0;
                if (v1) *p1 = x1;
            }
        }
        lds_sync<WG>();
        // trailing update (lower tiles): A[i-blk, j-blk] -= Lp_i Lp_j^T
        {
            const int tm = npan - pb - 1, tri = tm * (tm + 1) / 2;
            for (int t = wave; t < tri + (extra >> 4) * tm; t += nwaves) {
                int ti, tj = 0;
                if (t < tri) {
                    int rem = t;
                    while (rem >= tm - tj) { rem -= tm - tj; tj++; }
                    ti = tj + rem;
                } else {                                                 // rows of the extra panel against every remaining column block
                    ti = tm + (t - tri) / tm;
                    tj = (t - tri) % tm;
                }
                const int i0 = r0 + 16 + ti * 16, j0 = r0 + 16 + tj * 16;
                v4d_f acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 16; kk += 4) {
                    const double aj = A[(j0 + row16) + (r0 + kk + cg4) * lda];   // a-operand: Lp[j0 + r, k]
                    const double ai = A[(i0 + row16) + (r0 + kk + cg4) * lda];   // b-operand: Lp[i0 + c, k]
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aj, ai, acc, 0, 0, 0);
                }
#pragma unroll
                for (int reg = 0; reg < 4; reg++) A[(i0 + row16) + (j0 + cg4 + 4 * reg) * lda] -= acc[reg];
            }
        }
        lds_sync<WG>();
    }
    return bad;
}

// ---- triangular solves with ONE right-hand side inside ONE wave (no workgroup barriers) -------------------------------------
// x += bcast_K(x) * m[K], K = 0 .. 14 (forward) / K = 15 .. 1 (backward); a DPP read needs two wait states after the write
__device__ __forceinline__ void trsv16_chain_fwd(double &x, const double (&m)[16]) {
    asm volatile(
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %3 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %6 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %7 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %10 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %11 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %12 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %13 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %14 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %15 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(x)
        : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "v"(m[9]), "v"(m[10]), "v"(m[11]),
          "v"(m[12]), "v"(m[13]), "v"(m[14]));
}
__device__ __forceinline__ void trsv16_chain_bwd(double &x, const double (&m)[16]) {
    asm volatile(
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %15 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %14 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %13 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %12 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %11 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %10 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %9 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %8 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %7 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %6 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(x)
        : "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "v"(m[9]), "v"(m[10]), "v"(m[11]), "v"(m[12]),
          "v"(m[13]), "v"(m[14]), "v"(m[15]));
}

// One wave: x <- L^-1 x.  A: n16 x n16 in LDS (ld lda), lower triangle of L, ZERO above the diagonal and in rows / columns >= n;
// dinv: 1 / L[i,i] (0 for i >= n); x: n16 entries in LDS (entries >= n zero).
__device__ __forceinline__ void wave_trsv_fwd(const double *A, int lda, const double *dinv, double *x, int n16, int lane) {
    const int l15 = lane & 15;
    for (int r0 = 0; r0 < n16; r0 += 16) {
        const int row = r0 + l15;
        const double di = dinv[row];
        double m[16];
#pragma unroll
        for (int k = 0; k < 15; k++) m[k] = A[row + (r0 + k) * lda];            // all reads issued first (a select around a read is a branch)
#pragma unroll
        for (int k = 0; k < 15; k++) m[k] = (k < l15) ? -(m[k] * di) : 0.0;
        double xr = x[row] * di;
        trsv16_chain_fwd(xr, m);
        if (lane < 16) x[row] = xr;
        wave_sync();
        for (int i = r0 + 16 + lane; i < n16; i += 64) {      // rows below the panel
            double s = x[i];
#pragma unroll
            for (int k = 0; k < 16; k++) s = __builtin_fma(-A[i + (r0 + k) * lda], x[r0 + k], s);
            x[i] = s;
        }
        wave_sync();
    }
}
// One wave: x <- L^-T x (same storage of L).
__device__ __forceinline__ void wave_trsv_bwd(const double *A, int lda, const double *dinv, double *x, int n16, int lane) {
    const int l15 = lane & 15;
    for (int r0 = n16 - 16; r0 >= 0; r0 -= 16) {
        const int row = r0 + l15;
        const double di = dinv[row];
        double m[16];
#pragma unroll
        for (int k = 1; k < 16; k++) m[k] = A[(r0 + k) + row * lda];            // L^T[row, r0 + k] = L[r0 + k, row]
#pragma unroll
        for (int k = 1; k < 16; k++) m[k] = (k > l15) ? -(m[k] * di) : 0.0;
        double xr = x[row] * di;
        trsv16_chain_bwd(xr, m);
        if (lane < 16) x[row] = xr;
        wave_sync();
        for (int i = lane; i < r0; i += 64) {                   // rows above the panel
            double s = x[i];
#pragma unroll
            for (int k = 0; k < 16; k++) s = __builtin_fma(-A[(r0 + k) + i * lda], x[r0 + k], s);
            x[i] = s;
        }
        wave_sync();
    }
}


}  // namespace clrs
