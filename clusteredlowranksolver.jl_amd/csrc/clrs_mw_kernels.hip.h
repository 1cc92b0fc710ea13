// clrs_mw_kernels.hip.h -- the hot path in multi-word fp64 (K limbs per number, clrs_mw_arith.h): device side.
//
// Same path, same stages as the fp64 kernels: Cholesky of the X blocks (src/solver.jl:388-399),
// Schur assembly (compute_S_integrated!, :1062-1226), Cholesky of S_j / L_j^-1 B_j / Q / Cholesky of Q (:1244-1279) and
// the solve stage (:1527-1582) -- at the working precision the reference runs at (Arb midpoints, prec = 256 by default).
// Arrays of multi-word numbers are PLANAR: limb l of element i of an array of logical length `plane` is p[l*plane + i],
// so that every limb plane is an ordinary column-major fp64 array and loads stay coalesced.
//
// At this precision one multiply-add is 100-300 fp64 instructions, so the kernels are bound by the fp64 pipe and by the
// dependent chains of the factorisations, not by HBM: the design goals are (i) as many independent multiply-adds per
// barrier as the stage has, spread over workgroups where the stage allows (columns of V, entries of the pairing
// matrices, entries of S_j), (ii) dot products through the unnormalised accumulator (one renormalisation per dot
// product, not per term), (iii) one Newton reciprocal square root per pivot, evaluated redundantly by every wave so that
// no broadcast sits on the critical path, and reciprocal diagonals kept for all later substitutions.
#ifndef CLRS_MW_KERNELS_HIP_H
#define CLRS_MW_KERNELS_HIP_H

#include <hip/hip_runtime.h>

#include "clrs_mw_arith.h"

// Template parameters: K = limbs of every computed number, DK = limbs of the problem data (sampled vectors, lambda, dense
// A_p, B; the reference holds them at `prec` bits too, src/interface.jl:1078-1112).  DK = 1 is plain fp64 data.
#define MW_NT 256            // threads per workgroup, every kernel
#define MW_CT 8              // columns of V per workgroup in k_mw_zt
#define MW_INFO_NONE 0x7f7f7f7f

typedef long long mwi64;

struct MwBlk {               // one PSD block (j, l)
    int j, n, kind, delta, U, cnt, P, pad;
    mwi64 xyoff;             // offset in the xy layout
    mwi64 rd_off;            // offset of its reciprocal Cholesky diagonal in xrd (sum of n over earlier blocks)
    mwi64 v_off;             // low rank: V, n x U column-major fp64 (expanded unique vectors)
    mwi64 vrow_off;          // low rank: first nonzero row of each unique vector [U]
    mwi64 z_off;             // Z / T scratch, n x U
    mwi64 g_off;             // GX / GY scratch, U x U
    mwi64 tptr_off;          // CSR over the cluster's constraints into the sorted term arrays [P+1]
    mwi64 a_off;             // dense: stack of A_e, cnt matrices n x n fp64
    mwi64 sd_off;            // dense: contribution table cnt x cnt
    mwi64 w_off;             // dense: T_e = X^-1 A_e Y, cnt matrices n x n
    mwi64 dmap_off;          // dense: constraint -> entry (or -1) [P]
    mwi64 d0;                // dense: first entry in dense_p
    mwi64 t0;                // low rank: first term (sorted arrays and original order share the range)
    int m, pad2;
};
struct MwClu {               // one cluster j
    int P, b0, b1, lds;      // constraints; block range; 1 = S_j (and B_j) fit in LDS
    mwi64 coff, Soff;
};
struct MwDev {
    int J, N, NB, nlr, ndn, pad0;
    mwi64 xylen, xlen, Slen, T, xrdlen;
    const MwBlk *blk;
    const MwClu *clu;
    const int *lr_list, *dn_list;       // indices of the low-rank / dense blocks
    const double *V;                    // problem data are planar with DK limbs (DK = 1: plain fp64, DK = 2: double-double, ...)
    const int *vrow;
    const int *st_a, *st_b;             // sorted terms: unique-vector index of pointers_left[s][(r,p,k)] / pointers_right[r][(s,p,k)]
    const double *st_lam;
    const int *tptr;
    const int *st_orig, *st_p, *st_war, *st_wac, *st_trl, *st_trd, *st_flag;   // sorted terms, for the iteration around the path (clrs_mw_ipm.hip.h)
    const int *ay_a, *ay_b, *ay_blk;    // original term order: pairing of the term
    const double *dA;
    const int *dmap, *dense_p;
    const double *B;                    // stacked B, xlen x N column-major
    mwi64 Vp, lamp, dAp, Bp;            // plane lengths of the problem data V, st_lam, dA, B (DK limbs each, planar)
    double *Z, *Tm, *GX, *GY, *W, *Sd;  // scratch, planar
    mwi64 zlen, glen, wlen, sdlen;
    double *S, *LB, *Q, *Qs;            // S layout; stacked L^-1 B (xlen x N); Q (N x N); (unused)
    double *xrd, *srd, *qrd;            // reciprocal diagonals of chol(X_b), L_j, L_Q
    double *t, *u, *AY;                 // t = L^-1 rhs_x (xlen); u slabs (J x N); pairings per term
    int *info;                          // [0] factor status, [1] Cholesky-of-X status
};

namespace mwk {
using namespace mwa;

__device__ __forceinline__ void tri_index(int e, int &ii, int &jj) {      // e -> (ii >= jj) of a packed lower triangle
    ii = (int)((__builtin_sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
    while ((ii + 1) * (ii + 2) / 2 <= e) ii++;
    while (ii * (ii + 1) / 2 > e) ii--;
    jj = e - ii * (ii + 1) / 2;
}

// In-place lower Cholesky of the n x n matrix M (planar, leading dimension ld), reciprocal diagonal to rd.
// approx_cholesky! (src/tools.jl:69-107): returns false at the first non-positive pivot (the strict upper triangle is
// left to the caller).  Right-looking: per pivot one Newton rsqrt (every thread, redundantly), the column scaling and
// the rank-1 update of the trailing triangle spread over the workgroup; two barriers per pivot.
template <int K>
__device__ bool wg_potrf(double *M, long plane, int n, int ld, double *rd, long rdplane, int tid) {
    for (int k = 0; k < n; k++) {
        const long kk = k + (long)k * ld;
        mw<K> d = ld_<K>(M, plane, kk);
        if (!(d.l[0] > 0.0)) return false;
        mw<K> rs = rsqrt<K>(d);
        for (int i = k + 1 + tid; i < n; i += MW_NT) {
            const long idx = i + (long)k * ld;
            st<K>(M, plane, idx, mul<K>(ld_<K>(M, plane, idx), rs));
        }
        __syncthreads();
        if (tid == 0) {
            st<K>(M, plane, kk, sqrt_with_rsqrt<K>(d, rs));
            st<K>(rd, rdplane, k, rs);
        }
        const int m = n - k - 1, cnt = m * (m + 1) / 2;
        for (int e = tid; e < cnt; e += MW_NT) {
            int ii, jj;
            tri_index(e, ii, jj);
            const int i = k + 1 + ii, j = k + 1 + jj;
            const long idx = i + (long)j * ld;
            st<K>(M, plane, idx, fnma<K>(ld_<K>(M, plane, idx), ld_<K>(M, plane, i + (long)k * ld), ld_<K>(M, plane, j + (long)k * ld)));
        }
        __syncthreads();
    }
    return true;
}

// B <- L^-1 B (n x nrhs, planar), L lower with reciprocal diagonal rd
template <int K>
__device__ void wg_trsm_lower(const double *L, long lplane, int ldl, const double *rd, long rdplane, int n, double *B, long bplane,
                              int ldb, int nrhs, int tid) {
    for (int k = 0; k < n; k++) {
        mw<K> r = ld_<K>(rd, rdplane, k);
        for (int c = tid; c < nrhs; c += MW_NT) {
            const long idx = k + (long)c * ldb;
            st<K>(B, bplane, idx, mul<K>(ld_<K>(B, bplane, idx), r));
        }
        __syncthreads();
        const int m = n - k - 1;
        for (int e = tid; e < m * nrhs; e += MW_NT) {
            const int i = k + 1 + e % m, c = e / m;
            const long idx = i + (long)c * ldb;
            st<K>(B, bplane, idx, fnma<K>(ld_<K>(B, bplane, idx), ld_<K>(L, lplane, i + (long)k * ldl), ld_<K>(B, bplane, k + (long)c * ldb)));
        }
        __syncthreads();
    }
}
// B <- L^-T B
template <int K>
__device__ void wg_trsm_lower_t(const double *L, long lplane, int ldl, const double *rd, long rdplane, int n, double *B, long bplane,
                                int ldb, int nrhs, int tid) {
    for (int k = n - 1; k >= 0; k--) {
        mw<K> r = ld_<K>(rd, rdplane, k);
        for (int c = tid; c < nrhs; c += MW_NT) {
            const long idx = k + (long)c * ldb;
            st<K>(B, bplane, idx, mul<K>(ld_<K>(B, bplane, idx), r));
        }
        __syncthreads();
        for (int e = tid; e < k * nrhs; e += MW_NT) {
            const int i = e % k, c = e / k;
            const long idx = i + (long)c * ldb;
            st<K>(B, bplane, idx, fnma<K>(ld_<K>(B, bplane, idx), ld_<K>(L, lplane, k + (long)i * ldl), ld_<K>(B, bplane, k + (long)c * ldb)));
        }
        __syncthreads();
    }
}

// copy an n x n (or rows x cols) planar matrix between a strided global array and a packed LDS array
template <int K>
__device__ void wg_copy(double *dst, long dplane, int ldd, const double *src, long splane, int lds_, int rows, int cols, int tid) {
    for (int e = tid; e < rows * cols; e += MW_NT) {
        const int i = e % rows, c = e / rows;
#pragma unroll
        for (int l = 0; l < K; l++) dst[(long)l * dplane + i + (long)c * ldd] = src[(long)l * splane + i + (long)c * lds_];
    }
}

}  // namespace mwk

extern __shared__ double mw_lds[];

// ---------------------------------------------------------------------------------------------------------------------
// Cholesky of every X block: Xchol_b = chol(X_b), strict upper zero, reciprocal diagonal kept.
// One workgroup per block; `lds` = 1: the block is factored in LDS.
// ---------------------------------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(MW_NT) void k_mw_potrf_x(const MwDev q, const double *__restrict__ X, double *__restrict__ Xc, int lds) {
    using namespace mwk;
    const MwBlk &k = q.blk[blockIdx.x];
    const int n = k.n, tid = threadIdx.x;
    double *M;
    long plane;
    if (lds) {
        M = mw_lds;
        plane = (long)n * n;
        wg_copy<K>(M, plane, n, X + k.xyoff, q.xylen, n, n, n, tid);
    } else {
        M = Xc + k.xyoff;
        plane = q.xylen;
        wg_copy<K>(M, plane, n, X + k.xyoff, q.xylen, n, n, n, tid);
    }
    __syncthreads();
    const bool ok = wg_potrf<K>(M, plane, n, n, q.xrd + k.rd_off, q.xrdlen, tid);
    if (!ok && tid == 0) atomicMin(&q.info[1], (int)blockIdx.x + 1);
    __syncthreads();
    for (int e = tid; e < n * n; e += MW_NT) {
        const int i = e % n, c = e / n;
#pragma unroll
        for (int l = 0; l < K; l++) Xc[(long)l * q.xylen + k.xyoff + e] = (i >= c) ? M[(long)l * plane + e] : 0.0;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Per low-rank block and per tile of MW_CT unique vectors:  T = Y V  and  Z = L^-1 V  (L = chol X_b).
// These are the reference's part_r products (src/solver.jl:1125, 1137) with X^-1 = L^-T L^-1 split over the two sides
// of the pairing: V^T X^-1 V = Z^T Z, so the explicit inverse (inv_cho_precomp!, :1117) is never formed.
// ---------------------------------------------------------------------------------------------------------------------
template <int K, int DK>
__global__ __launch_bounds__(MW_NT) void k_mw_zt(const MwDev q, const double *__restrict__ Xc, const double *__restrict__ Y, int lds_L) {
    using namespace mwk;
    const MwBlk &k = q.blk[q.lr_list[blockIdx.y]];
    const int n = k.n, U = k.U, tid = threadIdx.x, dl = k.delta;
    const int c0 = blockIdx.x * MW_CT;
    if (c0 >= U) return;
    const int nc = min(MW_CT, U - c0);
    const double *V = q.V + k.v_off;
    const int *vrow = q.vrow + k.vrow_off;
    // T[:, c] = Y[:, rows(c)] V[rows(c), c]
    for (int e = tid; e < n * nc; e += MW_NT) {
        const int i = e % n, c = c0 + e / n, r0 = vrow[c];
        acc<K> s;
        acc_zero<K>(s);
        for (int kk = r0; kk < r0 + dl; kk++) acc_fma<K, K, DK>(s, ld_<K>(Y + k.xyoff, q.xylen, i + (long)kk * n), ld_<DK>(V, q.Vp, kk + (long)c * n));
        st<K>(q.Tm + k.z_off, q.zlen, i + (long)c * n, acc_result<K>(s));
    }
    // Z tile in LDS: forward substitution with L
    double *Zt = mw_lds;
    const long zp = (long)n * MW_CT;
    const double *L = Xc + k.xyoff;
    long lplane = q.xylen;
    if (lds_L) {
        double *Ls = mw_lds + (long)K * zp;
        wg_copy<K>(Ls, (long)n * n, n, Xc + k.xyoff, q.xylen, n, n, n, tid);
        L = Ls;
        lplane = (long)n * n;
    }
    for (int e = tid; e < n * nc; e += MW_NT) {
        const int i = e % n, c = e / n;
#pragma unroll
        for (int l = 0; l < K; l++) Zt[(long)l * zp + e] = l < DK ? V[(long)l * q.Vp + i + (long)(c0 + c) * n] : 0.0;
    }
    __syncthreads();
    wg_trsm_lower<K>(L, lplane, n, q.xrd + k.rd_off, q.xrdlen, n, Zt, zp, n, nc, tid);
    for (int e = tid; e < n * nc; e += MW_NT) {
#pragma unroll
        for (int l = 0; l < K; l++) q.Z[(long)l * q.zlen + k.z_off + (long)c0 * n + e] = Zt[(long)l * zp + e];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Pairing matrices of a low-rank block: GX = Z^T Z = V^T X^-1 V, GY = V^T T = V^T Y V (U x U, symmetric; the
// reference's bilinear_pairings_Xinv / _Y, src/solver.jl:1131, 1143).  One thread per entry of the lower triangle.
// ---------------------------------------------------------------------------------------------------------------------
template <int K, int DK>
__global__ __launch_bounds__(MW_NT) void k_mw_gram(const MwDev q) {
    using namespace mwk;
    const MwBlk &k = q.blk[q.lr_list[blockIdx.y]];
    const int n = k.n, U = k.U, dl = k.delta;
    const int e = blockIdx.x * MW_NT + threadIdx.x;
    if (e >= U * (U + 1) / 2) return;
    int a, b;
    tri_index(e, a, b);
    const double *V = q.V + k.v_off;
    const int *vrow = q.vrow + k.vrow_off;
    const double *Z = q.Z + k.z_off, *T = q.Tm + k.z_off;
    acc<K> s;
    acc_zero<K>(s);
    // rows above the first nonzero row of either vector are zero in Z = L^-1 V
    const int i0 = max(vrow[a], vrow[b]);
    for (int i = i0; i < n; i++) acc_fma<K, K, K>(s, ld_<K>(Z, q.zlen, i + (long)a * n), ld_<K>(Z, q.zlen, i + (long)b * n));
    mw<K> gx = acc_result<K>(s);
    acc_zero<K>(s);
    const int r0 = vrow[a];
    for (int i = r0; i < r0 + dl; i++) acc_fma<K, K, DK>(s, ld_<K>(T, q.zlen, i + (long)b * n), ld_<DK>(V, q.Vp, i + (long)a * n));
    mw<K> gy = acc_result<K>(s);
    st<K>(q.GX + k.g_off, q.glen, a + (long)b * U, gx);
    st<K>(q.GX + k.g_off, q.glen, b + (long)a * U, gx);
    st<K>(q.GY + k.g_off, q.glen, a + (long)b * U, gy);
    st<K>(q.GY + k.g_off, q.glen, b + (long)a * U, gy);
}

// ---------------------------------------------------------------------------------------------------------------------
// Dense ("high rank") block: T_e = X^-1 A_e Y for every matrix of the block, then the table Sd[e, e'] = <A_e', T_e>
// (src/solver.jl:1089-1104).  One workgroup per block; n = 1 blocks take one thread per entry.
// ---------------------------------------------------------------------------------------------------------------------
template <int K, int DK>
__global__ __launch_bounds__(MW_NT) void k_mw_dense(const MwDev q, const double *__restrict__ Xc, const double *__restrict__ Y) {
    using namespace mwk;
    const MwBlk &k = q.blk[q.dn_list[blockIdx.x]];
    const int n = k.n, cnt = k.cnt, tid = threadIdx.x;
    const double *A = q.dA + k.a_off;
    double *W = q.W + k.w_off;
    const long nn = (long)n * n;
    if (n == 1) {
        mw<K> rd = ld_<K>(q.xrd + k.rd_off, q.xrdlen, 0);
        mw<K> yx = mul<K>(ld_<K>(Y + k.xyoff, q.xylen, 0), mul<K>(rd, rd));      // Y / X
        for (int e = tid; e < cnt; e += MW_NT) st<K>(W, q.wlen, e, mulx<K, K, DK>(yx, ld_<DK>(A, q.dAp, e)));
    } else {
        for (int e = 0; e < cnt; e++) {
            // M = A_e; M <- L^-1 M; M <- L^-T M; T_e = M Y
            double *M = mw_lds;
            for (int i = tid; i < nn; i += MW_NT) {
#pragma unroll
                for (int l = 0; l < K; l++) M[(long)l * nn + i] = l < DK ? A[(long)l * q.dAp + (long)e * nn + i] : 0.0;
            }
            __syncthreads();
            wg_trsm_lower<K>(Xc + k.xyoff, q.xylen, n, q.xrd + k.rd_off, q.xrdlen, n, M, nn, n, n, tid);
            wg_trsm_lower_t<K>(Xc + k.xyoff, q.xylen, n, q.xrd + k.rd_off, q.xrdlen, n, M, nn, n, n, tid);
            for (int o = tid; o < nn; o += MW_NT) {
                const int i = o % n, c = o / n;
                acc<K> s;
                acc_zero<K>(s);
                for (int kk = 0; kk < n; kk++) acc_fma<K, K, K>(s, ld_<K>(M, nn, i + (long)kk * n), ld_<K>(Y + k.xyoff, q.xylen, kk + (long)c * n));
                st<K>(W, q.wlen, (long)e * nn + o, acc_result<K>(s));
            }
            __syncthreads();
        }
    }
    __syncthreads();
    // Sd[e, e'] = <A_e', T_e>, e <= e' computed, mirrored
    for (int o = tid; o < cnt * (cnt + 1) / 2; o += MW_NT) {
        int e2, e1;
        tri_index(o, e2, e1);       // e2 >= e1
        acc<K> s;
        acc_zero<K>(s);
        for (long i = 0; i < nn; i++) acc_fma<K, K, DK>(s, ld_<K>(W, q.wlen, (long)e1 * nn + i), ld_<DK>(A, q.dAp, (long)e2 * nn + i));
        mw<K> v = acc_result<K>(s);
        st<K>(q.Sd + k.sd_off, q.sdlen, e1 + (long)e2 * cnt, v);
        st<K>(q.Sd + k.sd_off, q.sdlen, e2 + (long)e1 * cnt, v);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// S_j[p, q] = sum over the blocks of the cluster: low rank  sum_{t in p, t' in q} lam_t lam_t' GX[a_t, b_t'] GY[a_t', b_t]
// with a_t = pointers_left[s][(r,p,k)], b_t = pointers_right[r][(s,p,k)] of the term t = (p,r,s,k) (src/solver.jl:1198-1203)
// (the accumulation loops src/solver.jl:1176-1212 with the four Dict lookups replaced by the sorted term table),
// dense  Sd[e_p, e_q].  One thread per entry q >= p, mirrored write (symmetric!, src/tools.jl:43-57).
// ---------------------------------------------------------------------------------------------------------------------
template <int K, int DK>
__global__ __launch_bounds__(MW_NT) void k_mw_saccum(const MwDev q) {
    using namespace mwk;
    const MwClu &c = q.clu[blockIdx.y];
    const int P = c.P;
    const int e = blockIdx.x * MW_NT + threadIdx.x;
    if (e >= P * (P + 1) / 2) return;
    int qq, pp;
    tri_index(e, qq, pp);           // qq >= pp
    acc<K> s;
    acc_zero<K>(s);
    for (int b = c.b0; b < c.b1; b++) {
        const MwBlk &k = q.blk[b];
        if (k.kind == 0) {
            const int *tp = q.tptr + k.tptr_off;
            const int U = k.U;
            for (int t = tp[pp]; t < tp[pp + 1]; t++) {
                for (int t2 = tp[qq]; t2 < tp[qq + 1]; t2++) {
                    mw<K> gx = ld_<K>(q.GX + k.g_off, q.glen, q.st_a[t] + (long)q.st_b[t2] * U);
                    mw<K> gy = ld_<K>(q.GY + k.g_off, q.glen, q.st_a[t2] + (long)q.st_b[t] * U);
                    mw<K> w = mul<K>(gx, gy);
                    constexpr int LL = (2 * DK + 1 < K) ? 2 * DK + 1 : K;   // lambda lambda' is exact in 2 DK limbs; one more bin so that none of them rounds
                    mw<LL> ll = mulx<LL, DK, DK>(ld_<DK>(q.st_lam, q.lamp, t), ld_<DK>(q.st_lam, q.lamp, t2));
                    acc_fma<K, K, LL>(s, w, ll);
                }
            }
        } else {
            const int *dm = q.dmap + k.dmap_off;
            const int e1 = dm[pp], e2 = dm[qq];
            if (e1 >= 0 && e2 >= 0) acc_add<K, K>(s, ld_<K>(q.Sd + k.sd_off, q.sdlen, e1 + (long)e2 * k.cnt));
        }
    }
    mw<K> v = acc_result<K>(s);
    st<K>(q.S + c.Soff, q.Slen, pp + (long)qq * P, v);
    st<K>(q.S + c.Soff, q.Slen, qq + (long)pp * P, v);
}

// A_Y per term: w^T Y v (src/solver.jl:1152-1170)
template <int K>
__global__ __launch_bounds__(MW_NT) void k_mw_ay(const MwDev q) {
    using namespace mwk;
    const mwi64 t = (mwi64)blockIdx.x * MW_NT + threadIdx.x;
    if (t >= q.T) return;
    const int b = q.ay_blk[t];
    if (b < 0) return;
    const MwBlk &k = q.blk[b];
    st<K>(q.AY, q.T, t, ld_<K>(q.GY + k.g_off, q.glen, q.ay_a[t] + (long)q.ay_b[t] * k.U));
}

// ---------------------------------------------------------------------------------------------------------------------
// Factorisation of a cluster: L_j = chol(S_j) (in place in the S buffer), LinvB_j = L_j^-1 B_j.
// ---------------------------------------------------------------------------------------------------------------------
template <int K, int DK>
__global__ __launch_bounds__(MW_NT) void k_mw_factor(const MwDev q) {
    using namespace mwk;
    const int j = blockIdx.x, tid = threadIdx.x;
    const MwClu &c = q.clu[j];
    const int P = c.P, N = q.N;
    double *Sg = q.S + c.Soff;
    double *M, *Bm;
    long mplane, bplane;
    int ldb;
    if (c.lds) {
        M = mw_lds;
        mplane = (long)P * P;
        Bm = mw_lds + (long)K * mplane;
        bplane = (long)P * N;
        ldb = P;
        wg_copy<K>(M, mplane, P, Sg, q.Slen, P, P, P, tid);
    } else {
        M = Sg;
        mplane = q.Slen;
        Bm = q.LB + c.coff;
        bplane = q.xlen * (long)N;
        ldb = (int)q.xlen;
    }
    // B_j (fp64) -> multi-word
    for (int e = tid; e < P * N; e += MW_NT) {
        const int i = e % P, cc = e / P;
#pragma unroll
        for (int l = 0; l < K; l++) Bm[(long)l * bplane + i + (long)cc * ldb] = l < DK ? q.B[(long)l * q.Bp + c.coff + i + (long)cc * q.xlen] : 0.0;
    }
    __syncthreads();
    const bool ok = wg_potrf<K>(M, mplane, P, P, q.srd + c.coff, q.xlen, tid);
    if (!ok) {
        if (tid == 0) atomicMin(&q.info[0], j + 1);
        return;
    }
    if (N > 0) wg_trsm_lower<K>(M, mplane, P, q.srd + c.coff, q.xlen, P, Bm, bplane, ldb, N, tid);
    __syncthreads();
    // L_j back to the S buffer with a zero strict upper triangle; LinvB to the stacked buffer
    for (int e = tid; e < P * P; e += MW_NT) {
        const int i = e % P, cc = e / P;
#pragma unroll
        for (int l = 0; l < K; l++) Sg[(long)l * q.Slen + e] = (i >= cc) ? M[(long)l * mplane + e] : 0.0;
    }
    if (c.lds) wg_copy<K>(q.LB + c.coff, q.xlen * (long)N, (int)q.xlen, Bm, bplane, ldb, P, N, tid);
}

// Q = sum_j LinvB_j^T LinvB_j = LB^T LB over the stacked rows (src/solver.jl:1264-1271): one thread per entry a >= b
template <int K>
__global__ __launch_bounds__(MW_NT) void k_mw_qgram(const MwDev q) {
    using namespace mwk;
    const int N = q.N;
    const int e = blockIdx.x * MW_NT + threadIdx.x;
    if (e >= N * (N + 1) / 2) return;
    int a, b;
    tri_index(e, a, b);
    const long plane = q.xlen * (long)N;
    acc<K> s;
    acc_zero<K>(s);
    for (long r = 0; r < q.xlen; r++) acc_fma<K, K, K>(s, ld_<K>(q.LB, plane, r + a * q.xlen), ld_<K>(q.LB, plane, r + b * q.xlen));
    mw<K> v = acc_result<K>(s);
    st<K>(q.Q, (long)N * N, a + (long)b * N, v);
    st<K>(q.Q, (long)N * N, b + (long)a * N, v);
}

// Cholesky of Q (src/solver.jl:1274)
template <int K>
__global__ __launch_bounds__(MW_NT) void k_mw_potrf_q(const MwDev q, int lds) {
    using namespace mwk;
    const int N = q.N, tid = threadIdx.x;
    if (q.info[0] != MW_INFO_NONE) return;           // a cluster failed: the reference throws before reaching Q
    double *M = q.Q;
    long plane = (long)N * N;
    if (lds) {
        M = mw_lds;
        wg_copy<K>(M, plane, N, q.Q, plane, N, N, N, tid);
        __syncthreads();
    }
    const bool ok = wg_potrf<K>(M, plane, N, N, q.qrd, N, tid);
    if (!ok && tid == 0) atomicMin(&q.info[0], q.J + 1);
    __syncthreads();
    for (int e = tid; e < N * N; e += MW_NT) {
        const int i = e % N, cc = e / N;
#pragma unroll
        for (int l = 0; l < K; l++) q.Q[(long)l * plane + e] = (i >= cc) ? M[(long)l * plane + e] : 0.0;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Solve stage (src/solver.jl:1527-1582), three launches: per cluster t_j = L_j^-1 rhs_x[j], u_j = LinvB_j^T t_j;
// dy = Q^-1 (rhs_y - sum_j u_j); per cluster dx_j = L_j^-T (t_j + LinvB_j dy).
// ---------------------------------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(MW_NT) void k_mw_solve_fwd(const MwDev q, const double *__restrict__ rhs_x) {
    using namespace mwk;
    const int j = blockIdx.x, tid = threadIdx.x;
    const MwClu &c = q.clu[j];
    const int P = c.P, N = q.N;
    const double *L = q.S + c.Soff;
    long lplane = q.Slen;
    double *tv = mw_lds;                     // t: P numbers, planar with plane P
    if (c.lds) {
        double *Ls = mw_lds + (long)K * P;
        wg_copy<K>(Ls, (long)P * P, P, q.S + c.Soff, q.Slen, P, P, P, tid);
        L = Ls;
        lplane = (long)P * P;
    }
    for (int i = tid; i < P; i += MW_NT) {
#pragma unroll
        for (int l = 0; l < K; l++) tv[(long)l * P + i] = rhs_x[(long)l * q.xlen + c.coff + i];
    }
    __syncthreads();
    wg_trsm_lower<K>(L, lplane, P, q.srd + c.coff, q.xlen, P, tv, P, P, 1, tid);
    for (int i = tid; i < P; i += MW_NT) {
#pragma unroll
        for (int l = 0; l < K; l++) q.t[(long)l * q.xlen + c.coff + i] = tv[(long)l * P + i];
    }
    const long plane = q.xlen * (long)N;
    for (int a = tid; a < N; a += MW_NT) {
        acc<K> s;
        acc_zero<K>(s);
        for (int r = 0; r < P; r++) acc_fma<K, K, K>(s, ld_<K>(q.LB, plane, c.coff + r + a * q.xlen), ld_<K>(tv, P, r));
        st<K>(q.u, (long)q.J * N, (long)j * N + a, acc_result<K>(s));
    }
}

template <int K>
__global__ __launch_bounds__(MW_NT) void k_mw_solve_mid(const MwDev q, const double *__restrict__ rhs_y, double *__restrict__ dy, int lds) {
    using namespace mwk;
    const int N = q.N, tid = threadIdx.x;
    double *v = mw_lds;                      // N numbers, plane N
    const double *L = q.Q;
    long lplane = (long)N * N;
    if (lds) {
        double *Ls = mw_lds + (long)K * N;
        wg_copy<K>(Ls, lplane, N, q.Q, lplane, N, N, N, tid);
        L = Ls;
    }
    for (int a = tid; a < N; a += MW_NT) {
        acc<K> s;
        acc_zero<K>(s);
        acc_add<K, K>(s, ld_<K>(rhs_y, N, a));
        for (int j = 0; j < q.J; j++) acc_add<K, K>(s, ld_<K>(q.u, (long)q.J * N, (long)j * N + a), -1.0);
        st<K>(v, N, a, acc_result<K>(s));
    }
    __syncthreads();
    wg_trsm_lower<K>(L, lplane, N, q.qrd, N, N, v, N, N, 1, tid);
    wg_trsm_lower_t<K>(L, lplane, N, q.qrd, N, N, v, N, N, 1, tid);
    for (int a = tid; a < N; a += MW_NT) {
#pragma unroll
        for (int l = 0; l < K; l++) dy[(long)l * N + a] = v[(long)l * N + a];
    }
}

template <int K>
__global__ __launch_bounds__(MW_NT) void k_mw_solve_bwd(const MwDev q, const double *__restrict__ dy, double *__restrict__ dx) {
    using namespace mwk;
    const int j = blockIdx.x, tid = threadIdx.x;
    const MwClu &c = q.clu[j];
    const int P = c.P, N = q.N;
    const double *L = q.S + c.Soff;
    long lplane = q.Slen;
    double *w = mw_lds;
    if (c.lds) {
        double *Ls = mw_lds + (long)K * P;
        wg_copy<K>(Ls, (long)P * P, P, q.S + c.Soff, q.Slen, P, P, P, tid);
        L = Ls;
        lplane = (long)P * P;
    }
    const long plane = q.xlen * (long)N;
    for (int r = tid; r < P; r += MW_NT) {
        acc<K> s;
        acc_zero<K>(s);
        acc_add<K, K>(s, ld_<K>(q.t, q.xlen, c.coff + r));
        for (int a = 0; a < N; a++) acc_fma<K, K, K>(s, ld_<K>(q.LB, plane, c.coff + r + a * q.xlen), ld_<K>(dy, N, a));
        st<K>(w, P, r, acc_result<K>(s));
    }
    __syncthreads();
    wg_trsm_lower_t<K>(L, lplane, P, q.srd + c.coff, q.xlen, P, w, P, P, 1, tid);
    for (int i = tid; i < P; i += MW_NT) {
#pragma unroll
        for (int l = 0; l < K; l++) dx[(long)l * q.xlen + c.coff + i] = w[(long)l * P + i];
    }
}

#endif
