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

#include <type_traits>

#include "clrs_mw_arith.h"

// Template parameters: K = limbs of every computed number, DK = limbs of the problem data (sampled vectors, lambda, dense
// A_p, B; the reference holds them at `prec` bits too, src/interface.jl:1078-1112).  DK = 1 is plain fp64 data.
#define MW_NT 256            // threads per workgroup, every kernel but the three factorisations
#define MW_PT 512            // threads per workgroup of k_mw_potrf_x, k_mw_factor, k_mw_potrf_q: one wave on the dependent chain, seven behind it
#define MW_CT 8              // columns of V per workgroup in k_mw_zt
#define MW_INFO_NONE 0x7f7f7f7f
#define MW_INFO_TIMEOUT 0x7f7f7f00   // status word of a launch that waited for a word of another stream (mw_wait_word, k_mwi_wait) longer than MW_WAIT_TICKS: larger than every
                                     // block number, so a real failure of the same decomposition wins the atomicMin; the iteration reports error code 5, not a failed block
#define MW_WAIT_TICKS 3000000000ull  // 30 s of wall_clock64 (100 MHz): a bound against hangs only -- a long wait (a large instance, a GPU shared with other work) is not a failure
#define MW_INV_WG 4          // workgroups that share the columns of an inverse factor (k_mw_factor, k_mw_potrf_q, k_mw_bp_diag)

typedef long long mwi64;

struct MwBlk {               // one PSD block (j, l)
    int j, n, kind, delta, U, cnt, P, inv;   // chol(X_b)^-1 is formed beside the factor (Xi): 1 = in LDS, 2 = in place in memory (the block fits in LDS once, not twice), 0 = not
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
    int P, b0, b1, lds;      // constraints; block range; 1 = S_j and the inverse of its factor fit in LDS side by side (k_mw_factor), 0 = blocked path
    mwi64 coff, Soff;
    int one_term, pad;       // 1 = at most four PSD blocks and at most one low-rank term per (constraint, block): S_j by k_mw_saccum_one
};
// Limbs of the FACTOR stage and of the products of the solve stage when the context runs them in fewer limbs than its K (MwDev::kf < K; clrs_mw_options.factor_limbs):
// mixed-precision iterative refinement.  L_j, L_j^-1, L^-1 B, Q, L_Q, L_Q^-1 and both passes of inverse-factor products carry mw_kf_of(K) limbs (their upper
// planes are stored as zeros, so that every K-limb reader stays valid), the residuals r_x = rhs_x - S dx + B dy, r_y = rhs_y - B^T dx of the refinement step
// and the sum dx + dx' carry K.  One K-1 limb pass has a forward error of cond 2^(-53 (K - 1)); the correction squares it.  The refined solution is as good as
// that of K-limb factors while the first pass alone is good to 2^-53 (one limb): k_mw_solve_bwd MODE 2 measures max|dx'| / max|dx| (MwDev::refstat) and the
// caller returns to K limbs when it is not (prototype numbers: cohnelkies(8,15) iterations 1 / 28 / 55 first pass 2^-123 / 2^-99 / 2^-45).
__host__ __device__ constexpr int mw_kf_of(int K) { return (K == 5 || K == 6) ? K - 1 : K; }

// The reference's `matmul_prec` (src/solver.jl:125, 304, 312-313, 1125-1143): the products that lead to the pairing matrices (part_r = Y V, X^-1 V and
// bilinear_pairings = W^T part_r) at fewer bits than the rest.  Here: T = Y V, Z = chol(X)^-1 V, GX = Z^T Z, GY = V^T T in MwDev::km limbs (stored with the upper
// K - km planes zero; S_j is accumulated from them in K limbs, as the reference does at `prec`).  Limb counts on offer: the instantiated ones from K / 2 up;
// the host rounds a request up to the next of them.
__host__ __device__ constexpr bool mw_km_ok(int K, int KM) { return KM == K || (KM < K && 2 * KM >= K && (KM <= 6 || KM == 8)); }
#define MW_KM_CASE(KMc) if constexpr (mw_km_ok(K, KMc) && KMc < K) { if (km == KMc) { f(std::integral_constant<int, KMc>{}); return; } }
template <int K, class F>
__device__ __forceinline__ void mw_km_switch(int km, F f) {
    MW_KM_CASE(2) MW_KM_CASE(3) MW_KM_CASE(4) MW_KM_CASE(5) MW_KM_CASE(6) MW_KM_CASE(8)
    f(std::integral_constant<int, K>{});
}

struct MwDev {
    int J, N, NB, nlr, ndn, kf;         // kf: limbs of the factor stage and of the solve's products (K, or mw_kf_of(K))
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
    const int *drow_ptr, *drow_blk, *drow_en;   // per constraint row (stacked, xlen + 1): its dense entries as (block, entry) pairs
    int dn_big, maxcnt;                 // some dense block has n > 1 (the dense branch then runs one workgroup per matrix); largest cnt of a dense block
    const double *B;                    // stacked B, xlen x N column-major
    mwi64 Vp, lamp, dAp, Bp;            // plane lengths of the problem data V, st_lam, dA, B (DK limbs each, planar)
    double *Z, *Tm, *GX, *GY, *W, *Sd;  // scratch, planar
    mwi64 zlen, glen, wlen, sdlen;
    double *S, *LB, *Q, *Qs;            // S layout; stacked L^-1 B (xlen x N); Q (N x N); (unused)
    double *Xf, *Xb;                    // row- / column-scaled strict triangles of the Cholesky factors of the X blocks (xy layout; unit-diagonal substitutions)
    double *xrd, *srd, *qrd;            // reciprocal diagonals of chol(X_b), L_j, L_Q
    double *Si, *Qi;                    // explicit inverses L_j^-1 (S layout) and L_Q^-1, lower triangular: every triangular solve with them is a product
    double *Xi;                         // chol(X_b)^-1 of the blocks with inv = 1 (xy layout)
    double *t, *u, *AY;                 // t = L^-1 rhs_x (xlen); u slabs (J x N); pairings per term
    // iterative refinement of the solve stage over many workgroups (k_mw_refine): the residual r_x (xlen), B^T dx of this rank's rows (N), the
    // correction (xlen, N); uadd: while the correction is solved, the vector subtracted from rhs_y beside sum_j u_j (= u2), else null
    double *S0;                         // S_j as assembled (S layout; written by the FACTOR stage as it reads S_j, which it overwrites with L_j): the residuals of the refinement need S_j
    int *mark_word;                     // null, or a word the FIRST workgroup of the next Cholesky / factor launch stores mark_value to as it starts: "everything in front of
    int mark_value, km;                 // (km: limbs of the pairing products -- the reference's matmul_prec, src/solver.jl:125, 1125-1143 -- K by default; mw_km_switch)  this launch on its stream is complete", for kernels of another stream that wait inside their launch (mw_wait_word) instead of for an event
    int pipe_q, pipe_bp;                // index of Q's region in pipe_pc; of the first matrix of the blocked path (k_mw_bp_diag_pipe: region pipe_bp + MwBp::slot)
    unsigned long long *pipe_stamps;    // diagnostic builds: [16][40] step stamps of the pipelined factorisations (clrs_mw_debug_pipe_stamps), or null
    unsigned long long *pipe_pc;        // hand-off granules of the pipelined factorisations (clrs_mw_pipe.hip.h): [J + 1][MWP_PC_WORDS], or null
    double *ub;                         // u' slabs of the refinement step (J x N; k_mw_solve_bwd MODE 1 writes them while other workgroups read u)
    // A right-hand side that is AFFINE in a scalar known late (the corrector of the interior-point iteration: rhs_x = rhs0 + mu_c tau, clrs_mw_ipm_host.inc): the
    // forward half ran on rhs0 (q.t, q.u) and, once per iteration, on tau (aff_t, aff_u); k_mw_solve_bwd MODE 1 waits for aff_wait >= aff_wait_value, reads
    // the K-limb scalar at aff_mu (planar, plane aff_mu_plane) and uses t + mu aff_t, u_j + mu aff_u_j, rhs_x + mu aff_rhs wherever it reads t, u_j, rhs_x.  null: off
    const double *aff_mu, *aff_rhs, *aff_t, *aff_u;
    const int *aff_wait;
    int aff_wait_value, aff_mu_plane;
    // a second right-hand side whose forward half rides on the launch of k_mw_potrf_q behind the first one's (workgroups nq + J ..): t = Si rhs into ride2_t, u_j into ride2_u
    const double *ride2_rhs;
    double *ride2_t, *ride2_u;
    double *AX;                         // per-term pairings w^T X^-1 v, beside AY (k_mw_saccum[_one]); null: not kept
    unsigned long long *refstat;        // bit patterns of non-negative doubles, atomicMax'ed by the correction's backward half (MODE 2): [0] max|dx'|, [1] max|dx|, [2] max|dy'|, [3] max|dy|
    double *rx2, *u2, *dx2, *dy2;
    const double *uadd;
    int *info;                          // [0] factor status, [1] Cholesky-of-X status
    int *pcnt;                          // [2 J + 3] arrival counters of the workgroups that share one factorisation (cluster j; J: Q; J + 1 + slot: matrices of the blocked path)
    // cluster sharding over ranks (one process per GPU): this context holds the clusters of rank `rank`; the partial Q and the
    // partial u of every rank are gathered into world slots and summed in rank order by every rank (src/solver.jl:1268-1269, 1550-1553)
    int rank, world, gathered, pad3;    // gathered: u comes from the gather slots (world > 1, or a communicator is attached)
    double *Qg, *ug;                    // [world][limbs * N * N], [world][limbs * N]
    // exact-product path of the pairing matrices (clrs_mw_exact.hip.h): blocks with mws_off[b] >= 0 are taken by k_mws_pair when mws_on
    const long long *mws_off;
    int mws_on, mwx_on;
    const long long *mwd_off;           // dense blocks with mwd_off[b] >= 0: X^-1 (A_e Y) by k_mwx_dense when mwd_on
    int mwd_on, pad5;
    const long long *mwx_off;           // blocks with mwx_off[b] >= 0: pairing matrices of ANY size from the digits of Z, T, V (k_mwx_slice, k_mwx_gram) when mwx_on
};

__device__ __forceinline__ void mw_mark(const MwDev &q);
namespace mwk {
using namespace mwa;

// LDS pointers carry their address space so that the accesses are ds_read / ds_write (a generic pointer makes every access a
// flat_load with its longer latency and its wait on both memory counters); the primitives are templates over the pointer types
typedef __attribute__((address_space(3))) double lds_d;

template <int K, class P>
__device__ __forceinline__ mw<K> ldx(P p, long plane, long i) {
    mw<K> r;
#pragma unroll
    for (int l = 0; l < K; l++) r.l[l] = p[(long)l * plane + i];
    return r;
}
template <int K, class P>
__device__ __forceinline__ void stx(P p, long plane, long i, const mw<K> &v) {
#pragma unroll
    for (int l = 0; l < K; l++) p[(long)l * plane + i] = v.l[l];
}

__device__ __forceinline__ void tri_index(int e, int &ii, int &jj) {      // e -> (ii >= jj) of a packed lower triangle
    ii = (int)((__builtin_sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
    while ((ii + 1) * (ii + 2) / 2 <= e) ii++;
    while (ii * (ii + 1) / 2 > e) ii--;
    jj = e - ii * (ii + 1) / 2;
}

// sum over groups of W consecutive lanes (W a power of two <= 64); every lane of the group gets the result
template <int K, int W>
__device__ __forceinline__ mw<K> lanes_sum(mw<K> v) {
#pragma unroll
    for (int off = W / 2; off > 0; off >>= 1) {
        mw<K> o;
#pragma unroll
        for (int l = 0; l < K; l++) o.l[l] = __shfl_xor(v.l[l], off, 64);
        v = add<K>(v, o);
    }
    return v;
}

// Position of entry (i, c), i >= c, of the lower triangular W of wg_potrf: full column-major storage with leading dimension ldw, or
// (ldw = 0) packed by columns, n (n + 1) / 2 entries -- the form the LDS copies take, so that a matrix and the inverse of its
// factor fit side by side up to n = 52 at 5 limbs (and n = 32 at 10)
__device__ __forceinline__ int w_col(int c, int n, int ldw) { return ldw ? c * ldw : c * n - c * (c + 1) / 2; }      // entry (i, c) is at w_col(c) + i (32-bit: the matrices of one workgroup are small)
__device__ __forceinline__ long w_index(int i, int c, int n, int ldw) { return w_col(c, n, ldw) + i; }
#define MW_TRI(n) ((long)(n) * ((n) + 1) / 2)

// Exact scaling of one elimination step: ex even with d 2^-ex in [1/2, 2); p1 = 2^-ex, ph = 2^(-ex/2)  (d > 0, normal)
__device__ __forceinline__ void pivot_scale(double head, double &p1, double &ph) {
    int ex = ((__double2hiint(head) >> 20) & 0x7ff) - 1023;
    ex += ex & 1;
    p1 = __hiloint2double((1023 - ex) << 20, 0);
    ph = __hiloint2double((1023 - ex / 2) << 20, 0);
}
#define MW_POTRF_SCR(K, n) (2 * (K) * (n))      // doubles of LDS scratch wg_potrf needs for an n x n matrix

// In-place lower Cholesky of the n x n matrix M (planar, leading dimension ld; only the lower triangle is read), reciprocal
// diagonal to rd.  approx_cholesky! (src/tools.jl:69-107): returns false at the first non-positive pivot.
//
// A K-limb reciprocal or reciprocal square root costs three to four K-limb products, and a Cholesky (or LDL^T) elimination has
// one of them on the dependent chain of every pivot.  The elimination is therefore carried in FRACTION-FREE form: with
// a~ = s_k a (s_k the product of the scaled pivots so far) one step is
//     a~_ij  <-  (d~_k a~_ij - a~_ik a~_jk) 2^-ex_k ,      s_(k+1) = s_k d~_k 2^-ex_k ,
// two products into one accumulator and an exact power of two that keeps s_k near 1: no division, no square root, no
// look-ahead, every entry of the trailing matrix equally cheap and ONE barrier per pivot.  The pivots keep their sign
// (s_k > 0), so the failure test is unchanged.  With INV the same step is applied to a unit matrix W beside M ([M | I]
// elimination), which leaves s_i (U^-1)_ij in row i.  What the chain no longer does happens once, for all pivots in parallel,
// after the loop:  d_k = d~_k / s_k,  1/sqrt(d_k),  L_kk = sqrt(d_k),  f_k = 1 / (s_k sqrt(d_k));  L_ik = a~_ik f_k,
// (L^-1)_ij = W_ij f_i.  The rounding errors are those of the classical elimination (each step rounds its entry once, relative
// to the larger of its two terms); the scaling is exact.
// With UT (default: without INV) the strict upper triangle of M is left holding U^T (u_ik = a~_ik / d~_k, the column-scaled
// factor of the backward substitutions); with INV the triangular solves are products with W and U is formed only on request.
// `scr`: LDS, MW_POTRF_SCR(K, n) doubles.
// `cw`, `cnw`: this workgroup forms the columns c = cw (mod cnw) of W only.  The elimination of M cannot be split without
// exchanging pivot columns, but the columns of W are independent of each other given M's: cnw workgroups that each repeat
// the elimination of M (bit for bit the same) and share out the columns of W finish in the time of M plus a cnw-th of W.
template <int K, bool INV, int NT = MW_NT, bool UT = !INV, class PM, class PR, class PW>
__device__ __forceinline__ bool wg_potrf(PM M, long plane, int n, int ld, PR rd, long rdplane, PW W, long wplane, int ldw, lds_d *scr, int tid, int cw = 0,
                                         int cnw = 1) {
    if (INV) {
        for (int e = tid; e < n * n; e += NT) {
            const int i = e % n, c = e / n;
            if (ldw && cnw > 1 && c % cnw != cw) continue;      // W in memory, shared with other workgroups: the columns of this one only
            if (ldw || i >= c) stx<K>(W, wplane, ldw ? i + (long)c * ldw : w_index(i, c, n, 0), i == c ? from_double<K>(1.0) : zero<K>());
        }
        __syncthreads();
    }
    lds_d *fs = scr, *us = scr + (long)K * n;                           // us holds s_k until the pivot's post-processing replaces it
    mw<K> srun = from_double<K>(1.0);                                   // s_k, carried by the last thread (idle in most tail rounds)
    if (tid == NT - 1) stx<K>(us, n, 0, srun);
    for (int k = 0; k < n; k++) {
#ifdef MW_STAMPS
        if (tid == 0 && k > 0) g_stamps[k] = wall_clock64();
#endif
        const mw<K> d = ldx<K>(M, plane, k + (long)k * ld);
        if (!(d.l[0] > 0.0)) return false;                              // every thread reads the same pivot: uniform exit
        if (k + 1 < n) {
            double p1, ph;
            pivot_scale(d.l[0], p1, ph);
            const mw<K> dh = mul_pow2<K>(d, p1);
            const int m = n - k - 1, trail = m * (m + 1) / 2;                    // the trailing triangle of M ...
            const int wcols = !INV || k < cw ? 0 : (k - cw) / cnw + 1;           // ... and rows k+1.. of this workgroup's columns <= k of W
            const int total = trail + m * wcols;
            for (int e = tid; e < total; e += NT) {
                int i, c;
                const bool tr = e < trail;
                if (tr) { tri_index(e, i, c); i += k + 1; c += k + 1; }
                else { const int e2 = e - trail; i = k + 1 + e2 % m; c = cw + cnw * (e2 / m); }
                const mw<K> ci = mul_pow2<K>(ldx<K>(M, plane, i + k * ld), ph);
                const int cb = tr ? c * ld : w_col(c, n, ldw);               // column base of the entry in M or in W
                const mw<K> cj = mul_pow2<K>(tr ? ldx<K>(M, plane, c + k * ld) : ldx<K>(W, wplane, cb + k), ph);
                const mw<K> v = tr ? ldx<K>(M, plane, cb + i) : ldx<K>(W, wplane, cb + i);
                acc<K> s;
                acc_zero<K>(s);
                acc_fma<K, K, K>(s, dh, v);
                acc_fma<K, K, K>(s, ci, cj, -1.0);
                const mw<K> r = acc_result<K>(s);
                if (tr) stx<K>(M, plane, cb + i, r);
                else stx<K>(W, wplane, cb + i, r);
            }
            if (tid == NT - 1) {
                srun = mul<K>(srun, dh);
                stx<K>(us, n, k + 1, srun);
                if (INV && (!ldw || cnw == 1 || (k + 1) % cnw == cw)) stx<K>(W, wplane, w_index(k + 1, k + 1, n, ldw), srun);
            }
        }
        __syncthreads();
    }
#ifdef MW_STAMPS
    if (tid == 0) g_stamps[n] = wall_clock64();
#endif
    // per pivot, in parallel: f_k = 1 / sqrt(s_k d~_k) = 1 / (s_k sqrt(d_k)), then 1/sqrt(d_k) = f_k s_k, sqrt(d_k) = d~_k f_k,
    // 1/d~_k = f_k^2 s_k: one reciprocal square root and products
    for (int k = tid; k < n; k += NT) {
        const long kk = k + (long)k * ld;
        const mw<K> sk = ldx<K>(us, n, k), dt = ldx<K>(M, plane, kk);
        const mw<K> f = rsqrt<K>(mul<K>(sk, dt)), rs = mul<K>(f, sk);
        stx<K>(rd, rdplane, k, rs);
        stx<K>(M, plane, kk, mul<K>(dt, f));
        stx<K>(fs, n, k, f);
        if (UT) stx<K>(us, n, k, mul<K>(f, rs));
    }
    __syncthreads();
#ifdef MW_STAMPS
    if (tid == 0) g_stamps[100] = wall_clock64();
#endif
    // one product per task: L_ik = a~_ik f_k; without INV also U^T into the upper triangle, with INV (L^-1)_ij = W_ij f_i
    const int T = n * (n - 1) / 2;
    for (int e = tid; e < (INV ? 2 * T : T); e += NT) {
        int i, c;
        tri_index(e < T ? e : e - T, i, c);
        i += 1;                                                          // strict lower triangle: i > c
        if (e < T) {
            const mw<K> a = ldx<K>(M, plane, i + (long)c * ld);
            if (UT) stx<K>(M, plane, c + (long)i * ld, mul<K>(a, ldx<K>(us, n, c)));
            stx<K>(M, plane, i + (long)c * ld, mul<K>(a, ldx<K>(fs, n, c)));
        } else if (c % cnw == cw) {
            const long idx = w_index(i, c, n, ldw);
            stx<K>(W, wplane, idx, mul<K>(ldx<K>(W, wplane, idx), ldx<K>(fs, n, i)));
        }
    }
    if (INV) for (int i = tid; i < n; i += NT) if (!ldw || cnw == 1 || i % cnw == cw) stx<K>(W, wplane, w_index(i, i, n, ldw), ldx<K>(rd, rdplane, i));
    __syncthreads();
#ifdef MW_STAMPS
    if (tid == 0) g_stamps[101] = wall_clock64();
#endif
    return true;
}

// The same two scaled triangles after wg_potrf, whose strict upper triangle already holds U^T = Bk: F[i,k] = L[i,k] / L[i,i].
template <int K, int NT = MW_NT, class PL, class PR, class PF, class PB>
__device__ __forceinline__ void wg_scaled_factors_u(PL L, long lplane, int ldl, PR rd, long rdplane, int n, PF F, long fplane, int ldf, PB Bk,
                                                    long bplane, int ldb, int tid) {
    for (int e = tid; e < n * n; e += NT) {
        const int i = e % n, k = e / n;
        if (i > k) {
            stx<K>(F, fplane, i + (long)k * ldf, mul<K>(ldx<K>(L, lplane, i + (long)k * ldl), ldx<K>(rd, rdplane, i)));
        } else {
            stx<K>(F, fplane, i + (long)k * ldf, zero<K>());
            stx<K>(Bk, bplane, i + (long)k * ldb, i == k ? zero<K>() : ldx<K>(L, lplane, i + (long)k * ldl));
        }
    }
}

// y = T v and y = T^T v for a lower triangular n x n matrix T (an explicit inverse factor), v and y planar vectors in LDS
// (y != v): eight lanes per row, the partial sums joined by shuffles.  Ends with a barrier.
#define MW_TV_W 8
template <int K, class PT, class PV>
__device__ __forceinline__ void wg_trmv_n(PT T, long tplane, int ldt, int n, PV v, long vplane, PV y, long yplane, int tid) {
    const int sub = tid % MW_TV_W;
    for (int i0 = 0; i0 < n; i0 += MW_NT / MW_TV_W) {
        const int i = i0 + tid / MW_TV_W;
        const bool live = i < n;
        const int ii = live ? i : 0;
        acc<K> s;
        acc_zero<K>(s);
        for (int c = sub; c <= ii; c += MW_TV_W) acc_fma<K, K, K>(s, ldx<K>(T, tplane, ii + (long)c * ldt), ldx<K>(v, vplane, c));
        const mw<K> r = lanes_sum<K, MW_TV_W>(acc_result<K>(s));
        if (live && sub == 0) stx<K>(y, yplane, i, r);
    }
    __syncthreads();
}
template <int K, class PT, class PV>
__device__ __forceinline__ void wg_trmv_t(PT T, long tplane, int ldt, int n, PV v, long vplane, PV y, long yplane, int tid) {
    const int sub = tid % MW_TV_W;
    for (int i0 = 0; i0 < n; i0 += MW_NT / MW_TV_W) {
        const int i = i0 + tid / MW_TV_W;
        const bool live = i < n;
        const int ii = live ? i : 0;
        acc<K> s;
        acc_zero<K>(s);
        for (int r = ii + sub; r < n; r += MW_TV_W) acc_fma<K, K, K>(s, ldx<K>(T, tplane, r + (long)ii * ldt), ldx<K>(v, vplane, r));
        const mw<K> w = lanes_sum<K, MW_TV_W>(acc_result<K>(s));
        if (live && sub == 0) stx<K>(y, yplane, i, w);
    }
    __syncthreads();
}

// Scaled strict triangles of a Cholesky factor: F[i,k] = L[i,k] / L[i,i], i > k (forward substitution with a unit diagonal:
// x_i = b_i / L_ii - sum_k F[i,k] x_k) and, stored TRANSPOSED so that a substitution step reads a contiguous column,
// Bk[i,k] = L[k,i] / L[i,i], i < k (backward: x_i = b_i / L_ii - sum_{k>i} Bk[i,k] x_k).
template <int K, class PL, class PR, class PF, class PB>
__device__ __forceinline__ void wg_scaled_factors(PL L, long lplane, int ldl, PR rd, long rdplane, int n, PF F, long fplane, int ldf, PB Bk,
                                                  long bplane, int ldb, int tid) {
    for (int e = tid; e < n * n; e += MW_NT) {
        const int i = e % n, k = e / n;
        if (i > k) {
            const mw<K> l = ldx<K>(L, lplane, i + (long)k * ldl);
            stx<K>(F, fplane, i + (long)k * ldf, mul<K>(l, ldx<K>(rd, rdplane, i)));
            stx<K>(Bk, bplane, k + (long)i * ldb, mul<K>(l, ldx<K>(rd, rdplane, k)));      // entry (k, i) of the transposed factor
        } else {
            stx<K>(F, fplane, i + (long)k * ldf, zero<K>());
            if (i == k) stx<K>(Bk, bplane, i + (long)k * ldb, zero<K>());
        }
    }
}

// B <- L^-1 B with the row-scaled factor F (n x nrhs, planar): rows scaled by 1/L_ii first, then one multiply-add per
// entry and ONE barrier per column of the factor
template <int K, class PF, class PR, class PB>
__device__ __forceinline__ void wg_trsm_f(PF F, long fplane, int ldf, PR rd, long rdplane, int n, PB B, long bplane, int ldb, int nrhs, int tid) {
    for (int e = tid; e < n * nrhs; e += MW_NT) {
        const int i = e % n, c = e / n;
        const long idx = i + (long)c * ldb;
        stx<K>(B, bplane, idx, mul<K>(ldx<K>(B, bplane, idx), ldx<K>(rd, rdplane, i)));
    }
    __syncthreads();
    for (int k = 0; k < n - 1; k++) {
        const int m = n - k - 1;
        for (int e = tid; e < m * nrhs; e += MW_NT) {
            const int i = k + 1 + e % m, c = e / m;
            const long idx = i + (long)c * ldb;
            stx<K>(B, bplane, idx, fnma<K>(ldx<K>(B, bplane, idx), ldx<K>(F, fplane, i + (long)k * ldf), ldx<K>(B, bplane, k + (long)c * ldb)));
        }
        __syncthreads();
    }
}
// B <- L^-T B with the column-scaled factor Bk
template <int K, class PF, class PR, class PB>
__device__ __forceinline__ void wg_trsm_b(PF Bk, long fplane, int ldf, PR rd, long rdplane, int n, PB B, long bplane, int ldb, int nrhs, int tid) {
    for (int e = tid; e < n * nrhs; e += MW_NT) {
        const int i = e % n, c = e / n;
        const long idx = i + (long)c * ldb;
        stx<K>(B, bplane, idx, mul<K>(ldx<K>(B, bplane, idx), ldx<K>(rd, rdplane, i)));
    }
    __syncthreads();
    for (int k = n - 1; k > 0; k--) {
        for (int e = tid; e < k * nrhs; e += MW_NT) {
            const int i = e % k, c = e / k;
            const long idx = i + (long)c * ldb;
            stx<K>(B, bplane, idx, fnma<K>(ldx<K>(B, bplane, idx), ldx<K>(Bk, fplane, i + (long)k * ldf), ldx<K>(B, bplane, k + (long)c * ldb)));
        }
        __syncthreads();
    }
}

// true (uniformly over the workgroup) in the workgroup that arrives last at `counter` out of `total`; every workgroup publishes
// its global writes before it counts itself, the last one resets the counter for the next launch
__device__ __forceinline__ bool wg_last_block(int *counter, unsigned total) {
    __shared__ int last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = atomicAdd((unsigned *)counter, 1u);
        last = (old == total - 1) ? 1 : 0;
        if (last) *counter = 0;
    }
    __syncthreads();
    if (last) __threadfence();
    return last != 0;
}

// copy a rows x cols planar matrix between two arrays (any address spaces)
// Four elements per thread and pass, every load of a pass issued before its first store (the index of a thread past the end is clamped
// instead of guarded): a pass is one round trip to the source instead of four, and a 96 x 96 block is 5 passes instead of 18.
template <int K, int NT = MW_NT, class PD, class PS>
__device__ __forceinline__ void wg_copy(PD dst, long dplane, int ldd, PS src, long splane, int lds_, int rows, int cols, int tid) {
    constexpr int U = K <= 6 ? 4 : 2;
    const int total = rows * cols;
    for (int e0 = tid; e0 < total; e0 += U * NT) {
        double v[U][K];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = min(e0 + u * NT, total - 1), i = e % rows, c = e / rows;
#pragma unroll
            for (int l = 0; l < K; l++) v[u][l] = src[(long)l * splane + i + (long)c * lds_];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = e0 + u * NT, i = e % rows, c = e / rows;
            if (e < total) {
#pragma unroll
                for (int l = 0; l < K; l++) dst[(long)l * dplane + i + (long)c * ldd] = v[u][l];
            }
        }
    }
}

}  // namespace mwk

extern __shared__ double mw_lds[];
#define MW_LDS ((mwk::lds_d *)mw_lds)
// diagnostic builds (-DCLRS_MW_STAMPS): launch-boundary stamps of the factorisation's chain into rows 8.. of the pipeline's stamp buffer (scripts/chain_stamps.py)
#ifdef CLRS_MW_STAMPS
#define MW_CHAIN_STAMP(q, idx, cond) do { if ((q).pipe_stamps && (cond)) (q).pipe_stamps[8 * 40 + (idx)] = wall_clock64(); } while (0)
#else
#define MW_CHAIN_STAMP(q, idx, cond) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------------------------------
// Cholesky of every X block: Xchol_b = chol(X_b), strict upper zero; reciprocal diagonal and the two scaled triangles kept
// for the substitutions that follow.  One workgroup per block; `lds` = 1: the block is factored in LDS.
// ---------------------------------------------------------------------------------------------------------------------
#define MW_X_SHARE 32        // PSD blocks of more rows whose inverse factor is formed in LDS: MW_INV_WG workgroups share its columns (k_mw_potrf_x)
__host__ __device__ __forceinline__ bool mw_x_shares(const MwBlk &k) { return k.inv == 1 && k.n > MW_X_SHARE; }
template <int K, bool INV, class PM, class PW>
__device__ __forceinline__ void mw_potrf_x_body(const MwDev &q, const MwBlk &k, PM M, long plane, PW W, long wplane, bool w_in_place, double *__restrict__ Xc, mwk::lds_d *bc,
                                                int tid, int bid) {
    using namespace mwk;
    const int n = k.n;
    // W in memory (blocks whose factor and inverse do not fit in LDS side by side), or W in LDS beside a block of more than MW_X_SHARE rows (the
    // elimination of [M | I] is then bound by the instruction rate of one compute unit: 244 us for the 48 x 48 blocks of Nsphere_packing N = 3):
    // gridDim.y workgroups repeat the elimination of M in their own LDS and share out the columns of W, like the factorisations of S_j and Q; the
    // first one writes the factor, each its columns of the inverse
    const bool share = w_in_place || (INV && mw_x_shares(k));
    const int cw = share ? blockIdx.y : 0, cnw = share ? gridDim.y : 1;
    const bool ok = wg_potrf<K, INV, MW_PT, true>(M, plane, n, n, q.xrd + k.rd_off, q.xrdlen, W, wplane, w_in_place ? n : 0, bc, tid, cw, cnw);   // the LDS copy of W is packed
    if (!ok && tid == 0) atomicMin(&q.info[1], bid + 1);
    __syncthreads();
    if (INV && !w_in_place && cnw > 1) {                  // this workgroup's columns of the inverse, out of its LDS
        for (int e = tid; e < n * n; e += MW_PT) {
            const int i = e % n, c = e / n;
            if (c % cnw != cw) continue;
#pragma unroll
            for (int l = 0; l < K; l++) q.Xi[(long)l * q.xylen + k.xyoff + e] = (ok && i >= c) ? (double)W[(long)l * wplane + w_index(i, c, n, 0)] : 0.0;
        }
    }
    if (cw != 0) return;
    if (ok) wg_scaled_factors_u<K, MW_PT>(M, plane, n, q.xrd + k.rd_off, q.xrdlen, n, q.Xf + k.xyoff, q.xylen, n, q.Xb + k.xyoff, q.xylen, n, tid);
    for (int e = tid; e < n * n; e += MW_PT) {
        const int i = e % n, c = e / n;
#pragma unroll
        for (int l = 0; l < K; l++) {
            Xc[(long)l * q.xylen + k.xyoff + e] = (i >= c) ? (double)M[(long)l * plane + e] : 0.0;
            if (INV && !w_in_place && cnw == 1) q.Xi[(long)l * q.xylen + k.xyoff + e] = (ok && i >= c) ? (double)W[(long)l * wplane + w_index(i, c, n, 0)] : 0.0;
        }
    }
}
// Workgroups NB .. 2 NB - 1 (launched by the interior-point iteration only, when every block has inv = 1) factor a SECOND
// block-diagonal matrix, Y, and keep just the inverse of its factor (Yi) and a failure flag per block: the step length of Y
// (src/solver.jl:1644-1655) needs chol(Y)^-1, which depends on nothing computed during the iteration, so it rides along on
// compute units the launch would leave idle.
template <int K>
__global__ __launch_bounds__(MW_PT) void k_mw_potrf_x(const MwDev q, const double *__restrict__ X, double *__restrict__ Xc, int lds, const double *__restrict__ Y2,
                                                      double *__restrict__ Yi, int *__restrict__ yfail) {
    using namespace mwk;
    mw_mark(q);
    const bool second = (int)blockIdx.x >= q.NB;
    const MwBlk &k = q.blk[second ? blockIdx.x - q.NB : blockIdx.x];
    const int n = k.n, tid = threadIdx.x;
    if (blockIdx.y != 0 && !(lds && (k.inv == 2 || mw_x_shares(k)))) return;   // more than one workgroup per matrix only where the inverse is formed in memory, or beside a large block
    lds_d *bc = MW_LDS;                                   // scratch of wg_potrf, in front of the matrix
    if (second) {
        lds_d *M = MW_LDS + MW_POTRF_SCR(K, n);
        wg_copy<K, MW_PT>(M, (long)n * n, n, Y2 + k.xyoff, q.xylen, n, n, n, tid);
        __syncthreads();
        bool ok;
        if (k.inv == 1) {
            lds_d *W = M + (long)K * n * n, *rdl = W + (long)K * MW_TRI(n);
            const int cw = mw_x_shares(k) ? blockIdx.y : 0, cnw = mw_x_shares(k) ? gridDim.y : 1;
            ok = wg_potrf<K, true, MW_PT, false>(M, (long)n * n, n, n, rdl, n, W, MW_TRI(n), 0, bc, tid, cw, cnw);
            if (ok) {
                for (int e = tid; e < n * n; e += MW_PT) {
                    const int i = e % n, c = e / n;
                    if (c % cnw != cw) continue;
#pragma unroll
                    for (int l = 0; l < K; l++) Yi[(long)l * q.xylen + k.xyoff + e] = (i >= c) ? (double)W[(long)l * MW_TRI(n) + w_index(i, c, n, 0)] : 0.0;
                }
            }
        } else {                                        // the inverse in place in memory
            lds_d *rdl = M + (long)K * n * n;
            ok = wg_potrf<K, true, MW_PT, false>(M, (long)n * n, n, n, rdl, n, Yi + k.xyoff, q.xylen, n, bc, tid, blockIdx.y, gridDim.y);
        }
        if (tid == 0 && blockIdx.y == 0) yfail[blockIdx.x - q.NB] = ok ? 0 : 1;
        return;
    }
    if (lds) {
        lds_d *M = MW_LDS + MW_POTRF_SCR(K, n);
        wg_copy<K, MW_PT>(M, (long)n * n, n, X + k.xyoff, q.xylen, n, n, n, tid);
        __syncthreads();
        if (k.inv == 1) mw_potrf_x_body<K, true>(q, k, M, (long)n * n, M + (long)K * n * n, MW_TRI(n), false, Xc, bc, tid, blockIdx.x);
        else if (k.inv == 2) mw_potrf_x_body<K, true>(q, k, M, (long)n * n, q.Xi + k.xyoff, q.xylen, true, Xc, bc, tid, blockIdx.x);
        else mw_potrf_x_body<K, false>(q, k, M, (long)n * n, M, 0, false, Xc, bc, tid, blockIdx.x);
    } else {
        double *M = Xc + k.xyoff;
        wg_copy<K, MW_PT>(M, q.xylen, n, X + k.xyoff, q.xylen, n, n, n, tid);
        __syncthreads();
        mw_potrf_x_body<K, false>(q, k, M, q.xylen, M, 0, false, Xc, bc, tid, blockIdx.x);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Per low-rank block and per tile of MW_CT unique vectors:  T = Y V  and  Z = L^-1 V  (L = chol X_b).
// These are the reference's part_r products (src/solver.jl:1125, 1137) with X^-1 = L^-T L^-1 split over the two sides
// of the pairing: V^T X^-1 V = Z^T Z, so the explicit inverse (inv_cho_precomp!, :1117) is never formed.
// ---------------------------------------------------------------------------------------------------------------------
// T[:, c] = Y[:, rows(c)] V[rows(c), c] and (inverse factors) Z[:, c] = Xi V[:, c] for the columns c0 .. c0 + nc - 1: ZW lanes per entry
template <int K, int KM, int DK, int ZW>
__device__ __forceinline__ void mw_zt_products_km(const MwDev &q, const MwBlk &k, const double *__restrict__ Y, int c0, int nc, bool with_z) {
    using namespace mwk;
    const int n = k.n, tid = threadIdx.x, dl = k.delta, sub = tid % ZW;
    const double *V = q.V + k.v_off;
    const int *vrow = q.vrow + k.vrow_off;
    for (int e0 = 0; e0 < n * nc; e0 += MW_NT / ZW) {
        const int e = e0 + tid / ZW;
        const bool live = e < n * nc;
        const int ee = live ? e : 0;
        const int i = ee % n, c = c0 + ee / n, r0 = vrow[c];
        acc<KM> s;
        acc_zero<KM>(s);
        for (int kk = r0 + sub; kk < r0 + dl; kk += ZW) acc_fma<KM, KM, DK>(s, ldx<KM>(Y + k.xyoff, q.xylen, i + (long)kk * n), ldx<DK>(V, q.Vp, kk + (long)c * n));
        mw<KM> v = lanes_sum<KM, ZW>(acc_result<KM>(s));
        if (live && sub == 0) stx<K>(q.Tm + k.z_off, q.zlen, i + (long)c * n, cvt<K, KM>(v));
    }
    if (!with_z) return;
    const double *Xi = q.Xi + k.xyoff;                 // rows above the first nonzero row of the vector are zero
    for (int e0 = 0; e0 < n * nc; e0 += MW_NT / ZW) {
        const int e = e0 + tid / ZW;
        const bool live = e < n * nc;
        const int ee = live ? e : 0;
        const int i = ee % n, c = c0 + ee / n, r0 = vrow[c];
        acc<KM> s;
        acc_zero<KM>(s);
        for (int kk = r0 + sub; kk <= i; kk += ZW) acc_fma<KM, KM, DK>(s, ldx<KM>(Xi, q.xylen, i + (long)kk * n), ldx<DK>(V, q.Vp, kk + (long)c * n));
        mw<KM> v = lanes_sum<KM, ZW>(acc_result<KM>(s));
        if (live && sub == 0) stx<K>(q.Z + k.z_off, q.zlen, i + (long)c * n, cvt<K, KM>(v));
    }
}
template <int K, int DK, int ZW>
__device__ __forceinline__ void mw_zt_products(const MwDev &q, const MwBlk &k, const double *__restrict__ Y, int c0, int nc, bool with_z) {
    mw_km_switch<K>(q.km, [&](auto kmc) { mw_zt_products_km<K, decltype(kmc)::value, DK, ZW>(q, k, Y, c0, nc, with_z); });
}
// ct: columns of V per workgroup -- MW_CT with two lanes per entry, or (ct = 2, small launches with inverse factors: the named problems have twelve
// workgroups of eight columns) two columns with eight lanes per entry: a quarter of the multiply-adds per lane
template <int K, int DK>
__global__ __launch_bounds__(MW_NT) void k_mw_zt(const MwDev q, const double *__restrict__ Y, int lds_L, int use_inv, int ct) {
    using namespace mwk;
    if (q.mws_on && q.mws_off[q.lr_list[blockIdx.y]] >= 0) return;      // k_mws_pair forms the pairing matrices of this block
    const MwBlk &k = q.blk[q.lr_list[blockIdx.y]];
    const int n = k.n, U = k.U, tid = threadIdx.x;
    const int c0 = blockIdx.x * ct;
    if (c0 >= U) return;
    const int nc = min(ct, U - c0);
    const double *V = q.V + k.v_off;
    if (ct != MW_CT) { mw_zt_products<K, DK, 8>(q, k, Y, c0, nc, true); return; }      // (launched so only when every block has its inverse factor)
    mw_zt_products<K, DK, 2>(q, k, Y, c0, nc, use_inv && k.inv);
    if (use_inv && k.inv) return;
    // Z tile in LDS: forward substitution with the row-scaled factor of X_b
    lds_d *Zt = MW_LDS;
    const long zp = (long)n * MW_CT;
    for (int e = tid; e < n * nc; e += MW_NT) {
        const int i = e % n, c = e / n;
#pragma unroll
        for (int l = 0; l < K; l++) Zt[(long)l * zp + e] = l < DK ? V[(long)l * q.Vp + i + (long)(c0 + c) * n] : 0.0;
    }
    if (lds_L) {
        lds_d *Ls = MW_LDS + (long)K * zp;
        wg_copy<K>(Ls, (long)n * n, n, q.Xf + k.xyoff, q.xylen, n, n, n, tid);
        __syncthreads();
        wg_trsm_f<K>(Ls, (long)n * n, n, q.xrd + k.rd_off, q.xrdlen, n, Zt, zp, n, nc, tid);
    } else {
        __syncthreads();
        wg_trsm_f<K>(q.Xf + k.xyoff, q.xylen, n, q.xrd + k.rd_off, q.xrdlen, n, Zt, zp, n, nc, tid);
    }
    for (int e = tid; e < n * nc; e += MW_NT) {
#pragma unroll
        for (int l = 0; l < K; l++) q.Z[(long)l * q.zlen + k.z_off + (long)c0 * n + e] = Zt[(long)l * zp + e];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Pairing matrices of a low-rank block: GX = Z^T Z = V^T X^-1 V, GY = V^T T = V^T Y V (U x U, symmetric; the
// reference's bilinear_pairings_Xinv / _Y, src/solver.jl:1131, 1143).  Four lanes per entry of the lower triangle.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef MW_GRAM_W
#define MW_GRAM_W 4          // (sixteen lanes per entry: PolyOpt 2d = 40 0.462 -> 0.458 ms per iteration, Nsphere_packing N = 3 1.853 -> 1.860, the named problem unchanged: left at four)
#endif
// a 1 x 1 dense block: T_e = (Y / X) a_e and the table Sd[e, e'] = a_e' T_e right behind it, one thread per matrix, then per pair
template <int K, int DK>
__device__ __forceinline__ void mw_dense_1x1(const MwDev &q, const MwBlk &k, const double *__restrict__ Y, int tid) {
    using namespace mwk;
    const int cnt = k.cnt;
    const double *A = q.dA + k.a_off;
    double *W = q.W + k.w_off;
    mw<K> rd = ldx<K>(q.xrd + k.rd_off, q.xrdlen, 0);
    mw<K> yx = mul<K>(ldx<K>(Y + k.xyoff, q.xylen, 0), mul<K>(rd, rd));      // Y / X
    for (int ee = tid; ee < cnt; ee += MW_NT) stx<K>(W, q.wlen, ee, mulx<K, K, DK>(yx, ldx<DK>(A, q.dAp, ee)));
    __syncthreads();
    for (int o = tid; o < cnt * (cnt + 1) / 2; o += MW_NT) {
        int e2, e1;
        tri_index(o, e2, e1);
        const mw<K> v = mulx<K, K, DK>(ldx<K>(W, q.wlen, e1), ldx<DK>(A, q.dAp, e2));
        stx<K>(q.Sd + k.sd_off, q.sdlen, e1 + (long)e2 * cnt, v);
        stx<K>(q.Sd + k.sd_off, q.sdlen, e2 + (long)e1 * cnt, v);
    }
}
template <int K, int KM, int DK>
__device__ __forceinline__ void mw_gram_km(const MwDev &q) {
    using namespace mwk;
    const MwBlk &k = q.blk[q.lr_list[blockIdx.y]];
    const int n = k.n, U = k.U, dl = k.delta;
    const int e = blockIdx.x * (MW_NT / MW_GRAM_W) + threadIdx.x / MW_GRAM_W, sub = threadIdx.x % MW_GRAM_W;
    const int tot = U * (U + 1) / 2;
    if (blockIdx.x * (MW_NT / MW_GRAM_W) >= tot) return;
    const bool live = e < tot;
    int a, b;
    tri_index(live ? e : 0, a, b);
    const double *V = q.V + k.v_off;
    const int *vrow = q.vrow + k.vrow_off;
    const double *Z = q.Z + k.z_off, *T = q.Tm + k.z_off;
    acc<KM> s;
    acc_zero<KM>(s);
    // rows above the first nonzero row of either vector are zero in Z = L^-1 V
    const int i0 = max(vrow[a], vrow[b]);
    for (int i = i0 + sub; i < n; i += MW_GRAM_W) acc_fma<KM, KM, KM>(s, ldx<KM>(Z, q.zlen, i + (long)a * n), ldx<KM>(Z, q.zlen, i + (long)b * n));
    const mw<K> gx = cvt<K, KM>(lanes_sum<KM, MW_GRAM_W>(acc_result<KM>(s)));
    acc_zero<KM>(s);
    const int r0 = vrow[a];
    for (int i = r0 + sub; i < r0 + dl; i += MW_GRAM_W) acc_fma<KM, KM, DK>(s, ldx<KM>(T, q.zlen, i + (long)b * n), ldx<DK>(V, q.Vp, i + (long)a * n));
    const mw<K> gy = cvt<K, KM>(lanes_sum<KM, MW_GRAM_W>(acc_result<KM>(s)));
    if (live && sub == 0) {
        stx<K>(q.GX + k.g_off, q.glen, a + (long)b * U, gx);
        stx<K>(q.GX + k.g_off, q.glen, b + (long)a * U, gx);
        stx<K>(q.GY + k.g_off, q.glen, a + (long)b * U, gy);
        stx<K>(q.GY + k.g_off, q.glen, b + (long)a * U, gy);
    }
}
template <int K, int DK>
__global__ __launch_bounds__(MW_NT) void k_mw_gram(const MwDev q, const double *__restrict__ Y) {
    using namespace mwk;
    // rows of the grid beyond the low-rank blocks (the launch adds them when every dense block is 1 x 1): the dense blocks' tables, which depend on
    // X and Y only -- beside the pairing matrices instead of a launch of their own behind them (k_mw_dense_t)
    if ((int)blockIdx.y >= q.nlr) {
        if (blockIdx.x == 0) mw_dense_1x1<K, DK>(q, q.blk[q.dn_list[blockIdx.y - q.nlr]], Y, threadIdx.x);
        return;
    }
    if (q.mws_on && q.mws_off[q.lr_list[blockIdx.y]] >= 0) return;
    if (q.mwx_on && q.mwx_off[q.lr_list[blockIdx.y]] >= 0) return;      // k_mwx_gram
    mw_km_switch<K>(q.km, [&](auto kmc) { mw_gram_km<K, decltype(kmc)::value, DK>(q); });
}

// ---------------------------------------------------------------------------------------------------------------------
// Dense ("high rank") block: T_e = X^-1 A_e Y for every matrix of the block, then the table Sd[e, e'] = <A_e', T_e>
// (src/solver.jl:1089-1104).  k_mw_dense_t: one workgroup per (block, matrix) -- the matrices are independent -- with
// X^-1 A = Xi^T (Xi A) as products when the inverse factor of the block exists and two buffers fit in LDS (`prod`), by the
// two substitutions otherwise; n = 1 blocks take one thread per matrix and per pair in the same launch.  k_mw_dense_s (blocks
// with n > 1): one wave per pair (e, e') of the table.
// ---------------------------------------------------------------------------------------------------------------------
template <int K, int DK>
__global__ __launch_bounds__(MW_NT) void k_mw_dense_t(const MwDev q, const double *__restrict__ Y, int use_inv, int two_buffers, int panels) {
    using namespace mwk;
    const MwBlk &k = q.blk[q.dn_list[blockIdx.x]];
    const int n = k.n, cnt = k.cnt, tid = threadIdx.x, e = blockIdx.y;
    const double *A = q.dA + k.a_off;
    double *W = q.W + k.w_off;
    const long nn = (long)n * n;
    if (n == 1) {
        if (e == 0) mw_dense_1x1<K, DK>(q, k, Y, tid);
        return;
    }
    if (e >= cnt || (panels && use_inv && k.inv && n > 16)) return;          // the latter: k_mw_dense_tp or k_mwx_dense
    lds_d *M = MW_LDS, *M2 = M + (long)K * nn;
    if (use_inv && k.inv && two_buffers) {
        const double *Xi = q.Xi + k.xyoff;
        for (int o = tid; o < nn; o += MW_NT) {            // M = Xi A_e
            const int i = o % n, c = o / n;
            acc<K> s;
            acc_zero<K>(s);
            for (int r = 0; r <= i; r++) acc_fma<K, K, DK>(s, ldx<K>(Xi, q.xylen, i + (long)r * n), ldx<DK>(A, q.dAp, (long)e * nn + r + (long)c * n));
            stx<K>(M, nn, o, acc_result<K>(s));
        }
        __syncthreads();
        for (int o = tid; o < nn; o += MW_NT) {            // M2 = Xi^T M
            const int i = o % n, c = o / n;
            acc<K> s;
            acc_zero<K>(s);
            for (int r = i; r < n; r++) acc_fma<K, K, K>(s, ldx<K>(Xi, q.xylen, r + (long)i * n), ldx<K>(M, nn, r + (long)c * n));
            stx<K>(M2, nn, o, acc_result<K>(s));
        }
        __syncthreads();
        M = M2;
    } else {
        for (int i = tid; i < nn; i += MW_NT) {
#pragma unroll
            for (int l = 0; l < K; l++) M[(long)l * nn + i] = l < DK ? A[(long)l * q.dAp + (long)e * nn + i] : 0.0;
        }
        __syncthreads();
        wg_trsm_f<K>(q.Xf + k.xyoff, q.xylen, n, q.xrd + k.rd_off, q.xrdlen, n, M, nn, n, n, tid);
        wg_trsm_b<K>(q.Xb + k.xyoff, q.xylen, n, q.xrd + k.rd_off, q.xrdlen, n, M, nn, n, n, tid);
    }
    for (int o = tid; o < nn; o += MW_NT) {                // T_e = (X^-1 A_e) Y
        const int i = o % n, c = o / n;
        acc<K> s;
        acc_zero<K>(s);
        for (int kk = 0; kk < n; kk++) acc_fma<K, K, K>(s, ldx<K>(M, nn, i + (long)kk * n), ldx<K>(Y + k.xyoff, q.xylen, kk + (long)c * n));
        stx<K>(W, q.wlen, (long)e * nn + o, acc_result<K>(s));
    }
}
// T_e = X^-1 (A_e Y) for the blocks of side > 16 that have an inverse factor, by COLUMN PANELS: associated this way every product is independent
// column by column, so a matrix is shared by n / pc workgroups (pc = 256 / n columns: one entry per thread) instead of one workgroup walking
// four entries per thread through three products (SDPA x64: 512 matrices of 32 x 32 on 256 compute units, 2 waves per SIMD: 42 % of the pipe).
// k_mw_dense_t skips these blocks (`panels` = 1).
template <int K, int DK>
__global__ __launch_bounds__(MW_NT) void k_mw_dense_tp(const MwDev q, const double *__restrict__ Y) {
    using namespace mwk;
    const MwBlk &k = q.blk[q.dn_list[blockIdx.x]];
    const int n = k.n, tid = threadIdx.x, e = blockIdx.y;
    if (n <= 16 || !k.inv || e >= k.cnt) return;
    if (q.mwd_on && q.mwd_off[q.dn_list[blockIdx.x]] >= 0) return;      // k_mwx_dense
    const int pc = max(1, MW_NT / n), c0 = blockIdx.z * pc;
    if (c0 >= n) return;
    const int pw = min(pc, n - c0);
    const long nn = (long)n * n, np = (long)n * pc;
    const double *A = q.dA + k.a_off, *Xi = q.Xi + k.xyoff;
    lds_d *M1 = MW_LDS, *M2 = M1 + (long)K * np;
    for (int o = tid; o < n * pw; o += MW_NT) {            // M1 = A_e Y[:, panel]
        const int i = o % n, cl = o / n;
        acc<K> s;
        acc_zero<K>(s);
        for (int kk = 0; kk < n; kk++) acc_fma<K, K, DK>(s, ldx<K>(Y + k.xyoff, q.xylen, kk + (long)(c0 + cl) * n), ldx<DK>(A, q.dAp, (long)e * nn + i + (long)kk * n));
        stx<K>(M1, np, o, acc_result<K>(s));
    }
    __syncthreads();
    for (int o = tid; o < n * pw; o += MW_NT) {            // M2 = Xi M1
        const int i = o % n, cl = o / n;
        acc<K> s;
        acc_zero<K>(s);
        for (int r = 0; r <= i; r++) acc_fma<K, K, K>(s, ldx<K>(Xi, q.xylen, i + (long)r * n), ldx<K>(M1, np, r + (long)cl * n));
        stx<K>(M2, np, o, acc_result<K>(s));
    }
    __syncthreads();
    double *W = q.W + k.w_off;
    for (int o = tid; o < n * pw; o += MW_NT) {            // T_e[:, panel] = Xi^T M2
        const int i = o % n, cl = o / n;
        acc<K> s;
        acc_zero<K>(s);
        for (int r = i; r < n; r++) acc_fma<K, K, K>(s, ldx<K>(Xi, q.xylen, r + (long)i * n), ldx<K>(M2, np, r + (long)cl * n));
        stx<K>(W, q.wlen, (long)e * nn + i + (long)(c0 + cl) * n, acc_result<K>(s));
    }
}
// Sd[e, e'] = <A_e', T_e>, e <= e' computed, mirrored: W lanes per pair -- a wave for large blocks, eight lanes for blocks of a few dozen entries
// (the three-point bound has 9 dense blocks of sides <= 9 with up to 221 matrices each: 24 000 pairs of 81-term dot products, for which the six
// rounds of a wave-wide sum were five times the work)
template <int K, int DK, int W>
__device__ __forceinline__ void mw_dense_s_body(const MwDev &q) {
    using namespace mwk;
    const MwBlk &k = q.blk[q.dn_list[blockIdx.x]];
    if (k.n == 1) return;                                  // done by k_mw_dense_t
    const int cnt = k.cnt, lane = threadIdx.x % W;
    const int o0 = blockIdx.y * (MW_NT / W) + threadIdx.x / W, tot = cnt * (cnt + 1) / 2;
    if ((int)(blockIdx.y * (MW_NT / W)) >= tot) return;    // uniform over the workgroup
    const bool live = o0 < tot;
    const long nn = (long)k.n * k.n;
    const double *A = q.dA + k.a_off, *Wt = q.W + k.w_off;
    int e2, e1;
    tri_index(live ? o0 : 0, e2, e1);       // e2 >= e1
    acc<K> s;
    acc_zero<K>(s);
    for (long i = lane; i < nn; i += W) acc_fma<K, K, DK>(s, ldx<K>(Wt, q.wlen, (long)e1 * nn + i), ldx<DK>(A, q.dAp, (long)e2 * nn + i));
    const mw<K> v = lanes_sum<K, W>(acc_result<K>(s));
    if (live && lane == 0) {
        stx<K>(q.Sd + k.sd_off, q.sdlen, e1 + (long)e2 * cnt, v);
        stx<K>(q.Sd + k.sd_off, q.sdlen, e2 + (long)e1 * cnt, v);
    }
}
template <int K, int DK>
__global__ __launch_bounds__(MW_NT) void k_mw_dense_s(const MwDev q, int lanes) {
    if (lanes == 8) mw_dense_s_body<K, DK, 8>(q);
    else if (lanes == 16) mw_dense_s_body<K, DK, 16>(q);
    else mw_dense_s_body<K, DK, 64>(q);
}

// ---------------------------------------------------------------------------------------------------------------------
// S_j[p, q] = sum over the blocks of the cluster: low rank  sum_{t in p, t' in q} lam_t lam_t' GX[a_t, b_t'] GY[a_t', b_t]
// with a_t = pointers_left[s][(r,p,k)], b_t = pointers_right[r][(s,p,k)] of the term t = (p,r,s,k) (src/solver.jl:1198-1203)
// (the accumulation loops src/solver.jl:1176-1212 with the four Dict lookups replaced by the sorted term table),
// dense  Sd[e_p, e_q].  One thread per entry q >= p, mirrored write (symmetric!, src/tools.jl:43-57); the per-term pairings
// A_Y (src/solver.jl:1152-1170) are written by the first workgroups of the same launch.
// ---------------------------------------------------------------------------------------------------------------------
#define MW_SA_W 4            // most lanes per entry; the launch takes 1, 2 or 4 (the largest number of PSD blocks of a cluster, rounded up)
template <int K, int DK, int W>
__device__ __forceinline__ void mw_saccum_body(const MwDev &q) {
    using namespace mwk;
    const MwClu &c = q.clu[blockIdx.y];
    if (c.one_term) return;                              // k_mw_saccum_one
    const int P = c.P;
    if (blockIdx.x * (MW_NT / W) >= P * (P + 1) / 2) return;
    // W lanes per entry, one block of the cluster each: the chains of dependent loads (block -> term range -> pointers ->
    // pairings) of the blocks run side by side instead of one after the other
    const int e = blockIdx.x * (MW_NT / W) + threadIdx.x / W, sub = threadIdx.x % W;
    const bool live = e < P * (P + 1) / 2;
    int qq, pp;
    tri_index(live ? e : 0, qq, pp);           // qq >= pp
    acc<K> s;
    acc_zero<K>(s);
    for (int b = c.b0 + sub; b < c.b1; b += W) {
        const MwBlk &k = q.blk[b];
        if (k.kind == 0) {
            const int *tp = q.tptr + k.tptr_off;
            const int U = k.U;
            for (int t = tp[pp]; t < tp[pp + 1]; t++) {
                for (int t2 = tp[qq]; t2 < tp[qq + 1]; t2++) {
                    mw<K> gx = ldx<K>(q.GX + k.g_off, q.glen, q.st_a[t] + (long)q.st_b[t2] * U);
                    mw<K> gy = ldx<K>(q.GY + k.g_off, q.glen, q.st_a[t2] + (long)q.st_b[t] * U);
                    mw<K> w = mul<K>(gx, gy);
                    constexpr int LL = (2 * DK + 1 < K) ? 2 * DK + 1 : K;   // lambda lambda' is exact in 2 DK limbs; one more bin so that none of them rounds
                    mw<LL> ll = mulx<LL, DK, DK>(ldx<DK>(q.st_lam, q.lamp, t), ldx<DK>(q.st_lam, q.lamp, t2));
                    acc_fma<K, K, LL>(s, w, ll);
                }
            }
        } else {
            const int *dm = q.dmap + k.dmap_off;
            const int e1 = dm[pp], e2 = dm[qq];
            if (e1 >= 0 && e2 >= 0) acc_add<K, K>(s, ldx<K>(q.Sd + k.sd_off, q.sdlen, e1 + (long)e2 * k.cnt));
        }
    }
    const mw<K> v = lanes_sum<K, W>(acc_result<K>(s));
    if (live && sub == 0) {
        stx<K>(q.S + c.Soff, q.Slen, pp + (long)qq * P, v);
        stx<K>(q.S + c.Soff, q.Slen, qq + (long)pp * P, v);
    }
}
template <int K, int DK>
__global__ __launch_bounds__(MW_NT) void k_mw_saccum(const MwDev q, int lanes) {
    using namespace mwk;
    if (blockIdx.y == 0) {                               // A_Y per term: w^T Y v
        for (mwi64 t = (mwi64)blockIdx.x * MW_NT + threadIdx.x; t < q.T; t += (mwi64)gridDim.x * MW_NT) {
            const int b = q.ay_blk[t];
            if (b < 0) continue;
            const MwBlk &k = q.blk[b];
            stx<K>(q.AY, q.T, t, ldx<K>(q.GY + k.g_off, q.glen, q.ay_a[t] + (long)q.ay_b[t] * k.U));
            if (q.AX) stx<K>(q.AX, q.T, t, ldx<K>(q.GX + k.g_off, q.glen, q.ay_a[t] + (long)q.ay_b[t] * k.U));
        }
    }
    if (lanes == 1) mw_saccum_body<K, DK, 1>(q);
    else if (lanes == 2) mw_saccum_body<K, DK, 2>(q);
    else mw_saccum_body<K, DK, 4>(q);
}

// The same for clusters with at most four PSD blocks and at most ONE low-rank term per (constraint, block) -- every sampled problem whose
// constraint matrices are rank one per block (the sphere-packing and polynomial-optimisation families).  One lane per entry, and the blocks
// of the cluster UNROLLED: the term of the row and of the column in every block first, then their vector indices, then the pairings and
// weights of all blocks, then the arithmetic -- four rounds of independent loads instead of one chain of dependent ones per block and lane
// (k_mw_saccum with one lane per entry walks the blocks one after the other: 390 us on 2048 clusters; with a lane per block it pays
// for the sums across lanes: 245 us; this kernel: 108 us).  Folding the eigenvalues lambda lambda' into the columns of Z (so that GX carries
// them and an entry is one K x K multiply-add per block) was tried on top: 92 us here, but +24 us in k_mws_pair, whose VALU phases are on its
// critical path -- not kept.
template <int K, int DK>
__global__ __launch_bounds__(MW_NT) void k_mw_saccum_one(const MwDev q, int do_ay) {
    using namespace mwk;
    const MwClu &c = q.clu[blockIdx.y];
    if (!c.one_term) return;
    const int P = c.P;
    const int e = blockIdx.x * MW_NT + threadIdx.x;
    if (do_ay) {                                         // A_Y per term, w^T Y v, of this cluster's blocks (when no cluster is left to k_mw_saccum, which writes them all)
        for (int b = c.b0; b < c.b1; b++) {
            const MwBlk &k = q.blk[b];
            if (k.kind != 0) continue;
            const int *tp = q.tptr + k.tptr_off;
            for (int t = tp[0] + e; t < tp[P]; t += gridDim.x * MW_NT) {
                stx<K>(q.AY, q.T, t, ldx<K>(q.GY + k.g_off, q.glen, q.ay_a[t] + (long)q.ay_b[t] * k.U));
                if (q.AX) stx<K>(q.AX, q.T, t, ldx<K>(q.GX + k.g_off, q.glen, q.ay_a[t] + (long)q.ay_b[t] * k.U));
            }
        }
    }
    if (e >= P * (P + 1) / 2) return;
    int qq, pp;
    tri_index(e, qq, pp);           // qq >= pp
    constexpr int MB = 4;
    int tA[MB], tB[MB], kind[MB];
    bool has[MB];
    const MwBlk *kb[MB];
#pragma unroll
    for (int bi = 0; bi < MB; bi++) {
        const int b = min(c.b0 + bi, c.b1 - 1);
        kb[bi] = &q.blk[b];
        kind[bi] = kb[bi]->kind;
        has[bi] = c.b0 + bi < c.b1;
        tA[bi] = tB[bi] = 0;
        if (kind[bi] == 0) {
            const int *tp = q.tptr + kb[bi]->tptr_off;
            const int a0 = tp[pp], a1 = tp[pp + 1], b0 = tp[qq], b1 = tp[qq + 1];
            has[bi] = has[bi] && a1 > a0 && b1 > b0;
            tA[bi] = a0; tB[bi] = b0;
        } else {
            const int *dm = q.dmap + kb[bi]->dmap_off;
            tA[bi] = dm[pp]; tB[bi] = dm[qq];
            has[bi] = has[bi] && tA[bi] >= 0 && tB[bi] >= 0;
        }
    }
    long ix[MB], iy[MB];
#pragma unroll
    for (int bi = 0; bi < MB; bi++) {
        ix[bi] = iy[bi] = 0;
        if (kind[bi] == 0 && has[bi]) {
            const int U = kb[bi]->U;
            ix[bi] = q.st_a[tA[bi]] + (long)q.st_b[tB[bi]] * U;
            iy[bi] = q.st_a[tB[bi]] + (long)q.st_b[tA[bi]] * U;
        }
    }
    mw<K> gx[MB], gy[MB];
    mw<DK> l1[MB], l2[MB];
#pragma unroll
    for (int bi = 0; bi < MB; bi++) {
        if (kind[bi] == 0 && has[bi]) {
            gx[bi] = ldx<K>(q.GX + kb[bi]->g_off, q.glen, ix[bi]);
            gy[bi] = ldx<K>(q.GY + kb[bi]->g_off, q.glen, iy[bi]);
            l1[bi] = ldx<DK>(q.st_lam, q.lamp, tA[bi]);
            l2[bi] = ldx<DK>(q.st_lam, q.lamp, tB[bi]);
        } else if (has[bi]) {
            gx[bi] = ldx<K>(q.Sd + kb[bi]->sd_off, q.sdlen, tA[bi] + (long)tB[bi] * kb[bi]->cnt);
        }
    }
    acc<K> s;
    acc_zero<K>(s);
#pragma unroll
    for (int bi = 0; bi < MB; bi++) {
        if (!has[bi]) continue;
        if (kind[bi] == 0) {
            constexpr int LL = (2 * DK + 1 < K) ? 2 * DK + 1 : K;
            acc_fma<K, K, LL>(s, mul<K>(gx[bi], gy[bi]), mulx<LL, DK, DK>(l1[bi], l2[bi]));
        } else acc_add<K, K>(s, gx[bi]);
    }
    const mw<K> v = acc_result<K>(s);
    stx<K>(q.S + c.Soff, q.Slen, pp + (long)qq * P, v);
    stx<K>(q.S + c.Soff, q.Slen, qq + (long)pp * P, v);

}

// ---------------------------------------------------------------------------------------------------------------------
// Factorisation of a cluster: L_j = chol(S_j) (in place in the S buffer), LinvB_j = L_j^-1 B_j; the scaled triangles of L_j
// for the solve stage.
// ---------------------------------------------------------------------------------------------------------------------
// KS: limbs of the arrays (S, S0, Si and the LDS copies M, W, whose planes are laid out for KS); K <= KS: limbs of the elimination (MwDev::kf) -- the upper
// KS - K planes of L_j, 1 / diag and L_j^-1 are stored as zeros
template <int KS, int K, class PM, class PW>
__device__ __forceinline__ bool mw_factor_body(const MwDev &q, const MwClu &c, int j, PM M, long mplane, PW W, mwk::lds_d *bc, int tid, int cw, int cnw) {
    using namespace mwk;
    const int P = c.P;
    // (cw of cnw: the workgroups of a cluster share out the columns of L_j^-1)
    const bool ok = wg_potrf<K, true, MW_PT>(M, mplane, P, P, q.srd + c.coff, q.xlen, W, MW_TRI(P), 0, bc, tid, cw, cnw);     // W packed
    if (!ok && tid == 0) atomicMin(&q.info[0], j + 1);
    if (K < KS && ok && cw == 0) for (int i = tid; i < P; i += MW_PT) {
#pragma unroll
        for (int l = K; l < KS; l++) q.srd[(long)l * q.xlen + c.coff + i] = 0.0;
    }
    if (ok) {
        for (int e = tid; e < P * P; e += MW_PT) {        // this workgroup's columns of L_j^-1, zero above the diagonal
            const int i = e % P, cc = e / P;
            if (cc % cnw != cw) continue;
#pragma unroll
            for (int l = 0; l < KS; l++) q.Si[(long)l * q.Slen + c.Soff + e] = (i >= cc && l < K) ? (double)W[(long)l * MW_TRI(P) + w_index(i, cc, P, 0)] : 0.0;
        }
    }
    // L_j goes back to the S buffer, which is also the input: only once every workgroup of the cluster has read it, i.e. by the last one
    if (!wg_last_block(&q.pcnt[j], cnw) || !ok) return ok;
    double *Sg = q.S + c.Soff;
    for (int e = tid; e < P * P; e += MW_PT) {
        const int i = e % P, cc = e / P;
#pragma unroll
        for (int l = 0; l < KS; l++) Sg[(long)l * q.Slen + e] = (i >= cc && l < K) ? (double)M[(long)l * mplane + e] : 0.0;
    }
    return true;
}
template <int K>
__device__ __forceinline__ void mw_factor_cluster(const MwDev &q, int j, int cw, int cnw) {
    using namespace mwk;
    const int tid = threadIdx.x;
    const MwClu &c = q.clu[j];
    const int P = c.P;
    lds_d *bc = MW_LDS;
    if (c.lds) {
        lds_d *M = MW_LDS + MW_POTRF_SCR(K, P);
        wg_copy<K, MW_PT>(M, (long)P * P, P, q.S + c.Soff, q.Slen, P, P, P, tid);
        __syncthreads();
        if (cw == 0) wg_copy<K, MW_PT>(q.S0 + c.Soff, q.Slen, P, M, (long)P * P, P, P, P, tid);      // S_j as assembled, for the residuals of the refined solve
        if constexpr (mw_kf_of(K) < K) {
            if (q.kf < K) { mw_factor_body<K, mw_kf_of(K)>(q, c, j, M, (long)P * P, M + (long)K * P * P, bc, tid, cw, cnw); return; }
        }
        mw_factor_body<K, K>(q, c, j, M, (long)P * P, M + (long)K * P * P, bc, tid, cw, cnw);
    }
    // clusters too large for LDS are factored by the blocked, multi-workgroup path (k_mw_bp_*, driven by the host)
}
template <int K>
__global__ __launch_bounds__(MW_PT) void k_mw_factor(const MwDev q) { mw_mark(q); mw_factor_cluster<K>(q, blockIdx.x, blockIdx.y, gridDim.y); }

// S0_j = S_j for the clusters factored by the blocked path (the LDS kernels copy as they load): grid (tiles, J)
template <int K>
__global__ __launch_bounds__(MW_NT) void k_mw_keep_S(const MwDev q) {
    const MwClu &c = q.clu[blockIdx.y];
    if (c.lds) return;
    const long nn = (long)c.P * c.P;
    for (long e = (long)blockIdx.x * MW_NT + threadIdx.x; e < nn; e += (long)gridDim.x * MW_NT) {
#pragma unroll
        for (int l = 0; l < K; l++) q.S0[(long)l * q.Slen + c.Soff + e] = q.S[(long)l * q.Slen + c.Soff + e];
    }
}

// LinvB_j = L_j^-1 B_j (src/solver.jl:1256-1261) = Si_j B_j: a product with the explicit inverse, four lanes per entry.  KA <= K: limbs of the product (MwDev::kf)
#define MW_LBI_W 4
template <int K, int KA, int DK>
__device__ __forceinline__ void mw_linvb_body(const MwDev &q) {
    using namespace mwk;
    const int j = blockIdx.y;
    const MwClu &c = q.clu[j];
    const int P = c.P, N = q.N;
    if (blockIdx.x * (MW_NT / MW_LBI_W) >= P * N) return;
    if (q.info[0] != MW_INFO_NONE && q.info[0] <= j + 1) return;        // this cluster (or an earlier one) failed
    const int e = blockIdx.x * (MW_NT / MW_LBI_W) + threadIdx.x / MW_LBI_W, sub = threadIdx.x % MW_LBI_W;
    const bool live = e < P * N;
    const int ee = live ? e : 0, i = ee % P, a = ee / P;
    const double *Ti = q.Si + c.Soff;
    acc<KA> s;
    acc_zero<KA>(s);
    for (int r = sub; r <= i; r += MW_LBI_W) acc_fma<KA, KA, DK>(s, ldx<KA>(Ti, q.Slen, i + (long)r * P), ldx<DK>(q.B, q.Bp, c.coff + r + (long)a * q.xlen));
    const mw<KA> v = lanes_sum<KA, MW_LBI_W>(acc_result<KA>(s));
    if (live && sub == 0) stx<K>(q.LB, q.xlen * (long)N, c.coff + i + (long)a * q.xlen, cvt<K, KA>(v));
}
template <int K, int DK>
__global__ __launch_bounds__(MW_NT) void k_mw_linvb(const MwDev q) {
    MW_CHAIN_STAMP(q, 0, blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0);
    if constexpr (mw_kf_of(K) < K) { if (q.kf < K) { mw_linvb_body<K, mw_kf_of(K), DK>(q); return; } }
    mw_linvb_body<K, K, DK>(q);
}

// Q = sum_j LinvB_j^T LinvB_j = LB^T LB over the stacked rows (src/solver.jl:1264-1271): eight lanes per entry a >= b
#define MW_Q_W 8
template <int K, int KA, int W>
__device__ __forceinline__ void mw_qgram_body(const MwDev &q) {
    using namespace mwk;
    const int N = q.N;
    const int e = blockIdx.x * (MW_NT / W) + threadIdx.x / W, sub = threadIdx.x % W;
    const int tot = N * (N + 1) / 2;
    if (blockIdx.x * (MW_NT / W) >= tot) return;
    const bool live = e < tot;
    int a, b;
    tri_index(live ? e : 0, a, b);
    const long plane = q.xlen * (long)N;
    acc<KA> s;
    acc_zero<KA>(s);
    for (long r = sub; r < q.xlen; r += W) acc_fma<KA, KA, KA>(s, ldx<KA>(q.LB, plane, r + a * q.xlen), ldx<KA>(q.LB, plane, r + b * q.xlen));
    const mw<K> v = cvt<K, KA>(lanes_sum<KA, W>(acc_result<KA>(s)));
    if (live && sub == 0) {
        double *Qp = q.Qg + (long)q.rank * K * N * N;                 // this rank's partial sum over its clusters
        stx<K>(Qp, (long)N * N, a + (long)b * N, v);
        stx<K>(Qp, (long)N * N, b + (long)a * N, v);
    }
}
// lanes: 8, or 32 while the launch stays small (a few hundred entries: more workgroups with a quarter of the multiply-adds per lane)
template <int K>
__global__ __launch_bounds__(MW_NT) void k_mw_qgram(const MwDev q, int lanes) {
    MW_CHAIN_STAMP(q, 1, blockIdx.x == 0 && threadIdx.x == 0);
    MW_CHAIN_STAMP(q, 2, blockIdx.x == gridDim.x - 1 && threadIdx.x == 0);
    bool done = false;
    if constexpr (mw_kf_of(K) < K) {
        if (q.kf < K) {
            if (lanes == 32) mw_qgram_body<K, mw_kf_of(K), 32>(q);
            else mw_qgram_body<K, mw_kf_of(K), MW_Q_W>(q);
            done = true;
        }
    }
    if (!done) {
        if (lanes == 32) mw_qgram_body<K, K, 32>(q);
        else mw_qgram_body<K, K, MW_Q_W>(q);
    }
    MW_CHAIN_STAMP(q, 5, blockIdx.x == gridDim.x - 1 && threadIdx.x == 0);       // (end of the last workgroup)
}

template <int K>
__device__ __forceinline__ void mw_solve_fwd_cluster(const MwDev &q, int j, const double *__restrict__ rhs_x);      // (solve stage, below)
// Cholesky of Q (src/solver.jl:1274) and its scaled triangles.  KS / K: limbs of the arrays / of the elimination, as mw_factor_body
template <int KS, int K, class PM, class PW>
__device__ __forceinline__ void mw_potrf_q_body(const MwDev &q, PM M, long plane, PW W, mwk::lds_d *bc, int tid, int cw, int cnw) {
    using namespace mwk;
    const int N = q.N;
    // (cw of cnw: the workgroups share out the columns of L_Q^-1; the first one also writes L_Q)
    const bool ok = wg_potrf<K, true, MW_PT>(M, plane, N, N, q.qrd, N, W, MW_TRI(N), 0, bc, tid, cw, cnw);     // W packed
    if (!ok) {
        if (tid == 0) atomicMin(&q.info[0], q.J + 1);
        return;
    }
    if (K < KS && cw == 0) for (int i = tid; i < N; i += MW_PT) {
#pragma unroll
        for (int l = K; l < KS; l++) q.qrd[(long)l * N + i] = 0.0;
    }
    for (int e = tid; e < N * N; e += MW_PT) {
        const int i = e % N, cc = e / N;
#pragma unroll
        for (int l = 0; l < KS; l++) {
            if (cw == 0) q.Q[(long)l * N * N + e] = (i >= cc && l < K) ? (double)M[(long)l * plane + e] : 0.0;
            if (cc % cnw == cw) q.Qi[(long)l * N * N + e] = (i >= cc && l < K) ? (double)W[(long)l * MW_TRI(N) + w_index(i, cc, N, 0)] : 0.0;
        }
    }
}
// wait (one lane polls, bounded) until *word >= value, then make what the signalling stream wrote before it visible to this workgroup
// store q.mark_value to q.mark_word (first thread of the launch): see MwDev::mark_word
__device__ __forceinline__ void mw_mark(const MwDev &q) {
    if (q.mark_word && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)
        __hip_atomic_store(q.mark_word, q.mark_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (a mark that never comes -- it cannot while kernels of different queues may run side by side: the marking kernel is enqueued first; a profiler that
// serialises kernels, e.g. rocprofv3 --pmc, breaks that: use CLRS_MW_STREAM_WORDS=0 there -- ends the wait after MW_WAIT_TICKS of WALL CLOCK, however long the
// polls take, and leaves MW_INFO_TIMEOUT in *info: the iteration then reports error code 5 (stream synchronisation timed out), not a failed block)
__device__ __forceinline__ void mw_wait_word(const int *word, int value, int *info, int code) {
    (void)code;
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        unsigned long long t0 = 0;
        bool late = false;
        while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < value) {
            __builtin_amdgcn_s_sleep(8);
            if ((++spins & 4095u) == 0) {                      // the clock is looked at every few thousand polls only
                const unsigned long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                else if (now - t0 > MW_WAIT_TICKS) { late = true; break; }
            }
        }
        if (late) atomicMin(info, MW_INFO_TIMEOUT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}
template <int K>
__global__ __launch_bounds__(MW_PT) void k_mw_potrf_q(const MwDev q, int nq, const double *__restrict__ fwd_rhs, const int *__restrict__ wait_word, int wait_value) {
    using namespace mwk;
    const int N = q.N, tid = threadIdx.x;
    MW_CHAIN_STAMP(q, 3, blockIdx.x == 0 && threadIdx.x == 0);
    MW_CHAIN_STAMP(q, 4, (int)blockIdx.x == nq && threadIdx.x == 0);
    // workgroups behind the nq of Q (the interior-point iteration adds them): the first product pair of the next solve, t_j = L_j^-1 rhs_x[j] and
    // u_j = LinvB_j^T t_j per cluster, which needs the clusters' factors only -- beside the 31 pivots of Q instead of behind them.  Their right-hand
    // side comes from another stream: they wait for its mark here (wait_word), not the whole launch in front of an event
    if ((int)blockIdx.x >= nq) {
        if (wait_word) mw_wait_word(wait_word, wait_value, &q.info[0], q.J + 1);
        if ((int)blockIdx.x >= nq + q.J) {           // the second right-hand side (MwDev::ride2_rhs)
            MwDev q2 = q;
            q2.t = q.ride2_t; q2.u = q.ride2_u;
            mw_solve_fwd_cluster<K>(q2, blockIdx.x - nq - q.J, q.ride2_rhs);
        } else mw_solve_fwd_cluster<K>(q, blockIdx.x - nq, fwd_rhs);
        return;
    }
    if (q.info[0] != MW_INFO_NONE) return;           // a cluster failed: the reference throws before reaching Q
    lds_d *bc = MW_LDS;
    const long nn = (long)N * N;
    lds_d *M = MW_LDS + MW_POTRF_SCR(K, N);
    for (int e = tid; e < nn; e += MW_PT) {                // Q = sum over the ranks' partial sums, in rank order on every rank
        acc<K> s;
        acc_zero<K>(s);
        for (int r = 0; r < q.world; r++) acc_add<K, K>(s, ldx<K>(q.Qg + (long)r * K * nn, nn, e));
        stx<K>(M, nn, e, acc_result<K>(s));
    }
    __syncthreads();
    if constexpr (mw_kf_of(K) < K) {
        if (q.kf < K) { mw_potrf_q_body<K, mw_kf_of(K)>(q, M, nn, M + (long)K * nn, bc, tid, blockIdx.x, nq); return; }
    }
    mw_potrf_q_body<K, K>(q, M, nn, M + (long)K * nn, bc, tid, blockIdx.x, nq);
    // a Q too large for LDS: k_mw_qsum + the blocked path (k_mw_bp_*)
}

// Q = sum over the ranks' partial sums (rank order), for the blocked factorisation of a Q that does not fit in LDS
template <int K>
__global__ __launch_bounds__(MW_NT) void k_mw_qsum(const MwDev q) {
    using namespace mwk;
    const long nn = (long)q.N * q.N;
    for (long e = (long)blockIdx.x * MW_NT + threadIdx.x; e < nn; e += (long)gridDim.x * MW_NT) {
        acc<K> s;
        acc_zero<K>(s);
        for (int r = 0; r < q.world; r++) acc_add<K, K>(s, ldx<K>(q.Qg + (long)r * K * nn, nn, e));
        stx<K>(q.Q, nn, e, acc_result<K>(s));
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Blocked Cholesky AND inverse factor of a matrix that does not fit twice in LDS (clusters with P > 44 at 5 limbs, Q with
// N > 44), in place in global memory, panel width MW_PB, over many workgroups:
//   k_mw_bp_diag   the diagonal block and its inverse by one workgroup in LDS (wg_potrf with the [M | I] elimination);
//   k_mw_bp_panel  the rows below it, L_panel = A_panel M_d^T: a product with the inverse of the diagonal block, MW_BP_PR rows
//                  per workgroup, eight lanes per entry (the rows are independent);
//   k_mw_bp_syrk   the trailing matrix, MW_BP_SW lanes per entry of its lower triangle (MW_PB-term accumulator dot products);
//   k_mw_bp_inv    after the last block column: the off-diagonal blocks of L^-1 by block distance d = 1, 2, ...:
//                  (L^-1)_ji = -(L^-1)_jj sum_{i <= k < j} L_jk (L^-1)_ki, independent column by column: one workgroup
//                  per pair (j, i) and column, sixteen lanes per entry;
//   k_mw_bp_finish zero strict upper triangle of L.
// With L^-1 explicit, LinvB and every solve of a large cluster are products over many lanes, like those of a small one.
// ---------------------------------------------------------------------------------------------------------------------
#define MW_PB_OF(K) 32       // panel width: a MW_PB x MW_PB matrix and the packed inverse of its factor, K limbs each, fit in LDS up to K = 10 (k_mw_bp_diag)
#ifndef MW_BP_PR
#define MW_BP_PR 1           // rows of the panel per workgroup: sixteen lanes per entry (the chain of launches per block column is latency bound: few multiply-adds per
#endif                       // lane matter more than full lanes; two rows / eight lanes until the end of round 5: Nsphere_packing N = 3 1.869 -> 1.859 ms per iteration)
#define MW_BP_IC 1           // columns of an inverse block per workgroup: sixteen lanes per entry
#ifndef MW_BP_SW
#define MW_BP_SW 16          // lanes per entry of the trailing update (four until the end of round 5: eight dependent multiply-adds with two loads from memory each per lane;
                             // sixteen: Nsphere_packing N = 3 1.891 -> 1.862 ms per iteration, N = 2 1.083 -> 1.059, three-point as named 2.068 -> 2.055, SDPA x64 1.647 -> 1.631)
#endif
struct MwBp {                // one matrix being factored: planar M and its inverse factor Mi (same plane length and leading dimension), reciprocal diagonal rd
    double *M, *Mi, *rd;
    mwi64 plane, rdplane;
    int n, ld, code, which;            // info[which] = code on failure
    int slot, pad;                     // arrival counter pcnt[J + 1 + slot] of the workgroups that share a diagonal block
};
// (the blocked path with MwDev::kf < K: every kernel below reads KA = kf planes of the matrices it works on, in place, and stores its results with the upper
// K - KA planes zero -- what the trailing matrix has not been through yet still holds the K-limb input, read as its truncation)
template <int K, int KA>
__device__ __forceinline__ void mw_bp_diag_body(const MwDev &q, const MwBp &m, int j0) {
    using namespace mwk;
    constexpr int MW_PB = MW_PB_OF(K);
    const int nb = min(MW_PB, m.n - j0), tid = threadIdx.x;
    lds_d *scr = MW_LDS, *D = MW_LDS + MW_POTRF_SCR(K, MW_PB), *W = D + (long)K * MW_PB * MW_PB, *rdl = W + (long)K * MW_TRI(MW_PB);
    const int cw = blockIdx.x, cnw = gridDim.x;         // the workgroups share out the columns of the inverse of the diagonal block
    // a failure recorded earlier (an earlier block column, or a sibling workgroup of this one that ran first) skips the work but not
    // the arrival count below, which every workgroup of the launch must reach
    bool ok = false;
    if (q.info[m.which] == MW_INFO_NONE) {
        wg_copy<KA, MW_PT>(D, (long)nb * nb, nb, m.M + j0 + (long)j0 * m.ld, m.plane, m.ld, nb, nb, tid);
        __syncthreads();
        ok = wg_potrf<KA, true, MW_PT>(D, (long)nb * nb, nb, nb, rdl, nb, W, MW_TRI(nb), 0, scr, tid, cw, cnw);     // W packed
        if (!ok && tid == 0) atomicMin(&q.info[m.which], m.code);
    }
    if (ok) {
        for (int e = tid; e < nb * nb; e += MW_PT) {
            const int i = e % nb, c = e / nb;
            if (c % cnw == cw) stx<K>(m.Mi, m.plane, (j0 + i) + (long)(j0 + c) * m.ld, i >= c ? cvt<K, KA>(ldx<KA>(W, MW_TRI(nb), w_index(i, c, nb, 0))) : zero<K>());
        }
    }
    // the factor overwrites its input: by the workgroup that finishes last, when all have read it
    if (!wg_last_block(&q.pcnt[q.J + 1 + m.slot], cnw) || !ok) return;
    for (int e = tid; e < nb * nb; e += MW_PT) {
        const int i = e % nb, c = e / nb;
        stx<K>(m.M, m.plane, (j0 + i) + (long)(j0 + c) * m.ld, i >= c ? cvt<K, KA>(ldx<KA>(D, (long)nb * nb, e)) : zero<K>());
    }
    for (int i = tid; i < nb; i += MW_PT) stx<K>(m.rd, m.rdplane, j0 + i, cvt<K, KA>(ldx<KA>(rdl, nb, i)));
}
template <int K>
__global__ __launch_bounds__(MW_PT) void k_mw_bp_diag(const MwDev q, const MwBp *__restrict__ ms, int nm, int j0) {
    mw_mark(q);
    using namespace mwk;
    // rows of the grid beyond the nm matrices of this launch (the first launch of the factorisation of the S_j adds them): the clusters that fit
    // in LDS, factored beside the first diagonal block of the large ones instead of in a launch of their own in front of it
    if ((int)blockIdx.y >= nm) { mw_factor_cluster<K>(q, blockIdx.y - nm, blockIdx.x, gridDim.x); return; }
    const MwBp m = ms[blockIdx.y];
    if (j0 >= m.n) return;
    if constexpr (mw_kf_of(K) < K) { if (q.kf < K) { mw_bp_diag_body<K, mw_kf_of(K)>(q, m, j0); return; } }
    mw_bp_diag_body<K, K>(q, m, j0);
}
// rows below the diagonal block: L[r, c] = sum_{k <= c} A[r, k] M_d[c, k]
template <int K, int KA>
__device__ __forceinline__ void mw_bp_panel_body(const MwDev &q, const MwBp &m, int j0) {
    using namespace mwk;
    constexpr int MW_PB = MW_PB_OF(K);
    if (j0 >= m.n || q.info[m.which] != MW_INFO_NONE) return;
    const int nb = min(MW_PB, m.n - j0), tid = threadIdx.x;
    const int r0 = j0 + nb + blockIdx.x * MW_BP_PR;
    if (r0 >= m.n) return;
    const int nr = min(MW_BP_PR, m.n - r0);
    lds_d *At = MW_LDS;                                     // the MW_BP_PR x nb tile of A, read before any of it is overwritten
    const long ap = (long)MW_BP_PR * MW_PB;
    for (int e = tid; e < MW_BP_PR * nb; e += MW_PT) {
        const int r = e % MW_BP_PR, c = e / MW_BP_PR;
        if (r < nr) stx<KA>(At, ap, e, ldx<KA>(m.M, m.plane, (r0 + r) + (long)(j0 + c) * m.ld));
    }
    __syncthreads();
    constexpr int LP = MW_PT / (MW_BP_PR * MW_PB);          // lanes per entry
    const int sub = tid % LP;
    for (int e0 = 0; e0 < MW_BP_PR * nb; e0 += MW_PT / LP) {
        const int e = e0 + tid / LP;
        const bool live = e < MW_BP_PR * nb && (e % MW_BP_PR) < nr;
        const int ee = live ? e : 0, r = ee % MW_BP_PR, c = ee / MW_BP_PR;
        acc<KA> s;
        acc_zero<KA>(s);
        for (int kk = sub; kk <= c; kk += LP) acc_fma<KA, KA, KA>(s, ldx<KA>(At, ap, r + (long)kk * MW_BP_PR), ldx<KA>(m.Mi, m.plane, (j0 + c) + (long)(j0 + kk) * m.ld));
        const mw<KA> v = lanes_sum<KA, LP>(acc_result<KA>(s));
        if (live && sub == 0) stx<K>(m.M, m.plane, (r0 + r) + (long)(j0 + c) * m.ld, cvt<K, KA>(v));
    }
}
template <int K>
__global__ __launch_bounds__(MW_PT) void k_mw_bp_panel(const MwDev q, const MwBp *__restrict__ ms, int j0) {
    const MwBp m = ms[blockIdx.y];
    if constexpr (mw_kf_of(K) < K) { if (q.kf < K) { mw_bp_panel_body<K, mw_kf_of(K)>(q, m, j0); return; } }
    mw_bp_panel_body<K, K>(q, m, j0);
}
// trailing update: A[i, j] -= sum_c L[i, j0 + c] L[j, j0 + c], i >= j >= j0 + nb
// (split into the first MW_PB columns -- all the next diagonal block and panel need -- and a rest that rides on the next diagonal block's launch: measured,
// no gain: the launch of the first columns takes what the whole update takes, a 32-term dot product over the lanes of an entry, whatever the entry count)
template <int K, int KA>
__device__ __forceinline__ void mw_bp_syrk_body(const MwDev &q, const MwBp &m, int j0) {
    using namespace mwk;
    constexpr int MW_PB = MW_PB_OF(K);
    if (j0 >= m.n || q.info[m.which] != MW_INFO_NONE) return;
    const int nb = min(MW_PB, m.n - j0), t0 = j0 + nb, mm = m.n - t0;
    const long tot = (long)mm * (mm + 1) / 2;
    if ((long)blockIdx.x * (MW_NT / MW_BP_SW) >= tot) return;      // uniform over the workgroup
    const long e = (long)blockIdx.x * (MW_NT / MW_BP_SW) + threadIdx.x / MW_BP_SW;
    const int sub = threadIdx.x % MW_BP_SW;
    const bool live = e < tot;
    int ii, jj;
    tri_index(live ? (int)e : 0, ii, jj);
    const int i = t0 + ii, j = t0 + jj;
    acc<KA> s;
    acc_zero<KA>(s);
    if (sub == 0) acc_add<KA, KA>(s, ldx<KA>(m.M, m.plane, i + (long)j * m.ld));
    for (int c = sub; c < nb; c += MW_BP_SW) acc_fma<KA, KA, KA>(s, ldx<KA>(m.M, m.plane, i + (long)(j0 + c) * m.ld), ldx<KA>(m.M, m.plane, j + (long)(j0 + c) * m.ld), -1.0);
    const mw<KA> v = lanes_sum<KA, MW_BP_SW>(acc_result<KA>(s));
    if (live && sub == 0) stx<K>(m.M, m.plane, i + (long)j * m.ld, cvt<K, KA>(v));
}
template <int K>
__global__ __launch_bounds__(MW_NT) void k_mw_bp_syrk(const MwDev q, const MwBp *__restrict__ ms, int j0) {
    const MwBp m = ms[blockIdx.y];
    if constexpr (mw_kf_of(K) < K) { if (q.kf < K) { mw_bp_syrk_body<K, mw_kf_of(K)>(q, m, j0); return; } }
    mw_bp_syrk_body<K, K>(q, m, j0);
}
// blocks (j, i) of L^-1 with j - i = d: T = sum_{i <= k < j} L_jk (L^-1)_ki (the block columns between are contiguous: one
// sum over the rows i MW_PB .. j MW_PB - 1), then (L^-1)_ji = -(L^-1)_jj T
template <int K, int KA>
__device__ __forceinline__ void mw_bp_inv_block_ka(const MwDev &q, const MwBp &m, int bi, int bj, int cblk) {
    using namespace mwk;
    constexpr int MW_PB = MW_PB_OF(K);
    if (q.info[m.which] != MW_INFO_NONE) return;
    const int tid = threadIdx.x;
    if (bj * MW_PB >= m.n) return;                          // (a matrix with fewer block columns than the largest of the launch)
    const int ci0 = bi * MW_PB, ni = min(MW_PB, m.n - ci0), rj0 = bj * MW_PB, nj = min(MW_PB, m.n - rj0);
    const int c0 = cblk * MW_BP_IC;
    if (c0 >= ni) return;
    const int pc = min(MW_BP_IC, ni - c0);
    lds_d *T = MW_LDS;
    const long tp = (long)MW_PB * MW_BP_IC;
    constexpr int LW = MW_PT / (MW_PB * MW_BP_IC);             // lanes per entry: 16
    const int e = tid / LW, sub = tid % LW, r = e % MW_PB, cl = e / MW_PB;
    const bool live = r < nj && cl < pc;
    const int rr = live ? r : 0, col = ci0 + c0 + (live ? cl : 0);
    {
        acc<KA> s;
        acc_zero<KA>(s);
        for (int t = col + sub; t < rj0; t += LW)              // rows of (L^-1)[:, col] above its diagonal are zero
            acc_fma<KA, KA, KA>(s, ldx<KA>(m.M, m.plane, (rj0 + rr) + (long)t * m.ld), ldx<KA>(m.Mi, m.plane, t + (long)col * m.ld));
        const mw<KA> v = lanes_sum<KA, LW>(acc_result<KA>(s));
        if (sub == 0) stx<KA>(T, tp, e, live ? v : zero<KA>());
    }
    __syncthreads();
    {
        acc<KA> s;
        acc_zero<KA>(s);
        for (int t = sub; t <= rr; t += LW) acc_fma<KA, KA, KA>(s, ldx<KA>(m.Mi, m.plane, (rj0 + rr) + (long)(rj0 + t) * m.ld), ldx<KA>(T, tp, t + (long)(live ? cl : 0) * MW_PB), -1.0);
        const mw<KA> v = lanes_sum<KA, LW>(acc_result<KA>(s));
        if (live && sub == 0) stx<K>(m.Mi, m.plane, (rj0 + r) + (long)col * m.ld, cvt<K, KA>(v));
    }
}
template <int K>
__device__ __forceinline__ void mw_bp_inv_block(const MwDev &q, const MwBp &m, int bi, int bj, int cblk) {
    if constexpr (mw_kf_of(K) < K) { if (q.kf < K) { mw_bp_inv_block_ka<K, mw_kf_of(K)>(q, m, bi, bj, cblk); return; } }
    mw_bp_inv_block_ka<K, K>(q, m, bi, bj, cblk);
}
template <int K>
__global__ __launch_bounds__(MW_PT) void k_mw_bp_inv(const MwDev q, const MwBp *__restrict__ ms, int d) { mw_bp_inv_block<K>(q, ms[blockIdx.z], blockIdx.x, blockIdx.x + d, blockIdx.y); }
// the same by block ROWS: the blocks (row, i), i < row, need the rows above (and the inverse of the row's diagonal block): row j can be formed as soon as the
// diagonal block j is, beside the rest of the factorisation -- k_mw_bp_diag_pipe carries row j on the launch of diagonal block j + 1, this kernel takes the last row
template <int K>
__global__ __launch_bounds__(MW_PT) void k_mw_bp_inv_row(const MwDev q, const MwBp *__restrict__ ms, int row) { mw_bp_inv_block<K>(q, ms[blockIdx.z], blockIdx.x, row, blockIdx.y); }
// zero strict upper triangle of L (the blocks above the diagonal still hold the symmetric input)
template <int K>
__global__ __launch_bounds__(MW_NT) void k_mw_bp_finish(const MwDev q, const MwBp *__restrict__ ms) {
    using namespace mwk;
    constexpr int MW_PB = MW_PB_OF(K);
    const MwBp m = ms[blockIdx.y];
    if (q.info[m.which] != MW_INFO_NONE) return;
    const long nn = (long)m.n * m.n;
    for (long e = (long)blockIdx.x * MW_NT + threadIdx.x; e < nn; e += (long)gridDim.x * MW_NT) {
        const int i = (int)(e % m.n), k = (int)(e / m.n);
        if (i / MW_PB < k / MW_PB) stx<K>(m.M, m.plane, i + (long)k * m.ld, zero<K>());
    }
}

// partial u of this rank: sum of its clusters' u_j into its gather slot (sharded solve only)
template <int K>
__global__ __launch_bounds__(MW_NT) void k_mw_usum(const MwDev q) {
    using namespace mwk;
    const int N = q.N;
    for (int a = blockIdx.x * MW_NT + threadIdx.x; a < N; a += gridDim.x * MW_NT) {
        acc<K> s;
        acc_zero<K>(s);
        for (int j = 0; j < q.J; j++) acc_add<K, K>(s, ldx<K>(q.u, (long)q.J * N, (long)j * N + a));
        if (q.uadd) acc_add<K, K>(s, ldx<K>(q.uadd, N, a));
        stx<K>(q.ug + (long)q.rank * K * N, N, a, acc_result<K>(s));
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Solve stage (src/solver.jl:1527-1582), three launches: per cluster t_j = L_j^-1 rhs_x[j], u_j = LinvB_j^T t_j;
// dy = Q^-1 (rhs_y - sum_j u_j); per cluster dx_j = L_j^-T (t_j + LinvB_j dy).  Every triangular solve is a product with
// the explicit inverse factor (wg_trmv_n / _t), every dot product runs over eight lanes.
// ---------------------------------------------------------------------------------------------------------------------
#define MW_S_W 8
// u_j = LinvB_j^T t_j (t in LDS, planar with plane P).  KA <= K: limbs of the products (MwDev::kf); what is stored carries K planes, the upper ones zero
template <int K, int KA>
__device__ __forceinline__ void mw_solve_u(const MwDev &q, const MwClu &c, int j, mwk::lds_d *tv, int tid) {
    using namespace mwk;
    const int P = c.P, N = q.N;
    const long plane = q.xlen * (long)N;
    const int sub = tid % MW_S_W;
    for (int a0 = 0; a0 < N; a0 += MW_NT / MW_S_W) {
        const int a = a0 + tid / MW_S_W;
        const bool live = a < N;
        const int aa = live ? a : 0;
        acc<KA> s;
        acc_zero<KA>(s);
        for (int r = sub; r < P; r += MW_S_W) acc_fma<KA, KA, KA>(s, ldx<KA>(q.LB, plane, c.coff + r + aa * q.xlen), ldx<KA>(tv, P, r));
        mw<KA> v = lanes_sum<KA, MW_S_W>(acc_result<KA>(s));
        if (live && sub == 0) stx<K>(q.u, (long)q.J * N, (long)j * N + a, cvt<K, KA>(v));
    }
}
template <int K, int KA>
__device__ __forceinline__ void mw_solve_fwd_cluster_ka(const MwDev &q, int j, const double *__restrict__ rhs_x) {
    using namespace mwk;
    const int tid = threadIdx.x;
    if (tid >= MW_NT) return;                              // (called from a launch with more threads: the other waves leave; barriers count the rest)
    const MwClu &c = q.clu[j];
    const int P = c.P;
    lds_d *tv = MW_LDS, *t2 = tv + (long)K * P;            // rhs_j and t_j, planar with plane P
    for (int i = tid; i < P; i += MW_NT) {
#pragma unroll
        for (int l = 0; l < KA; l++) tv[(long)l * P + i] = rhs_x[(long)l * q.xlen + c.coff + i];
    }
    __syncthreads();
    wg_trmv_n<KA>(q.Si + c.Soff, q.Slen, P, P, tv, P, t2, P, tid);      // t_j = Si_j rhs_j
    for (int i = tid; i < P; i += MW_NT) {
#pragma unroll
        for (int l = 0; l < K; l++) q.t[(long)l * q.xlen + c.coff + i] = l < KA ? (double)t2[(long)l * P + i] : 0.0;
    }
    mw_solve_u<K, KA>(q, c, j, t2, tid);
}
template <int K>
__device__ __forceinline__ void mw_solve_fwd_cluster(const MwDev &q, int j, const double *__restrict__ rhs_x) {
    if constexpr (mw_kf_of(K) < K) { if (q.kf < K) { mw_solve_fwd_cluster_ka<K, mw_kf_of(K)>(q, j, rhs_x); return; } }
    mw_solve_fwd_cluster_ka<K, K>(q, j, rhs_x);
}
template <int K>
__global__ __launch_bounds__(MW_NT) void k_mw_solve_fwd(const MwDev q, const double *__restrict__ rhs_x) { mw_solve_fwd_cluster<K>(q, blockIdx.x, rhs_x); }

// Limbs of the CORRECTION of the refined solve when the caller opts for fewer than K (MODE 1 / 2 below; refine = 2).  The correction is smaller than the
// solution by the backward error of the first pass, 2^(lam - 53 K) with lam the bits the inverse-factor products lose (cohnelkies(8,15): 57,
// Nsphere_packing(8,15): 82 on mid-trajectory iterates), and is itself computed to 2^(lam - 52 KC) of its size: the sum is good to the working precision
// while 2 lam <= 52 KC.  NOT the default: on the last iterates of Nsphere_packing(8,15,[1/2,1/2,1/2]) lam exceeds 100 bits and a 3-limb correction is
// worse than none (measured, scripts/sharded_debug.py) -- the default correction carries all K limbs (-25 us per iteration on the named problem otherwise).
__host__ __device__ constexpr int mw_kc(int K) { return K <= 3 ? K : K <= 6 ? 3 : K / 2; }

// dy = Q^-1 (rhs_y - sum_j u_j) into v (LDS, plane N; y: N more numbers of scratch).  The difference in K limbs (with the u_j of the refinement it
// cancels to the size of the residual), the two products with the explicit inverse of L_Q in KC.
template <int K, int KC>
__device__ __forceinline__ void mw_solve_mid_body(const MwDev &q, const double *__restrict__ rhs_y, mwk::lds_d *v, mwk::lds_d *y, int tid, const mwa::mw<K> *aff_mu = nullptr) {
    using namespace mwk;
    const int N = q.N;
    const long lplane = (long)N * N;
    for (int a = tid; a < N; a += MW_NT) {
        acc<K> s;
        acc_zero<K>(s);
        acc_add<K, K>(s, ldx<K>(rhs_y, N, a));
        if (q.gathered) {
            for (int r = 0; r < q.world; r++) acc_add<K, K>(s, ldx<K>(q.ug + (long)r * K * N, N, a), -1.0);
        } else {
            for (int j = 0; j < q.J; j++) acc_add<K, K>(s, ldx<K>(q.u, (long)q.J * N, (long)j * N + a), -1.0);
            if (aff_mu) for (int j = 0; j < q.J; j++) acc_fma<K, K, K>(s, ldx<K>(q.aff_u, (long)q.J * N, (long)j * N + a), *aff_mu, -1.0);
            if (q.uadd) acc_add<K, K>(s, ldx<K>(q.uadd, N, a), -1.0);
        }
        stx<K>(v, N, a, acc_result<K>(s));
    }
    __syncthreads();
    wg_trmv_n<KC>(q.Qi, lplane, N, N, v, N, y, N, tid);     // dy = Qi^T (Qi v): two products with the explicit inverse of L_Q
    wg_trmv_t<KC>(q.Qi, lplane, N, N, y, N, v, N, tid);
}
// KC < K: the correction dy' of the refined solve, to dy (a buffer of its own: k_mw_solve_bwd<.., 2> adds it)
template <int K, int KC>
__device__ __forceinline__ void mw_solve_mid_kernel_body(const MwDev &q, const double *__restrict__ rhs_y, double *__restrict__ dy) {
    using namespace mwk;
    const int N = q.N, tid = threadIdx.x;
    lds_d *v = MW_LDS, *y = v + (long)K * N;  // N numbers each, plane N
    mw_solve_mid_body<K, KC>(q, rhs_y, v, y, tid);
    for (int a = tid; a < N; a += MW_NT) {
#pragma unroll
        for (int l = 0; l < K; l++) dy[(long)l * N + a] = l < KC ? (double)v[(long)l * N + a] : 0.0;
    }
}
template <int K, int KC>
__global__ __launch_bounds__(MW_NT) void k_mw_solve_mid(const MwDev q, const double *__restrict__ rhs_y, double *__restrict__ dy) {
    if constexpr (KC == K && mw_kf_of(K) < K) { if (q.kf < K) { mw_solve_mid_kernel_body<K, mw_kf_of(K)>(q, rhs_y, dy); return; } }
    mw_solve_mid_kernel_body<K, KC>(q, rhs_y, dy);
}

// mid_rhs_y != null (small unsharded systems): every workgroup forms dy = Q^-1 (rhs_y - sum u_j) itself first (two 31-row products: cheaper than the
// launch of k_mw_solve_mid in front of this kernel); the first one writes it to dy.
// MODE 0: the plain backward half.
// MODE 1: ... followed by the first half of the refinement step (k_mw_refine's header) for this cluster: r_j = rhs_x[j] - S_j dx_j + B_j dy (K limbs: it
//         cancels), t'_j = Si_j r_j (KC limbs) and u'_j = LinvB_j^T t'_j + B_j^T dx_j (the second term is this cluster's share of -r_y, in K limbs) into
//         q.t, q.u, where the plain forward half would have left them.
// MODE 2: the backward half of the correction in KC limbs, added to what dx, dy hold.
// KP: limbs of the products of the first pass (MODE 0, 1) -- K, or MwDev::kf with factors of fewer limbs, in which case KC = KP as well
template <int K, int KP, int KC, int DK, int MODE>
__device__ __forceinline__ void mw_solve_bwd_body(const MwDev &q, const double *__restrict__ dy_in, double *__restrict__ dx, const double *__restrict__ mid_rhs_y,
                                                  double *__restrict__ dy_out, const double *__restrict__ rhs_x) {
    using namespace mwk;
    constexpr int KB = MODE == 2 ? KC : KP;                // limbs of this pass's products
    const int j = blockIdx.x, tid = threadIdx.x;
    const MwClu &c = q.clu[j];
    const int P = c.P, N = q.N;
    lds_d *dyl = MW_LDS, *w = MW_LDS + 2L * K * N, *w2 = w + (long)K * P, *w3 = w2 + (long)K * P;
    const long plane = q.xlen * (long)N;
    const int sub = tid % MW_S_W;
    // MODE 1 with a right-hand side affine in a scalar that another stream produces (MwDev::aff_mu): wait for it here, behind the launch, then read it
    const bool aff = MODE == 1 && q.aff_mu != nullptr;
    mw<K> amu = zero<K>();
    if (aff) {
        if (q.aff_wait) mw_wait_word(q.aff_wait, q.aff_wait_value, &q.info[0], q.J + 1);
        amu = ldx<K>(q.aff_mu, q.aff_mu_plane, 0);
    }
    if (mid_rhs_y) {
        mw_solve_mid_body<K, KB>(q, mid_rhs_y, dyl, dyl + (long)K * N, tid, aff ? &amu : nullptr);      // (ends with a barrier)
    } else if (N > 0) {
        for (int a = tid; a < N; a += MW_NT) {
#pragma unroll
            for (int l = 0; l < KB; l++) dyl[(long)l * N + a] = dy_in[(long)l * N + a];
        }
        __syncthreads();
    }
    double mx_c = 0.0, mx_v = 0.0;                          // MODE 2: largest |correction|, |value| this thread has seen (heads)
    if (j == 0 && (mid_rhs_y || MODE == 2))
        for (int a = tid; a < N; a += MW_NT) {
            if (MODE == 2) {
                const mw<K> old = ldx<K>(dy_out, N, a), cor = cvt<K, KB>(ldx<KB>(dyl, N, a));
                mx_c = fmax(mx_c, __builtin_fabs(cor.l[0])); mx_v = fmax(mx_v, __builtin_fabs(old.l[0]));
                stx<K>(dy_out, N, a, add<K>(old, cor));
            } else stx<K>(dy_out, N, a, cvt<K, KB>(ldx<KB>(dyl, N, a)));
        }
    if (MODE == 2 && q.refstat && j == 0) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { mx_c = fmax(mx_c, __shfl_xor(mx_c, off, 64)); mx_v = fmax(mx_v, __shfl_xor(mx_v, off, 64)); }
        if ((tid & 63) == 0 && tid < N) { atomicMax(q.refstat + 2, (unsigned long long)__double_as_longlong(mx_c)); atomicMax(q.refstat + 3, (unsigned long long)__double_as_longlong(mx_v)); }
    }
    mx_c = mx_v = 0.0;
    for (int r0 = 0; r0 < P; r0 += MW_NT / MW_S_W) {
        const int r = r0 + tid / MW_S_W;
        const bool live = r < P;
        const int rr = live ? r : 0;
        acc<KB> s;
        acc_zero<KB>(s);
        if (sub == 0) acc_add<KB, KB>(s, ldx<KB>(q.t, q.xlen, c.coff + rr));
        if (aff && sub == 1) acc_fma<KB, KB, KB>(s, ldx<KB>(q.aff_t, q.xlen, c.coff + rr), cvt<KB, K>(amu));
        for (int a = sub; a < N; a += MW_S_W) acc_fma<KB, KB, KB>(s, ldx<KB>(q.LB, plane, c.coff + rr + a * q.xlen), ldx<KB>(dyl, N, a));
        mw<KB> v = lanes_sum<KB, MW_S_W>(acc_result<KB>(s));
        if (live && sub == 0) stx<KB>(w, P, r, v);
    }
    __syncthreads();
    wg_trmv_t<KB>(q.Si + c.Soff, q.Slen, P, P, w, P, w2, P, tid);       // dx_j = Si_j^T w
    for (int i = tid; i < P; i += MW_NT) {
        if (MODE == 2) {
            const mw<K> old = ldx<K>(dx, q.xlen, c.coff + i), cor = cvt<K, KB>(ldx<KB>(w2, P, i));
            mx_c = fmax(mx_c, __builtin_fabs(cor.l[0])); mx_v = fmax(mx_v, __builtin_fabs(old.l[0]));
            stx<K>(dx, q.xlen, c.coff + i, add<K>(old, cor));
        } else stx<K>(dx, q.xlen, c.coff + i, cvt<K, KB>(ldx<KB>(w2, P, i)));
    }
    if (MODE == 2 && q.refstat) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { mx_c = fmax(mx_c, __shfl_xor(mx_c, off, 64)); mx_v = fmax(mx_v, __shfl_xor(mx_v, off, 64)); }
        if ((tid & 63) == 0 && tid < P) { atomicMax(q.refstat + 0, (unsigned long long)__double_as_longlong(mx_c)); atomicMax(q.refstat + 1, (unsigned long long)__double_as_longlong(mx_v)); }
    }
    if (MODE != 1) return;
    // r_j = rhs_x[j] - S_j dx_j + B_j dy into w
    const double *S0 = q.S0 + c.Soff;
    for (int r0 = 0; r0 < P; r0 += MW_NT / MW_S_W) {
        const int r = r0 + tid / MW_S_W;
        const bool live = r < P;
        const int rr = live ? r : 0;
        acc<K> s;
        acc_zero<K>(s);
        if (sub == 0) acc_add<K, K>(s, ldx<K>(rhs_x, q.xlen, c.coff + rr));
        if (aff && sub == 1) acc_fma<K, K, K>(s, ldx<K>(q.aff_rhs, q.xlen, c.coff + rr), amu);
        for (int cc = sub; cc < P; cc += MW_S_W) acc_fma<K, K, KB>(s, ldx<K>(S0, q.Slen, cc + (long)rr * P), ldx<KB>(w2, P, cc), -1.0);      // row rr = column rr (symmetric!)
        for (int a = sub; a < N; a += MW_S_W) acc_fma<K, DK, KB>(s, ldx<DK>(q.B, q.Bp, c.coff + rr + a * q.xlen), ldx<KB>(dyl, N, a));
        mw<K> v = lanes_sum<K, MW_S_W>(acc_result<K>(s));
        if (live && sub == 0) stx<K>(w, P, r, v);
    }
    __syncthreads();
    wg_trmv_n<KC>(q.Si + c.Soff, q.Slen, P, P, w, P, w3, P, tid);      // t'_j = Si_j r_j
    for (int i = tid; i < P; i += MW_NT) {
#pragma unroll
        for (int l = 0; l < K; l++) q.t[(long)l * q.xlen + c.coff + i] = l < KC ? (double)w3[(long)l * P + i] : 0.0;
    }
    for (int a0 = 0; a0 < N; a0 += MW_NT / MW_S_W) {                     // u'_j = LinvB_j^T t'_j + B_j^T dx_j
        const int a = a0 + tid / MW_S_W;
        const bool live = a < N;
        const int aa = live ? a : 0;
        acc<K> s;
        acc_zero<K>(s);
        for (int r = sub; r < P; r += MW_S_W) {
            acc_fma<K, KC, KC>(s, ldx<KC>(q.LB, plane, c.coff + r + aa * q.xlen), ldx<KC>(w3, P, r));
            acc_fma<K, DK, KB>(s, ldx<DK>(q.B, q.Bp, c.coff + r + aa * q.xlen), ldx<KB>(w2, P, r));
        }
        mw<K> v = lanes_sum<K, MW_S_W>(acc_result<K>(s));
        if (live && sub == 0) stx<K>(q.ub, (long)q.J * N, (long)j * N + a, v);
    }
}
template <int K, int KC, int DK, int MODE>
__global__ __launch_bounds__(MW_NT) void k_mw_solve_bwd(const MwDev q, const double *__restrict__ dy_in, double *__restrict__ dx, const double *__restrict__ mid_rhs_y,
                                                        double *__restrict__ dy_out, const double *__restrict__ rhs_x) {
    if constexpr (KC == K && mw_kf_of(K) < K) {
        if (q.kf < K) { mw_solve_bwd_body<K, mw_kf_of(K), mw_kf_of(K), DK, MODE>(q, dy_in, dx, mid_rhs_y, dy_out, rhs_x); return; }
    }
    mw_solve_bwd_body<K, K, KC, DK, MODE>(q, dy_in, dx, mid_rhs_y, dy_out, rhs_x);
}

// The same solve for clusters or a Q of more than ~64 rows: with one workgroup per cluster (and one for Q) the six products of the stage run one
// after the other on one compute unit each (55 us per launch at P = 96, N = 97).  Here every product is a launch of its own whose ROWS are
// spread over workgroups, MW_SW_L lanes (a whole wave) per row:
//   1  t = Si rhs_x (per cluster)          2  v = rhs_y - LB^T t (one dot product over all rows of all clusters)     3  z = Qi v
//   4  dy = Qi^T z                         5  t <- t + LB dy                                                          6  dx = Si^T t (per cluster)
// vz: 2 N numbers of scratch (v, then z), planar with plane 2 N.  Unsharded contexts only (the sharded solve exchanges the partial u_j).
#ifndef MW_SW_L
#define MW_SW_L 64           // lanes per row.  Sixteen until the end of round 5: a row of 192 was twelve dependent K-limb multiply-adds per lane and four shuffle steps; with a
                             // whole wave per row it is three and six, on four times the workgroups (whole iterations at 5 limbs: Nsphere_packing N = 3 2.030 -> 1.930 ms
                             // with 64 lanes, 1.955 with 32; N = 2 1.124 -> 1.087; three-point as named 2.106 -> 2.074; SDPA x64 1.694 -> 1.654)
#endif
// KA <= K: limbs of the products (MwDev::kf); the difference rhs_y - [uadd] - LB^T t of stage 2 -- which cancels to the size of a residual in the refinement's
// second pass -- is accumulated in K limbs from KA-limb operands; what is stored carries K planes, the upper ones zero
template <int K, int KA>
__device__ __forceinline__ void mw_solve_wide_body(const MwDev &q, int stage, const double *__restrict__ rhs_x, const double *__restrict__ rhs_y,
                                                   double *__restrict__ dx, double *__restrict__ dy, double *__restrict__ vz) {
    using namespace mwk;
    constexpr int RPW = MW_NT / MW_SW_L;
    const int sub = threadIdx.x % MW_SW_L, row = blockIdx.x * RPW + threadIdx.x / MW_SW_L, N = q.N;
    const long lbp = q.xlen * (long)N, qp = (long)N * N, vp = 2L * N;
    acc<KA> s;
    acc_zero<KA>(s);
    if (stage == 1 || stage == 6) {
        const MwClu &c = q.clu[blockIdx.y];
        const int P = c.P;
        if (blockIdx.x * RPW >= P) return;                  // uniform over the workgroup
        const bool live = row < P;
        const int i = live ? row : 0;
        const double *Si = q.Si + c.Soff;
        if (stage == 1) {
            for (int cc = sub; cc <= i; cc += MW_SW_L) acc_fma<KA, KA, KA>(s, ldx<KA>(Si, q.Slen, i + (long)cc * P), ldx<KA>(rhs_x, q.xlen, c.coff + cc));
        } else {
            for (int cc = i + sub; cc < P; cc += MW_SW_L) acc_fma<KA, KA, KA>(s, ldx<KA>(Si, q.Slen, cc + (long)i * P), ldx<KA>(q.t, q.xlen, c.coff + cc));
        }
        const mw<KA> v = lanes_sum<KA, MW_SW_L>(acc_result<KA>(s));
        if (live && sub == 0) stx<K>(stage == 1 ? q.t : dx, q.xlen, c.coff + i, cvt<K, KA>(v));
        return;
    }
    if (stage == 5) {
        if ((long)blockIdx.x * RPW >= q.xlen) return;
        const bool live = row < q.xlen;
        const long g = live ? row : 0;
        if (sub == 0) acc_add<KA, KA>(s, ldx<KA>(q.t, q.xlen, g));
        for (int a = sub; a < N; a += MW_SW_L) acc_fma<KA, KA, KA>(s, ldx<KA>(q.LB, lbp, g + a * q.xlen), ldx<KA>(dy, N, a));
        const mw<KA> v = lanes_sum<KA, MW_SW_L>(acc_result<KA>(s));
        if (live && sub == 0) stx<K>(q.t, q.xlen, g, cvt<K, KA>(v));
        return;
    }
    if (blockIdx.x * RPW >= N) return;
    const bool live = row < N;
    const int a = live ? row : 0;
    if (stage == 2) {
        acc<K> sk;
        acc_zero<K>(sk);
        if (sub == 0) {
            acc_add<K, K>(sk, ldx<K>(rhs_y, N, a));
            if (q.uadd) acc_add<K, K>(sk, ldx<K>(q.uadd, N, a), -1.0);
        }
        for (long g = sub; g < q.xlen; g += MW_SW_L) acc_fma<K, KA, KA>(sk, ldx<KA>(q.LB, lbp, g + a * q.xlen), ldx<KA>(q.t, q.xlen, g), -1.0);
        const mw<K> v = lanes_sum<K, MW_SW_L>(acc_result<K>(sk));
        if (live && sub == 0) stx<K>(vz, vp, a, v);
        return;
    } else if (stage == 3) {
        for (int cc = sub; cc <= a; cc += MW_SW_L) acc_fma<KA, KA, KA>(s, ldx<KA>(q.Qi, qp, a + (long)cc * N), ldx<KA>(vz, vp, cc));
    } else {
        for (int cc = a + sub; cc < N; cc += MW_SW_L) acc_fma<KA, KA, KA>(s, ldx<KA>(q.Qi, qp, cc + (long)a * N), ldx<KA>(vz, vp, N + cc));
    }
    const mw<KA> v = lanes_sum<KA, MW_SW_L>(acc_result<KA>(s));
    if (live && sub == 0) {
        if (stage == 3) stx<K>(vz, vp, N + a, cvt<K, KA>(v));
        else stx<K>(dy, N, a, cvt<K, KA>(v));
    }
}
template <int K>
__global__ __launch_bounds__(MW_NT) void k_mw_solve_wide(const MwDev q, int stage, const double *__restrict__ rhs_x, const double *__restrict__ rhs_y,
                                                         double *__restrict__ dx, double *__restrict__ dy, double *__restrict__ vz) {
    if constexpr (mw_kf_of(K) < K) { if (q.kf < K) { mw_solve_wide_body<K, mw_kf_of(K)>(q, stage, rhs_x, rhs_y, dx, dy, vz); return; } }
    mw_solve_wide_body<K, K>(q, stage, rhs_x, rhs_y, dx, dy, vz);
}

// ---------------------------------------------------------------------------------------------------------------------
// One step of iterative refinement of the solve stage.  The reference solves with substitutions (approx_solve_tril! / solve_cho_precomp! /
// approx_solve_triu!, src/solver.jl:1538, 1557, 1567-1572), whose residuals stay at the working accuracy whatever cond(L_j), cond(L_Q) are;
// a product with an explicit inverse factor has a residual of cond(L) eps instead.  The residuals of the computed (dx, dy),
//     r_x = rhs_x - S dx + B dy,        r_y = rhs_y - B^T dx,
// are formed in K limbs from the matrices as ASSEMBLED (S0) and solved for with the same inverse-factor products: the sum has the backward
// error of the working precision (measured: scripts/refine_check.py; cohnelkies(8,15) y rows 2^-208 -> 2^-268 at 5 limbs).
// This kernel is the form with rows over many workgroups, MW_SW_L lanes per row, beside k_mw_solve_wide (clusters or Q beyond 64 rows):
//   stage 1   rx2 = rhs_x - S_j dx_j + B_j dy (grid.y = cluster)   and, in the workgroups with blockIdx.y = J,   u2 = B^T dx over this rank's rows
//   stage 3   dx += dx2, dy += dy2
// The small systems take the same step inside the launches of the solve itself (k_mw_solve_bwd, MODE 1 / 2).
// ---------------------------------------------------------------------------------------------------------------------
template <int K, int DK>
__global__ __launch_bounds__(MW_NT) void k_mw_refine(const MwDev q, int stage, const double *__restrict__ rhs_x, double *__restrict__ dx, double *__restrict__ dy) {
    using namespace mwk;
    constexpr int RPW = MW_NT / MW_SW_L;
    const int sub = threadIdx.x % MW_SW_L, row = blockIdx.x * RPW + threadIdx.x / MW_SW_L, N = q.N;
    if (stage == 3) {
        double mc[2] = {0.0, 0.0}, mv[2] = {0.0, 0.0};      // largest |correction| and |value| of dx, dy this thread has seen: MwDev::refstat
        for (long i = (long)blockIdx.x * MW_NT + threadIdx.x; i < q.xlen + N; i += (long)gridDim.x * MW_NT) {
            const bool isx = i < q.xlen;
            double *v = isx ? dx : dy;
            const double *cr = isx ? q.dx2 : q.dy2;
            const long pl = isx ? (long)q.xlen : (long)N, at = isx ? i : i - q.xlen;
            const mw<K> old = ldx<K>(v, pl, at), cor = ldx<K>(cr, pl, at);
            mc[isx ? 0 : 1] = fmax(mc[isx ? 0 : 1], __builtin_fabs(cor.l[0])); mv[isx ? 0 : 1] = fmax(mv[isx ? 0 : 1], __builtin_fabs(old.l[0]));
            stx<K>(v, pl, at, add<K>(old, cor));
        }
        if (q.refstat) {
#pragma unroll
            for (int w = 0; w < 2; w++) {
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { mc[w] = fmax(mc[w], __shfl_xor(mc[w], off, 64)); mv[w] = fmax(mv[w], __shfl_xor(mv[w], off, 64)); }
                if ((threadIdx.x & 63) == 0 && mv[w] > 0.0) {
                    atomicMax(q.refstat + 2 * w, (unsigned long long)__double_as_longlong(mc[w]));
                    atomicMax(q.refstat + 2 * w + 1, (unsigned long long)__double_as_longlong(mv[w]));
                }
            }
        }
        return;
    }
    acc<K> s;
    acc_zero<K>(s);
    if ((int)blockIdx.y == q.J) {                           // u2[a] = sum_g B[g, a] dx[g]
        if (blockIdx.x * RPW >= N) return;
        const bool live = row < N;
        const int a = live ? row : 0;
        for (long g = sub; g < q.xlen; g += MW_SW_L) acc_fma<K, DK, K>(s, ldx<DK>(q.B, q.Bp, g + a * q.xlen), ldx<K>(dx, q.xlen, g));
        const mw<K> v = lanes_sum<K, MW_SW_L>(acc_result<K>(s));
        if (live && sub == 0) stx<K>(q.u2, N, a, v);
        return;
    }
    const MwClu &c = q.clu[blockIdx.y];
    const int P = c.P;
    if (blockIdx.x * RPW >= P) return;                      // uniform over the workgroup
    const bool live = row < P;
    const int i = live ? row : 0;
    const double *S0 = q.S0 + c.Soff;
    if (sub == 0) acc_add<K, K>(s, ldx<K>(rhs_x, q.xlen, c.coff + i));
    for (int cc = sub; cc < P; cc += MW_SW_L) acc_fma<K, K, K>(s, ldx<K>(S0, q.Slen, cc + (long)i * P), ldx<K>(dx, q.xlen, c.coff + cc), -1.0);      // row i = column i (symmetric!)
    for (int a = sub; a < N; a += MW_SW_L) acc_fma<K, DK, K>(s, ldx<DK>(q.B, q.Bp, c.coff + i + a * q.xlen), ldx<K>(dy, N, a));
    const mw<K> v = lanes_sum<K, MW_SW_L>(acc_result<K>(s));
    if (live && sub == 0) stx<K>(q.rx2, q.xlen, c.coff + i, v);
}

// reciprocal diagonals of Cholesky factors passed in by the caller (clrs_mw_schur_assemble with host or foreign factors)
template <int K>
__global__ __launch_bounds__(MW_NT) void k_mw_xrd(const MwDev q, const double *__restrict__ Xc) {
    using namespace mwk;
    const MwBlk &k = q.blk[blockIdx.x];
    for (int i = threadIdx.x; i < k.n; i += MW_NT) stx<K>(q.xrd + k.rd_off, q.xrdlen, i, recip<K>(ldx<K>(Xc + k.xyoff, q.xylen, i + (long)i * k.n)));
    __threadfence_block();
    __syncthreads();
    wg_scaled_factors<K>(Xc + k.xyoff, q.xylen, k.n, q.xrd + k.rd_off, q.xrdlen, k.n, q.Xf + k.xyoff, q.xylen, k.n, q.Xb + k.xyoff, q.xylen, k.n, threadIdx.x);
}

#endif
