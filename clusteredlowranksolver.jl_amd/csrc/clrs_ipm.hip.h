// clrs_ipm.hip.h -- the interior-point iteration AROUND the hot path, device resident (gfx950).
//
// once Schur assembly / factor / solve run on the GPU, the per-block cubic work of the
// iteration (residual R, the matrices Z and dY of the search direction, the step length) and the constraint-wise
// traces <A_*, .> / weighted sums sum_i a_i A_i dominate and force the iterates through PCIe every iteration.  These
// kernels keep x, y, X, Y and every intermediate in HBM; the host only sequences launches and reads one small record per
// iteration.  One workgroup per PSD block, the block LDS resident (n <= 48 here; larger blocks keep the host loop).
//
//   k_ipm_pre        X Y product, tr(XY), chol(X); P = sum x_i A_i - X -+ C, max |P|   src/solver.jl:369, 961-970, 388-399, 882-893
//   ipm_weighted     sum_i a_i A_i per block (device function of k_ipm_pre / k_ipm_dXdY)   compute_weighted_A! :1410-1470
//   k_ipm_dense_dot  <A_e, M> for the dense constraint matrices    trace_A :1290-1366
//   k_ipm_csum       per-constraint sums of the per-term traces    trace_A :1368-1407, residual d :863-879, rhs_x :1518-1523;
//                    its first call of an iteration also forms p = +-b - B^T x   :899-916
//   k_ipm_Z          Z = sym(X^-1 (P Y - R)), w^T Z v per term     compute_search_direction! :1501-1514
//   k_ipm_dXdY       dX = P + sum dx_i A_i, dY = sym(X^-1 (R - dX Y))   :1585-1613
//   k_ipm_step       min eig of L^-1 dM L^-T per block (Householder tridiagonalisation + Sturm multisection)  :1620-1693
//   k_ipm_update     x, X += alpha_d (dx, dX); y, Y += alpha_p (dy, dY); objective dots   :485-495, 793-804
//   k_ipm_scalars    the scalar control flow of the loop (mu, beta, step lengths, errors)  :369-374, 429-447, 470-483
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "clrs_wave.hip.h"

namespace clrs {

struct IBlock {
    int n, kind;          // side; 0 low rank, 1 dense
    int URt, ULt;         // expanded right / left unique vectors
    int T, cnt;           // terms (low rank) / matrices (dense)
    int xoff;             // offset of the block's cluster in the x layout
    int wmfma;            // low rank, bit 0: sum_i a_i A_i as one MFMA contraction over the terms; bit 1: Z V by MFMA in k_ipm_Z (their LDS operands fit: host plan)
    long long xyoff;      // offset in the X/Y layout
    long long vr_off, wl_off;   // static arena: V (n x URt), W (n x ULt), ld n (equal when the tables coincide)
    long long t0;         // first term of the block (original term order)
    long long a_off;      // dense: offset of the stack of matrices in the static arena
    long long d0;         // dense: first dense entry of the block
};

// scalars living in device memory (index into IpmBuf::scal)
enum {
    SC_MU = 0, SC_MU_P, SC_MU_C, SC_BETA_C, SC_ALPHA_P, SC_ALPHA_D, SC_DOBJ, SC_POBJ, SC_GAP, SC_DUAL_ERR, SC_PRIMAL_ERR,
    SC_PD_FEAS, SC_XY, SC_MAXP, SC_MAXp, SC_MAXd, SC_EIG_X, SC_EIG_Y, SC_ERRCODE, SC_K, SC_ITER, SC_PD_PREV,
    SC_INFO0 = 30, SC_INFO1 = 31,      // the library's two status words (factorisation, Cholesky of X), copied here for the host's single read
    SC_COUNT = 32
};

struct IpmParams {
    double beta_infeasible, beta_feasible, gamma, dual_error_threshold, primal_error_threshold, max_complementary_gap;
    double sgn, constant;   // +1 maximise / -1 minimise; objective constant
    double step_length_threshold;
    int safe_step, K;       // K = sum of the block sides
};

struct IpmBuf {
    // iterates and directions
    double *x, *y, *X, *Y, *dx, *dy, *dX, *dY;
    double *Xchol, *XY, *P, *Z;
    double *d, *p, *rhs_x;
    double *AZ;            // per term: lambda-free pairing w^T Z v
    double *AY;            // per term: w^T Y v (from the assembly)
    double *dtr;           // per dense entry: <A_e, M>
    double *part;          // per block partial sums: [NB][8]
    double *eig;           // per block: min eig for X part, Y part: [NB][2]
    double *scal;          // SC_COUNT doubles
    int *info;             // [2] factorisation status words of the context
    // static problem data
    const double *stat, *C, *c, *b, *B;
    const int *ayL, *ayR, *term_p;      // per original term
    const double *term_lam;
    const int *crow_ptr, *crow_term;    // CSR: constraint (global x index) -> terms (original indices)
    const int *drow_ptr, *drow_ent;     // CSR: constraint -> dense entries
    const int *dense_p;                 // per dense entry: cluster-local constraint
    const IBlock *blocks;
    int NB, N, xlen;
    long long xylen, T, D;
};

#ifdef CLRS_IPM_STAMPS
#define IPM_STAMP(i)                                                                                            \
    do {                                                                                                        \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                                              \
            unsigned long long t_;                                                                              \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                          \
            q.eig[2 * q.NB + (i)] = (double)t_;                                                                 \
        }                                                                                                       \
    } while (0)
#else
#define IPM_STAMP(i) do {} while (0)
#endif

__device__ __forceinline__ double block_reduce_sum(double v, double *red) {   // 256 threads, deterministic tree
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ double block_reduce_max(double v, double *red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// load an n x n block (ld n in memory) into LDS (ld lda), zero padded to n16 x n16.  Eight 16 x 16 tiles per batch, every load of a
// batch issued (from a clamped address, the select after the load) before any of them is stored: a block of the named configurations
// is one trip to memory instead of one per tile (a select AROUND a load is a branch, and each tile then waits for its own trip).
template <bool LOWER>
__device__ __forceinline__ void ipm_load_tiles(double *A, int lda, const double *G, int n, int n16, int tid) {
    const int i16 = tid & 15, j16 = tid >> 4, nt = n16 >> 4;
    for (int tj0 = 0; tj0 < nt; tj0 += 2)
        for (int ti0 = 0; ti0 < nt; ti0 += 4) {
            double v[8];
            bool in[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int ti = ti0 + (e & 3), tj = tj0 + (e >> 2);
                const int i = ti * 16 + i16, j = tj * 16 + j16;
                in[e] = ti < nt && tj < nt && i < n && j < n && (!LOWER || i >= j);
                v[e] = G[in[e] ? i + (long long)j * n : 0];
            }
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int ti = ti0 + (e & 3), tj = tj0 + (e >> 2);
                if (ti < nt && tj < nt) A[(ti * 16 + i16) + (tj * 16 + j16) * lda] = in[e] ? v[e] : 0.0;
            }
        }
}
__device__ __forceinline__ void ipm_load(double *A, int lda, const double *G, int n, int n16, int tid) { ipm_load_tiles<false>(A, lda, G, n, n16, tid); }
__device__ __forceinline__ void ipm_load_chol(double *A, int lda, double *dinv, const double *G, int n, int n16, int tid) {
    const double dg = G[(tid < n) ? tid + (long long)tid * n : 0];          // in flight with the tiles
    ipm_load_tiles<true>(A, lda, G, n, n16, tid);
    if (tid < n16) dinv[tid] = (tid < n) ? 1.0 / dg : 0.0;
}
// M <- (L L^T)^-1 M for an n x n matrix in LDS (columns as right-hand sides); all 256 threads
__device__ __forceinline__ void ipm_potrs(const double *L, int lda, const double *dinv, double *M, int n, int wave, int lane) {
    lds_trsm<false>(L, lda, dinv, M, 1, lda, n, n, wave, 4, lane);
    __syncthreads();
    lds_trsm<true>(L, lda, dinv, M, 1, lda, n, n, wave, 4, lane);
    __syncthreads();
}

__device__ __forceinline__ void ipm_weighted(const IpmBuf &q, const IBlock &k, const double *a, double *M, int lda, double *work, int tid);

// ---- k_ipm_pre: XY = X Y, partial tr(XY), Xchol = chol(X); then P = sum x_i A_i - X -+ C, max |P| (everything that depends on the
// iterate alone: one launch; src/solver.jl:388-399, 961-983) -------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ipm_pre(const IpmBuf q, const IpmParams prm) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double red[4];
    const IBlock k = q.blocks[blockIdx.x];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = k.n, n16 = (n + 15) & ~15, lda = n16 + 2;
    double *Xs = lds, *Ys = Xs + lda * n16, *Ps = Ys + lda * n16, *dinv = Ps + lda * n16;
    ipm_load(Xs, lda, q.X + k.xyoff, n, n16, tid);
    ipm_load(Ys, lda, q.Y + k.xyoff, n, n16, tid);
    __syncthreads();
    lds_gemm_tn(Xs, lda, Ys, lda, Ps, lda, n, n, n, wave, 4, lane);       // X symmetric: X^T Y = X Y
    __syncthreads();
    double tr = 0.0;
    const int i16 = tid & 15, j16 = tid >> 4;
    for (int j0 = 0; j0 < n; j0 += 16)
        for (int i0 = 0; i0 < n; i0 += 16) {
            const int i = i0 + i16, j = j0 + j16;
            if (i < n && j < n) {
                q.XY[k.xyoff + i + (long long)j * n] = Ps[i + j * lda];
                if (i == j) tr += Ps[i + j * lda];
            }
        }
    tr = block_reduce_sum(tr, red);
    if (tid == 0) q.part[blockIdx.x * 8 + 0] = tr;
    // Cholesky of X in place (identity padding)
    for (int j0 = 0; j0 < n16; j0 += 16)
        for (int i0 = 0; i0 < n16; i0 += 16) {
            const int i = i0 + i16, j = j0 + j16;
            if (i >= n || j >= n) Xs[i + j * lda] = (i == j) ? 1.0 : 0.0;
        }
    __syncthreads();
    const bool bad = lds_potrf(Xs, lda, dinv, n, wave, 4, lane);
    if (bad && lane == 0) atomicMin(q.info + 1, (int)blockIdx.x + 1);
    __syncthreads();
    for (int j0 = 0; j0 < n; j0 += 16)
        for (int i0 = 0; i0 < n; i0 += 16) {
            const int i = i0 + i16, j = j0 + j16;
            if (i < n && j < n) q.Xchol[k.xyoff + i + (long long)j * n] = (i >= j) ? Xs[i + j * lda] : 0.0;
        }
    __syncthreads();
    // ---- P (the LDS is free again) ----
    double *M = lds, *work = M + n * n;
    ipm_weighted(q, k, q.x, M, n, work, tid);
    double mx = 0.0;
    for (int e = tid; e < n * n; e += 256) {
        const double v = M[e] - q.X[k.xyoff + e] - prm.sgn * q.C[k.xyoff + e];
        q.P[k.xyoff + e] = v;
        mx = fmax(mx, fabs(v));
    }
    mx = block_reduce_max(mx, red);
    if (tid == 0) q.part[blockIdx.x * 8 + 1] = mx;
}

// ---- sum_i a_i A_i of one block into LDS matrix M (n x n, ld lda), a = coefficient vector in the x layout ------------------
// low rank: M[i,j] = sum_t a_{p(t)} lam_t W[i, ayL_t] V[j, ayR_t] over all terms of the block (both (r,s) and (s,r) are
// terms, so the result is the full symmetric matrix); dense: M = sum_e a_{p(e)} A_e.
__device__ __forceinline__ void ipm_weighted(const IpmBuf &q, const IBlock &k, const double *a, double *M, int lda, double *work, int tid) {
    const int n = k.n;
    const int i16 = tid & 15, j16 = tid >> 4;
    if (k.kind == 0 && (k.wmfma & 1)) {
        // M = (W diag(coef))_gathered V_gathered^T as one contraction over the terms: A[t, i] = coef_t W[i, ayL_t], B[t, j] = V[j, ayR_t]
        // (K x M operands of lds_gemm_tn, zero padded to 4 terms / 16 columns), 16 x 16 x 4 MFMA tiles instead of T FMAs per entry
        const int T = k.T, T4 = (T + 3) & ~3, ldt = ((T + 15) & ~15) + 2, n16 = (n + 15) & ~15;
        double *At = work, *Bt = At + ldt * n16, *coef = Bt + ldt * n16;
        int *tl = (int *)(coef + T4), *tr = tl + T4;
        for (int t = tid; t < T4; t += 256) {
            const bool in = t < T;
            const long long g = k.t0 + (in ? t : 0);
            const double cf = a[k.xoff + q.term_p[g]] * q.term_lam[g];
            const int l = q.ayL[g], r = q.ayR[g];
            coef[t] = in ? cf : 0.0;
            tl[t] = l;
            tr[t] = r;
        }
        __syncthreads();
        const double *Vg = q.stat + k.vr_off, *Wg = q.stat + k.wl_off;
        for (int e0 = 0; e0 < T4 * n16; e0 += 4 * 256) {
            double wv[4], vv[4];
            bool in[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {          // i fastest: coalesced columns of W / V; all loads of a batch before the first store
                const int e = e0 + u * 256 + tid, i = e % n16, t = min(e / n16, T4 - 1);
                in[u] = e < T4 * n16 && i < n;
                wv[u] = Wg[(in[u] ? i : 0) + (long long)tl[t] * n];
                vv[u] = Vg[(in[u] ? i : 0) + (long long)tr[t] * n];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int e = e0 + u * 256 + tid, i = e % n16, t = e / n16;
                if (e < T4 * n16) {
                    At[t + i * ldt] = in[u] ? coef[t] * wv[u] : 0.0;
                    Bt[t + i * ldt] = in[u] ? vv[u] : 0.0;
                }
            }
        }
        __syncthreads();
        lds_gemm_tn(At, ldt, Bt, ldt, M, lda, n, n, T, tid >> 6, 4, tid & 63);
        __syncthreads();
        return;
    }
    if (k.kind == 0) {
        // stage V, W and the per-term coefficients
        const double *Vg = q.stat + k.vr_off, *Wg = q.stat + k.wl_off;
        double *Vs = work, *Ws = (k.wl_off == k.vr_off) ? Vs : Vs + n * k.URt;
        double *coef = Vs + n * k.URt + ((k.wl_off == k.vr_off) ? 0 : n * k.ULt);
        int *tl = (int *)(coef + k.T), *tr = tl + k.T;
        for (int e = tid; e < n * k.URt; e += 256) Vs[e] = Vg[e];
        if (Ws != Vs)
            for (int e = tid; e < n * k.ULt; e += 256) Ws[e] = Wg[e];
        for (int t = tid; t < k.T; t += 256) {
            coef[t] = a[k.xoff + q.term_p[k.t0 + t]] * q.term_lam[k.t0 + t];
            tl[t] = q.ayL[k.t0 + t];
            tr[t] = q.ayR[k.t0 + t];
        }
        __syncthreads();
        for (int j0 = 0; j0 < n; j0 += 16)
            for (int i0 = 0; i0 < n; i0 += 16) {
                const int i = i0 + i16, j = j0 + j16;
                if (i < n && j < n) {
                    double s = 0.0;
                    for (int t = 0; t < k.T; t++) s += coef[t] * (Ws[i + tl[t] * n] * Vs[j + tr[t] * n]);
                    M[i + j * lda] = s;
                }
            }
    } else {
        const double *Ag = q.stat + k.a_off;
        for (int j0 = 0; j0 < n; j0 += 16)
            for (int i0 = 0; i0 < n; i0 += 16) {
                const int i = i0 + i16, j = j0 + j16;
                if (i < n && j < n) {
                    double s = 0.0;
                    for (int e = 0; e < k.cnt; e++) s += a[k.xoff + q.dense_p[k.d0 + e]] * Ag[(long long)e * n * n + i + j * n];
                    M[i + j * lda] = s;
                }
            }
    }
    __syncthreads();
}

// ---- k_ipm_dense_dot: dtr[e] = <A_e, M> for every dense entry (one wave per entry) --------------------------------------------
__global__ __launch_bounds__(256) void k_ipm_dense_dot(const IpmBuf q, const double *__restrict__ Mxy, const int *__restrict__ ent_block) {
    const long long e = blockIdx.x * 4ll + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (e >= q.D) return;
    const IBlock k = q.blocks[ent_block[e]];
    const int nn = k.n * k.n;
    const double *A = q.stat + k.a_off + (e - k.d0) * (long long)nn, *M = Mxy + k.xyoff;
    double s = 0.0;
    for (int i = lane; i < nn; i += 64) s += A[i] * M[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) q.dtr[e] = s;
}

// ---- k_ipm_csum: out[i] = sign * (base[i] - sum_t lam_t val[t] - sum_e dtr[e] - with_By * (B y)_i), max |out| -----------------
// one thread per constraint (global x index); fixed summation order
// Workgroups from p_from on (first call of an iteration only; -1: none) compute p = sgn b - B^T x instead, one wave per free variable
// (src/solver.jl:961-983): two independent residuals, one launch.
__global__ __launch_bounds__(256) void k_ipm_csum(const IpmBuf q, const double *__restrict__ base, double base_sign, const double *__restrict__ val,
                                                  int with_By, double *__restrict__ out, int part_slot, int p_from, double sgn) {
    __shared__ double red[4];
    if (p_from >= 0 && (int)blockIdx.x >= p_from) {
        const int kk = ((int)blockIdx.x - p_from) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (kk >= q.N) return;
        const double *col = q.B + (long long)kk * q.xlen;
        double s = 0.0;
        for (int i = lane; i < q.xlen; i += 64) s += col[i] * q.x[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) q.p[kk] = sgn * q.b[kk] - s;
        return;
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    double r = 0.0;
    if (i < q.xlen) {
        double s = base_sign * base[i];
        for (int u = q.crow_ptr[i]; u < q.crow_ptr[i + 1]; u++) {
            const int t = q.crow_term[u];
            s -= q.term_lam[t] * val[t];
        }
        for (int u = q.drow_ptr[i]; u < q.drow_ptr[i + 1]; u++) s -= q.dtr[q.drow_ent[u]];
        if (with_By)
            for (int kk = 0; kk < q.N; kk++) s -= q.B[i + (long long)kk * q.xlen] * q.y[kk];
        out[i] = s;
        r = fabs(s);
    }
    r = block_reduce_max(r, red);
    if (threadIdx.x == 0 && part_slot >= 0) q.part[(long long)(q.NB + blockIdx.x) * 8 + part_slot] = r;
}

// ---- k_ipm_scalars: the scalar control flow, one thread -----------------------------------------------------------------------
// stage 0: mu from tr(XY) partials; mu_p                                      (src/solver.jl:369-374)
// stage 1: errors after the residuals; pd_feas                               (:441-447, computed before the predictor here)
// stage 2: beta_c, mu_c after the predictor                                   (:429-434)
// stage 3: step lengths from the eigenvalues                                  (:462-483, :1684-1692)
// stage 4: objectives and gap after the update (grid partials of k_ipm_update) (:793-804, 844-847)
__device__ __forceinline__ void ipm_scalar_stage(const IpmBuf &q, const IpmParams &prm, int stage, int ngrid, int ncsum_or_row0) {
    const int ncsum = ncsum_or_row0, row0 = ncsum_or_row0;
    double *s = q.scal;
    if (stage == 0) {
        double xy = 0.0;
        for (int b = 0; b < q.NB; b++) xy += q.part[b * 8 + 0];
        s[SC_XY] = xy;
        s[SC_MU] = xy / prm.K;
        s[SC_MU_P] = (s[SC_PD_FEAS] != 0.0) ? 0.0 : prm.beta_infeasible * s[SC_MU];
        s[SC_PD_PREV] = s[SC_PD_FEAS];      // the feasibility the previous iteration found: beta_c is chosen with it (src/solver.jl:429-434 precede :441-447)
        if (s[SC_MU] > prm.max_complementary_gap) s[SC_ERRCODE] = 3.0;
    } else if (stage == 1) {
        double mP = 0.0, md = 0.0, mp = 0.0;
        for (int b = 0; b < q.NB; b++) mP = fmax(mP, q.part[b * 8 + 1]);
        for (int g = 0; g < ncsum; g++) md = fmax(md, q.part[(long long)(q.NB + g) * 8 + 0]);
        for (int kk = 0; kk < q.N; kk++) mp = fmax(mp, fabs(q.p[kk]));
        s[SC_MAXP] = mP; s[SC_MAXd] = md; s[SC_MAXp] = mp;
        s[SC_DUAL_ERR] = fmax(mp, mP);
        s[SC_PRIMAL_ERR] = md;
    } else if (stage == 2) {
        double a = 0.0, bb = 0.0, c = 0.0;
        for (int b = 0; b < q.NB; b++) { a += q.part[b * 8 + 2]; bb += q.part[b * 8 + 3]; c += q.part[b * 8 + 4]; }
        const double r = (s[SC_XY] + a + bb + c) / (s[SC_MU] * prm.K);
        const double beta = (r < 1.0) ? r * r : r;
        // beta_c with the feasibility of the PREVIOUS iteration (kept by stage 0: this stage may run once per workgroup and all of
        // them must read the same value), only then the new feasibility -- the order of the reference
        const bool was_feas = s[SC_PD_PREV] != 0.0;
        s[SC_BETA_C] = was_feas ? fmin(fmax(prm.beta_feasible, beta), 1.0) : fmax(prm.beta_infeasible, beta);
        const bool feas = s[SC_DUAL_ERR] < prm.dual_error_threshold && s[SC_PRIMAL_ERR] < prm.primal_error_threshold;
        s[SC_PD_FEAS] = feas ? 1.0 : 0.0;
        s[SC_MU_C] = s[SC_BETA_C] * s[SC_MU];
    } else if (stage == 3) {
        double ex = 1e300, ey = 1e300;
        for (int b = 0; b < q.NB; b++) {
            const double fx = (q.blocks[b].n == 1) ? 0.0 : 1e-5;    // the reference subtracts 1e-5 from the Lanczos estimate (:1680)
            ex = fmin(ex, q.eig[b * 2 + 0] - fx);
            ey = fmin(ey, q.eig[b * 2 + 1] - fx);
        }
        s[SC_EIG_X] = ex; s[SC_EIG_Y] = ey;
        const bool unsafe = (s[SC_PD_FEAS] != 0.0) && !prm.safe_step;
        double ad = (ex > -prm.gamma && !unsafe) ? 1.0 : -prm.gamma / ex;
        double ap = (ey > -prm.gamma && !unsafe) ? 1.0 : -prm.gamma / ey;
        if (s[SC_PD_FEAS] != 0.0 && prm.safe_step) ad = ap = fmin(ad, ap);
        s[SC_ALPHA_D] = ad; s[SC_ALPHA_P] = ap;
        if (fmin(ad, ap) < prm.step_length_threshold || !(ad == ad) || !(ap == ap)) s[SC_ERRCODE] = 4.0;     // :470-475
        if (q.info[0] != 0x7f7f7f7f || q.info[1] != 0x7f7f7f7f) s[SC_ERRCODE] = 1.0;                          // a Cholesky failed: SolverFailure
    } else if (stage == 4) {
        double cy = 0.0, cx = 0.0, by = 0.0, xy = 0.0;
        for (int g = 0; g < ngrid; g++) {
            const double *pt = q.part + (long long)(row0 + g) * 8;
            cy += pt[5]; cx += pt[6]; by += pt[7]; xy += pt[0];
        }
        s[SC_DOBJ] = prm.sgn * cx + prm.constant;
        s[SC_POBJ] = cy + by + prm.constant;
        s[SC_GAP] = fabs(s[SC_DOBJ] - s[SC_POBJ]) / fmax(1.0, fabs(s[SC_DOBJ] + s[SC_POBJ]));
        if (s[SC_ERRCODE] == 0.0) s[SC_ITER] += 1.0;
    }
}

__global__ void k_ipm_scalars(const IpmBuf q, const IpmParams prm, int stage, int ngrid, int ncsum_or_row0) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    ipm_scalar_stage(q, prm, stage, ngrid, ncsum_or_row0);
}

// ---- k_ipm_Z: Z = sym(X^-1 (P Y - R)), R = mu' I - XY [- dX dY]; per-term pairings w^T Z v ------------------------------------
// inl != 0 (few blocks): the scalar control flow that precedes this kernel (mu, errors / beta_c) is evaluated here by thread 0 of
// every workgroup instead of in one-thread kernels of their own; all workgroups compute identical values.
__global__ __launch_bounds__(256) void k_ipm_Z(const IpmBuf q, const IpmParams prm, int corrector, int inl, int ncsum) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double sh_mu;
    const IBlock k = q.blocks[blockIdx.x];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = k.n, n16 = (n + 15) & ~15, lda = n16 + 2;
    double *Ls = lds, *As = Ls + lda * n16, *Bs = As + lda * n16, *Ts = Bs + lda * n16, *dinv = Ts + lda * n16, *work = dinv + n16;
    if (tid == 0) {
        if (inl) {
            if (!corrector) { ipm_scalar_stage(q, prm, 0, 0, 0); ipm_scalar_stage(q, prm, 1, 0, ncsum); }
            else ipm_scalar_stage(q, prm, 2, 0, 0);
        }
        sh_mu = q.scal[corrector ? SC_MU_C : SC_MU_P];
    }
    __syncthreads();
    const double mu = sh_mu;
    const int i16 = tid & 15, j16 = tid >> 4;
    ipm_load_chol(Ls, lda, dinv, q.Xchol + k.xyoff, n, n16, tid);
    ipm_load(As, lda, q.P + k.xyoff, n, n16, tid);
    ipm_load(Bs, lda, q.Y + k.xyoff, n, n16, tid);
    __syncthreads();
    lds_gemm_tn(As, lda, Bs, lda, Ts, lda, n, n, n, wave, 4, lane);       // P symmetric: P^T Y = P Y
    __syncthreads();
    if (corrector) {   // dX dY, both symmetric
        ipm_load(As, lda, q.dX + k.xyoff, n, n16, tid);
        ipm_load(Bs, lda, q.dY + k.xyoff, n, n16, tid);
        __syncthreads();
        // accumulate into a second product buffer: reuse work as n x n scratch through As after the product
        double *Ds = work;     // lda * n16 doubles of scratch (sized by the host)
        lds_gemm_tn(As, lda, Bs, lda, Ds, lda, n, n, n, wave, 4, lane);
        __syncthreads();
        for (int j0 = 0; j0 < n; j0 += 16)
            for (int i0 = 0; i0 < n; i0 += 16) {
                const int i = i0 + i16, j = j0 + j16;
                if (i < n && j < n) Ts[i + j * lda] += Ds[i + j * lda];
            }
        __syncthreads();
    }
    // T = P Y - (mu I - XY - dX dY) = P Y + XY [+ dX dY] - mu I
    for (int j0 = 0; j0 < n16; j0 += 16)
        for (int i0 = 0; i0 < n16; i0 += 16) {
            const int i = i0 + i16, j = j0 + j16;
            double v = 0.0;
            if (i < n && j < n) v = Ts[i + j * lda] + q.XY[k.xyoff + i + (long long)j * n] - ((i == j) ? mu : 0.0);
            Ts[i + j * lda] = v;
        }
    __syncthreads();
    ipm_potrs(Ls, lda, dinv, Ts, n, wave, lane);
    // symmetrise into As (= Z), store
    for (int j0 = 0; j0 < n16; j0 += 16)
        for (int i0 = 0; i0 < n16; i0 += 16) {
            const int i = i0 + i16, j = j0 + j16;
            const double v = (i < n && j < n) ? 0.5 * (Ts[i + j * lda] + Ts[j + i * lda]) : 0.0;
            As[i + j * lda] = v;
            if (i < n && j < n) q.Z[k.xyoff + i + (long long)j * n] = v;
        }
    __syncthreads();
    if (k.kind == 0) {
        // pairings: AZ[t] = W[:, ayL_t]^T Z V[:, ayR_t]   (TZ = Z V column by column, then dots)
        const double *Vg = q.stat + k.vr_off, *Wg = q.stat + k.wl_off;
        if (k.wmfma & 2) {
            // TZ = Z V as one MFMA product (Z symmetric and zero padded in As; V staged zero padded), then one dot product per term
            const int UR16 = (k.URt + 15) & ~15;
            double *Vs = work, *TZ = Vs + lda * UR16;
            for (int e0 = 0; e0 < lda * UR16; e0 += 4 * 256) {
                double vv4[4];
                bool in[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int e = e0 + u * 256 + tid, i = e % lda, j = e / lda;
                    in[u] = e < lda * UR16 && i < n && j < k.URt;
                    vv4[u] = Vg[in[u] ? i + (long long)j * n : 0];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int e = e0 + u * 256 + tid;
                    if (e < lda * UR16) Vs[e] = in[u] ? vv4[u] : 0.0;
                }
            }
            __syncthreads();
            lds_gemm_tn(As, lda, Vs, lda, TZ, lda, n, k.URt, n, wave, 4, lane);
            __syncthreads();
            for (int t = tid; t < k.T; t += 256) {
                const double *w = Wg + (long long)q.ayL[k.t0 + t] * n, *tz = TZ + q.ayR[k.t0 + t] * lda;
                double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
                int i = 0;
                for (; i + 4 <= n; i += 4) { s0 += w[i] * tz[i]; s1 += w[i + 1] * tz[i + 1]; s2 += w[i + 2] * tz[i + 2]; s3 += w[i + 3] * tz[i + 3]; }
                for (; i < n; i++) s0 += w[i] * tz[i];
                q.AZ[k.t0 + t] = (s0 + s1) + (s2 + s3);
            }
            return;
        }
        double *TZ = work;                       // n x URt
        for (int e = tid; e < n * k.URt; e += 256) {
            const int i = e % n, u = e / n;
            double s = 0.0;
            for (int kk = 0; kk < n; kk++) s += As[i + kk * lda] * Vg[kk + (long long)u * n];
            TZ[e] = s;
        }
        __syncthreads();
        for (int t = tid; t < k.T; t += 256) {
            const double *w = Wg + (long long)q.ayL[k.t0 + t] * n, *tz = TZ + q.ayR[k.t0 + t] * n;
            double s = 0.0;
            for (int i = 0; i < n; i++) s += w[i] * tz[i];
            q.AZ[k.t0 + t] = s;
        }
    }
}

// ---- k_ipm_dXdY: dX = P + sum dx_i A_i; dY = sym(X^-1 (R - dX Y)); partial dots <X,dY>, <dX,Y>, <dX,dY> -----------------------
__global__ __launch_bounds__(256) void k_ipm_dXdY(const IpmBuf q, int corrector) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double red[4];
    const IBlock k = q.blocks[blockIdx.x];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = k.n, n16 = (n + 15) & ~15, lda = n16 + 2;
    double *Ls = lds, *As = Ls + lda * n16, *Bs = As + lda * n16, *Ts = Bs + lda * n16, *dinv = Ts + lda * n16, *M = dinv + n16, *work = M + n * n;
    const double mu = q.scal[corrector ? SC_MU_C : SC_MU_P];
    const int i16 = tid & 15, j16 = tid >> 4;
    ipm_weighted(q, k, q.dx, M, n, work, tid);
    // dX (old dX / dY are still needed for the corrector's R: read them before overwriting)
    double *Ds = work;   // product dXold dYold
    if (corrector) {
        ipm_load(As, lda, q.dX + k.xyoff, n, n16, tid);
        ipm_load(Bs, lda, q.dY + k.xyoff, n, n16, tid);
        __syncthreads();
        lds_gemm_tn(As, lda, Bs, lda, Ds, lda, n, n, n, wave, 4, lane);
        __syncthreads();
    }
    for (int j0 = 0; j0 < n16; j0 += 16)
        for (int i0 = 0; i0 < n16; i0 += 16) {
            const int i = i0 + i16, j = j0 + j16;
            double v = 0.0;
            if (i < n && j < n) {
                v = q.P[k.xyoff + i + (long long)j * n] + M[i + j * n];
                q.dX[k.xyoff + i + (long long)j * n] = v;
            }
            As[i + j * lda] = v;
        }
    ipm_load_chol(Ls, lda, dinv, q.Xchol + k.xyoff, n, n16, tid);
    ipm_load(Bs, lda, q.Y + k.xyoff, n, n16, tid);
    __syncthreads();
    lds_gemm_tn(As, lda, Bs, lda, Ts, lda, n, n, n, wave, 4, lane);       // dX Y
    __syncthreads();
    // T = R - dX Y = mu I - XY [- dXold dYold] - dX Y
    for (int j0 = 0; j0 < n16; j0 += 16)
        for (int i0 = 0; i0 < n16; i0 += 16) {
            const int i = i0 + i16, j = j0 + j16;
            double v = 0.0;
            if (i < n && j < n) {
                v = ((i == j) ? mu : 0.0) - q.XY[k.xyoff + i + (long long)j * n] - Ts[i + j * lda];
                if (corrector) v -= Ds[i + j * lda];
            }
            Ts[i + j * lda] = v;
        }
    __syncthreads();
    ipm_potrs(Ls, lda, dinv, Ts, n, wave, lane);
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int j0 = 0; j0 < n; j0 += 16)
        for (int i0 = 0; i0 < n; i0 += 16) {
            const int i = i0 + i16, j = j0 + j16;
            if (i < n && j < n) {
                const double dy = 0.5 * (Ts[i + j * lda] + Ts[j + i * lda]);
                const long long g = k.xyoff + i + (long long)j * n;
                q.dY[g] = dy;
                const double dxv = As[i + j * lda];
                s1 += q.X[g] * dy;
                s2 += dxv * Bs[i + j * lda];
                s3 += dxv * dy;
            }
        }
    s1 = block_reduce_sum(s1, red);
    s2 = block_reduce_sum(s2, red);
    s3 = block_reduce_sum(s3, red);
    if (tid == 0) {
        q.part[blockIdx.x * 8 + 2] = s1;
        q.part[blockIdx.x * 8 + 3] = s2;
        q.part[blockIdx.x * 8 + 4] = s3;
    }
}

// Householder tridiagonalisation of a symmetric n x n matrix (n <= NC <= 32) inside ONE wave with the matrix in REGISTERS: lane r
// holds row r (NC doubles, static indices: the loop over the columns is fully unrolled), the entries of v and w reach the other lanes
// through v_readlane (SGPR operands of the FMAs), the norms and dot products through DPP butterflies.  Nothing goes through LDS in
// the n - 2 dependent steps (the LDS form below, kept for n > 32, spends 2.4k cycles per step on its strided row reads: 46k of the
// 97k cycles of k_ipm_step for n = 21).  d -> dd[0..n), e -> ee[0..n-1).
typedef double v2d_ipm __attribute__((ext_vector_type(2)));
template <int NC>
__device__ __forceinline__ void ipm_householder_regs(const double *Ws, int lda, int n, double *dd, double *ee, int lane) {
    double a[NC];
    const int r = lane < NC ? lane : 0;
#pragma unroll
    for (int j = 0; j < NC; j += 2) {                      // row r = column r (both triangles are kept): contiguous, 16-byte aligned
        const v2d_ipm t = *(const v2d_ipm *)(Ws + r * lda + j);
        a[j] = t[0];
        a[j + 1] = t[1];
    }
    const bool rowok = lane < n;
#pragma unroll
    for (int j = 0; j < NC; j++) a[j] = (rowok && j < n) ? a[j] : 0.0;
#pragma unroll
    for (int c = 0; c < NC - 2; c++) {
        if (c < n - 2) {                                    // uniform
            const double xi = (lane > c) ? a[c] : 0.0;      // column c below the diagonal (rows >= n hold zeros)
            const double ss = wave_sum(xi * xi);
            const double x0 = readlane_f64(xi, c + 1);
            if (ss == 0.0) {
                if (lane == 0) ee[c] = 0.0;
            } else {
                // |x| = ss * rsqrt(ss) with two Newton steps, tau = 2 / v^T v by reciprocal + Newton: the IEEE sqrt and division are ~60
                // dependent instructions per step.  H = I - tau v v^T is orthogonal for ANY alpha as long as tau = 2 / v^T v for the v in
                // use; an alpha that is |x| only to an ulp leaves an entry of that size where a zero is assumed, like rounding does.
                double rs = __builtin_amdgcn_rsq(ss);
                rs = rs * __builtin_fma(-0.5 * ss * rs, rs, 1.5);
                rs = rs * __builtin_fma(-0.5 * ss * rs, rs, 1.5);
                const double nrm = ss * rs;
                const double alpha = (x0 > 0.0) ? -nrm : nrm;
                const double v0 = x0 - alpha;
                const double vi = (lane == c + 1) ? v0 : xi;
                const double vtv = ss - x0 * x0 + v0 * v0;
                double rt = __builtin_amdgcn_rcp(vtv);
                rt = __builtin_fma(__builtin_fma(-vtv, rt, 1.0), rt, rt);
                rt = __builtin_fma(__builtin_fma(-vtv, rt, 1.0), rt, rt);
                const double tau = 2.0 * rt;
                double pa = 0.0, pb = 0.0, pc2 = 0.0, pd = 0.0;          // four partial sums: the FMAs of one chain wait on each other
#pragma unroll
                for (int j = c + 1; j < NC; j++) {
                    const double vj = readlane_f64(vi, j);
                    if (((j - c - 1) & 3) == 0) pa = __builtin_fma(a[j], vj, pa);
                    else if (((j - c - 1) & 3) == 1) pb = __builtin_fma(a[j], vj, pb);
                    else if (((j - c - 1) & 3) == 2) pc2 = __builtin_fma(a[j], vj, pc2);
                    else pd = __builtin_fma(a[j], vj, pd);
                }
                double pi = (pa + pb) + (pc2 + pd);
                pi = (lane > c) ? pi * tau : 0.0;
                const double kk = wave_sum(pi * vi);
                const double wi = pi - 0.5 * tau * kk * vi;
#pragma unroll
                for (int j = c + 1; j < NC; j++) a[j] -= vi * readlane_f64(wi, j) + wi * readlane_f64(vi, j);
                if (lane == 0) ee[c] = alpha;
            }
        }
    }
    double dv = 0.0, ev = 0.0;
#pragma unroll
    for (int j = 0; j < NC; j++) {
        dv = (lane == j) ? a[j] : dv;
        ev = (j == n - 2) ? a[j] : ev;
    }
    if (lane < n) dd[lane] = dv;
    ev = readlane_f64(ev, n - 1);
    if (lane == 0) ee[n - 2] = ev;
}

// ---- k_ipm_step: smallest eigenvalue of L^-1 dM L^-T, M in {X, Y} ---------------------------------------------------------------
// grid = 2 NB: workgroup 2b handles (X, dX), 2b+1 handles (Y, dY).  Cholesky of M, two triangular solves, Householder
// tridiagonalisation in LDS, Sturm-count multisection with one shift per thread (256-section per round).
__global__ __launch_bounds__(256) void k_ipm_step(const IpmBuf q) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double red[4];
    __shared__ int zc[2][4];
    const int b = blockIdx.x >> 1, which = blockIdx.x & 1;
    const IBlock k = q.blocks[b];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = k.n, n16 = (n + 15) & ~15, lda = n16 + 2;
    // Ws carries 8 extra zero columns and vv / pp are 72 long: the Householder loops run in unguarded chunks of 8
    double *Ls = lds, *Ws = Ls + lda * n16, *dinv = Ws + lda * (n16 + 8), *dd = dinv + n16, *ee = dd + n16, *vv = ee + n16, *pp = vv + 72;
    const double *Mg = (which ? q.Y : q.X) + k.xyoff, *dMg = (which ? q.dY : q.dX) + k.xyoff;
    const int i16 = tid & 15, j16 = tid >> 4;
    if (n == 1) {
        if (tid == 0) q.eig[b * 2 + which] = dMg[0] / Mg[0];
        return;
    }
    IPM_STAMP(0);
    // L = chol(M)
    for (int j0 = 0; j0 < n16; j0 += 16)
        for (int i0 = 0; i0 < n16; i0 += 16) {
            const int i = i0 + i16, j = j0 + j16;
            Ls[i + j * lda] = (i < n && j < n) ? ((i >= j) ? Mg[i + (long long)j * n] : 0.0) : ((i == j) ? 1.0 : 0.0);
        }
    ipm_load(Ws, lda, dMg, n, n16, tid);
    for (int e = tid; e < 8 * lda; e += 256) Ws[n16 * lda + e] = 0.0;
    __syncthreads();
    IPM_STAMP(1);
    const bool bad = lds_potrf(Ls, lda, dinv, n, wave, 4, lane);
    if (bad && lane == 0) atomicMin(q.info + 1, 1000000 + b);          // Cholesky failed in the step length computation
    __syncthreads();
    IPM_STAMP(2);
    // W = L^-1 dM L^-T: solve on the columns, then on the rows
    lds_trsm<false>(Ls, lda, dinv, Ws, 1, lda, n, n, wave, 4, lane);
    __syncthreads();
    lds_trsm<false>(Ls, lda, dinv, Ws, lda, 1, n, n, wave, 4, lane);
    __syncthreads();
    IPM_STAMP(3);
    // symmetrise
    for (int j0 = 0; j0 < n; j0 += 16)
        for (int i0 = 0; i0 < n; i0 += 16) {
            const int i = i0 + i16, j = j0 + j16;
            if (i < n && j < n && i > j) {
                const double v = 0.5 * (Ws[i + j * lda] + Ws[j + i * lda]);
                Ws[i + j * lda] = v;
                Ws[j + i * lda] = v;
            }
        }
    __syncthreads();
    IPM_STAMP(4);
    // Householder tridiagonalisation inside ONE wave (n <= 64): lane i owns row c+1+i of the trailing matrix, the vector v and
    // w go through LDS as broadcasts, the norms and dot products through wave butterflies -- no workgroup barriers in the
    // n-2 dependent steps.  Both triangles of the trailing matrix are kept.
    const bool in_regs = n <= 32;
    if (wave == 0 && in_regs) {
        if (n <= 16) ipm_householder_regs<16>(Ws, lda, n, dd, ee, lane);
        else if (n <= 24) ipm_householder_regs<24>(Ws, lda, n, dd, ee, lane);
        else ipm_householder_regs<32>(Ws, lda, n, dd, ee, lane);
    }
    if (wave == 0 && !in_regs) {
        for (int c = 0; c < n - 2; c++) {
            const int m = n - c - 1;                    // rows c+1 .. n-1 <-> lanes 0 .. m-1
            const double xi = (lane < m) ? Ws[(c + 1 + lane) + c * lda] : 0.0;
            const double ss = wave_sum(xi * xi);
            const double x0 = readlane_f64(xi, 0);
            const double nrm = sqrt(ss);
            if (nrm == 0.0) {
                if (lane == 0) ee[c] = 0.0;
                continue;
            }
            const double alpha = (x0 > 0.0) ? -nrm : nrm;
            const double v0 = x0 - alpha;
            const double vi = (lane == 0) ? v0 : xi;
            const double tau = 2.0 / (ss - x0 * x0 + v0 * v0);
            vv[lane] = (lane < m) ? vi : 0.0;           // zero padded: the loops below run in unguarded chunks of 8
            wave_sync();
            double pi = 0.0;
            const int m8 = (m + 7) & ~7;
            if (lane < m) {
                const double *row = Ws + (c + 1 + lane) + (c + 1) * lda;
                for (int j0 = 0; j0 < m8; j0 += 8) {    // 16 independent LDS reads in flight per chunk
                    double a[8], v[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) { a[u] = row[(j0 + u) * lda]; v[u] = vv[j0 + u]; }
#pragma unroll
                    for (int u = 0; u < 8; u++) pi += a[u] * v[u];
                }
                pi *= tau;
            }
            const double kk = wave_sum(pi * vi);
            const double wi = pi - 0.5 * tau * kk * vi;
            pp[lane] = (lane < m) ? wi : 0.0;
            wave_sync();
            if (lane < m) {
                double *row = Ws + (c + 1 + lane) + (c + 1) * lda;
                for (int j0 = 0; j0 < m8; j0 += 8) {
                    double a[8], v[8], w[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) { a[u] = row[(j0 + u) * lda]; v[u] = vv[j0 + u]; w[u] = pp[j0 + u]; }
#pragma unroll
                    for (int u = 0; u < 8; u++) row[(j0 + u) * lda] = a[u] - (vi * w[u] + wi * v[u]);
                }
            }
            if (lane == 0) ee[c] = alpha;
            wave_sync();
        }
    }
    __syncthreads();
    IPM_STAMP(5);
    if (!in_regs) {
        if (tid < n) dd[tid] = Ws[tid + tid * lda];
        if (tid == 0) ee[n - 2] = Ws[(n - 1) + (n - 2) * lda];
    }
    __syncthreads();
    // Gershgorin interval
    double lo = 1e300, hi = -1e300;
    if (tid < n) {
        const double r = ((tid > 0) ? fabs(ee[tid - 1]) : 0.0) + ((tid < n - 1) ? fabs(ee[tid]) : 0.0);
        lo = dd[tid] - r;
        hi = dd[tid] + r;
    }
    lo = -block_reduce_max(-lo, red);
    hi = block_reduce_max(hi, red);
    // Sturm sequence in product form, p_{i+1} = (d_i - s) p_i - e_{i-1}^2 p_{i-1}: the number of sign changes is the number of eigenvalues
    // below s.  No division on the dependent chain (one FMA per step instead of reciprocal + Newton + FMA: 23k -> cycles of the
    // seven rounds); d and e are scaled by a power of two (exact) so that |d - s| <= 2, e^2 <= 1, and the pair (p_i, p_{i-1}) is
    // renormalised by a power of two every four steps, so nothing overflows or underflows for n <= 64.
    const double scale = fmax(fmax(fabs(lo), fabs(hi)), 1e-290);
    const int sexp = __builtin_amdgcn_frexp_exp(scale);           // scale < 2^sexp
    {
        double *d2 = vv, *e2s = pp;           // 72 doubles each, free after the tridiagonalisation
        if (tid < 72) {
            d2[tid] = (tid < n) ? ldexp(dd[tid], -sexp) : 0.0;
            const double es = (tid >= 1 && tid < n) ? ldexp(ee[tid - 1], -sexp) : 0.0;
            e2s[tid] = es * es;
        }
    }
    lo = ldexp(lo, -sexp);
    hi = ldexp(hi, -sexp);
    __syncthreads();
    // multisection for the smallest eigenvalue: count(s) = number of eigenvalues < s
    lo -= 1e-3;
    hi += 1e-3;
    for (int round = 0; round < 7; round++) {      // 257^7 > 1e16: the bracket shrinks to rounding level
        const double h = (hi - lo) * (1.0 / 257.0);      // (an IEEE division is ~35 dependent VALU operations)
        const double sft = lo + h * (tid + 1);
        double pm = 1.0, pc = (vv[0] - sft) + 1e-300;     // p_0, p_1
        // sign changes counted on the sign bits with integer VALU operations (a compare pair per step would go through SGPR masks)
        auto hi32 = [](double v) { return (unsigned)(__double_as_longlong(v) >> 32); };
        unsigned cnt = hi32(pc) >> 31;
        auto step = [&](double tvu, double e2u) {
            // + 1e-300 (in the term that is off the dependent chain): an exact zero becomes a tiny positive p, from which the recurrence
            // continues correctly whether or not the next e^2 is zero; for every other value (|p| >= 1e-70 between renormalisations) it is a no-op
            const double pn = __builtin_fma(tvu, pc, __builtin_fma(-e2u, pm, 1e-300));
            cnt += (hi32(pn) ^ hi32(pc)) >> 31;
            pm = pc;
            pc = pn;
        };
        int i0 = 1;
        double dn[4], en[4];                        // the next chunk's d and e^2 (broadcast LDS reads) are in flight while this chunk's chain runs
#pragma unroll
        for (int u = 0; u < 4; u++) { dn[u] = vv[i0 + u]; en[u] = pp[i0 + u]; }
        for (; i0 + 4 <= n; i0 += 4) {
            double tv[4], e2[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { tv[u] = dn[u] - sft; e2[u] = en[u]; }
#pragma unroll
            for (int u = 0; u < 4; u++) { dn[u] = vv[i0 + 4 + u]; en[u] = pp[i0 + 4 + u]; }    // 72-long buffers: no guard
#pragma unroll
            for (int u = 0; u < 4; u++) step(tv[u], e2[u]);
            const int ex = __builtin_amdgcn_frexp_exp(pc);
            pc = ldexp(pc, -ex);
            pm = ldexp(pm, -ex);
        }
        if (i0 < n) step(dn[0] - sft, en[0]);          // at most three steps left (uniform)
        if (i0 + 1 < n) step(dn[1] - sft, en[1]);
        if (i0 + 2 < n) step(dn[2] - sft, en[2]);
        // The counts are monotone in the shift, so the number of shifts with count 0 is the index of the sub-interval that holds the
        // smallest eigenvalue: one ballot per wave, four numbers through LDS, ONE barrier per round (buffers alternate by round).
        const unsigned long long zero = __ballot(cnt == 0u);
        if (lane == 0) zc[round & 1][wave] = __popcll(zero);
        __syncthreads();
        const int idx = (zc[round & 1][0] + zc[round & 1][1]) + (zc[round & 1][2] + zc[round & 1][3]);
        const double nlo = lo + h * idx;
        hi = (idx == 256) ? hi : lo + h * (idx + 1);
        lo = nlo;
    }
    lo = ldexp(lo, sexp);
    hi = ldexp(hi, sexp);
    IPM_STAMP(6);
    if (tid == 0) q.eig[b * 2 + which] = 0.5 * (lo + hi);
}

// ---- k_ipm_update: iterate update + objective / complementarity dots ------------------------------------------------------
__global__ __launch_bounds__(256) void k_ipm_update(const IpmBuf q, const IpmParams prm, int nblocks_grid, int row0, int inl) {
    __shared__ double red[4];
    __shared__ double sh_sc[3];
    if (threadIdx.x == 0) {
        if (inl) ipm_scalar_stage(q, prm, 3, 0, 0);      // step lengths from the eigenvalues, identical in every workgroup
        sh_sc[0] = q.scal[SC_ERRCODE]; sh_sc[1] = q.scal[SC_ALPHA_P]; sh_sc[2] = q.scal[SC_ALPHA_D];
    }
    __syncthreads();
    // after a failed factorisation or a too short step the iterate is left as it is (the reference returns the current
    // iterate, src/solver.jl:470-475, 594-623); the directions may then hold NaNs, so they are not even multiplied by 0
    const bool skip = sh_sc[0] != 0.0;
    const double ap = sh_sc[1], ad = sh_sc[2];
    double cy = 0.0, xy = 0.0, cx = 0.0, by = 0.0;
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < q.xylen; e += 256ll * nblocks_grid) {
        const double Xn = skip ? q.X[e] : q.X[e] + ad * q.dX[e], Yn = skip ? q.Y[e] : q.Y[e] + ap * q.dY[e];
        q.X[e] = Xn;
        q.Y[e] = Yn;
        cy += q.C[e] * Yn;
        xy += Xn * Yn;
    }
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < q.xlen; e += 256ll * nblocks_grid) {
        const double xn = skip ? q.x[e] : q.x[e] + ad * q.dx[e];
        q.x[e] = xn;
        cx += q.c[e] * xn;
    }
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < q.N; e += 256ll * nblocks_grid) {
        const double yn = skip ? q.y[e] : q.y[e] + ap * q.dy[e];
        q.y[e] = yn;
        by += q.b[e] * yn;
    }
    cy = block_reduce_sum(cy, red);
    xy = block_reduce_sum(xy, red);
    cx = block_reduce_sum(cx, red);
    by = block_reduce_sum(by, red);
    if (threadIdx.x == 0) {
        double *pt = q.part + (long long)(row0 + blockIdx.x) * 8;
        pt[5] = cy; pt[6] = cx; pt[7] = by; pt[0] = xy;
        if (blockIdx.x == 0) { q.scal[SC_INFO0] = (double)q.info[0]; q.scal[SC_INFO1] = (double)q.info[1]; }
    }
}

}  // namespace clrs
