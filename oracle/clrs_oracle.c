/*
 * clrs_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the interior-point hot path of ClusteredLowRankSolver.jl v2.1.0 and of
 * the solver loop around it, used ONLY as the checker for the HIP path (tests/, bench.py's
 * cpu_baseline leg, __graft_entry__.smoke()).  Nothing under clusteredlowranksolver.jl_amd/ may
 * link or call this file.
 *
 * The reference computes in Arb midpoints at `prec` bits (ordinary rounded multi-precision).  This
 * file is compiled twice with the scalar type REAL = double (libclrs_oracle_f64.so, also the "port"
 * CPU baseline) and REAL = __float128 (libclrs_oracle_f128.so, 113-bit, the high-precision checker).
 * Inputs come as (hi, lo) double pairs (value = hi + lo); outputs likewise.
 *
 * PARITY PINNING: the reference (Julia + Arblib/FLINT, unpinned, absent from /root/reference) can
 * be neither built nor imported here, and its tests hold no kernel-level fixtures.  The oracle is
 * pinned (tests/test_oracle_cpu.py)
 *   - end-to-end through the reference's own known answers that 113 bits can reach:
 *     delsarte(3,10,1/2) = 13.158314 (test/runtests_solver.jl:15), delsarte(8,3,1/2) = 240 (:86-87),
 *     three_point_spherical_codes(4,1//6,-1,4) = 10 (:26-27), x^2+1 -> 1 (README.md:149);
 *     cohnelkies(8,15) / Nsphere_packing(8,15,...) = pi^4/384 (:19-22) need the reference's 256-300 bits
 *     and are NOT reproduced;
 *   - kernel level against tests/golden/*.npz: 256-bit answers from an independent mpmath
 *     restatement of the definitions (dense trace formula for S, LU solve of the KKT system);
 *   - by structural identities (low-rank S == dense Tr(A_p X^-1 A_q Y), symmetry, definiteness).
 * Kernel-level parity (S, L, Q, dx, dy) against the reference's OWN intermediates stays "parity
 * unpinned": the reference exposes none.
 *
 * Functions cite the reference lines they follow (paths relative to /root/reference).
 * Matrices are column-major.  Indices are 0-based.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#if defined(ORACLE_MP)
/* multi-limb build (oracle/clrs_oracle_mp.cpp includes mpx.hpp first and compiles this file as C++):
 * the stand-in for the reference's Arb midpoints at `prec` bits (src/solver.jl:73,103) */
typedef mpx<ORACLE_MP> REAL;
#define RSQRT(x) mpx_sqrt(x)
#define RABS(x) mpx_abs(x)
#elif defined(ORACLE_QUAD)
#include <quadmath.h>
typedef __float128 REAL;
#define RSQRT(x) sqrtq(x)
#define RABS(x) fabsq(x)
#else
typedef double REAL;
#define RSQRT(x) sqrt(x)
#define RABS(x) fabs(x)
#endif

typedef long long i64;

/* ------------------------------------------------------------------------------------------ */
/* input description (mirrors clrs_amd.sdp.FlatSDP)                                            */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int n_clusters, n_free;
    const int *cluster_P;
    const double *B, *B_lo, *c, *c_lo, *b, *b_lo, *C, *C_lo;
    int maximize;
    double constant, constant_lo;
    int n_blocks;
    const int *block_cluster, *block_m, *block_delta, *block_kind;
    const i64 *term_ptr;
    const int *term_p, *term_r, *term_s, *term_rank;
    const double *term_lambda, *term_lambda_lo;
    const i64 *term_vec_ptr;
    const double *term_vs, *term_vs_lo, *term_ws, *term_ws_lo;
    const i64 *dense_ptr;
    const int *dense_p;
    const i64 *dense_A_ptr;
    const double *dense_A, *dense_A_lo;
    /* limb planes 3.. of the data arrays (FlatSDP.tails: the sampled problem at the working precision, as the reference holds it at `prec` bits --
     * convert_to_prec, src/interface.jl:1078-1112): n_tail planes each, planar with the array's own length; NULL / 0 = none */
    int n_tail;
    const double *B_tail, *c_tail, *b_tail, *C_tail, *term_lambda_tail, *term_vs_tail, *term_ws_tail, *dense_A_tail;
} oracle_sdp;

typedef struct {
    /* per block */
    int j, m, delta, n, kind;
    i64 off;              /* offset in X/Y layout */
    i64 t0, t1;           /* term range */
    i64 d0, d1;           /* dense entry range */
    /* tables of precompute_matrices_bilinear_pairings (src/solver.jl:985-1059), per sub-block r */
    int *UR, *UL;         /* [m] number of unique right / left vectors */
    REAL **rightvecs;     /* [m] delta x UR[r] */
    REAL **leftvecs;      /* [m] UL[r] x delta (rows are w^T, as in the reference) */
    REAL **bpY, **bpX;    /* [m*m] (s,r) -> UL[s] x UR[r] */
} oblock;

typedef struct {
    int J, N, NB;
    int maximize;
    REAL constant;
    int *P;
    i64 *coff;            /* cluster offsets in x */
    i64 *Soff;
    REAL **B;             /* [J] P_j x N */
    REAL *c, *b, *C;
    oblock *blk;
    i64 T, D;
    int *tp, *tr, *ts, *tk;
    REAL *tlam;
    i64 *tvptr;
    REAL *tvs, *tws;
    int *ridx;            /* pointers_right[r][(s,p,k)] for term t=(p,r,s,k) */
    int *lidx;            /* pointers_left[r][(s,p,k)]  for term t=(p,r,s,k) */
    i64 *partner;         /* index of term (p,s,r,k) */
    int *dp;
    i64 *dAptr;
    REAL *dA;
    i64 xylen, xlen, Slen;
    /* factorisation state (overwritten every iteration, src/solver.jl:298-317) */
    REAL *S;              /* concatenated S_j, holds L_j after factor */
    REAL **LinvB;         /* [J] P_j x N */
    REAL *Q;              /* N x N, holds L_Q after factor */
    REAL *AY;             /* [T] w^T Y v per term */
    REAL *Xinv_tmp;       /* largest block scratch x3 */
    int maxn;
    /* trajectory snapshots of oracle_solvesdp (oracle_set_snapshots): iterates at the top of the listed
     * iterations and the predictor's right-hand sides, as k-limb planar arrays */
    int snap_n, snap_k, snap_count;
    const int *snap_it;
    double *snap_X, *snap_Y, *snap_rx, *snap_ry;
    REAL last_obj[3];     /* dual objective, primal objective, duality gap of the last oracle_solvesdp at full working precision */
    /* starting iterate of the next oracle_solvesdp (oracle_set_start; the dualsol / primalsol keywords, src/solver.jl:202-239) */
    int start_k;
    const double *start_x, *start_y, *start_X, *start_Y;
} octx;

static REAL ld(const double *hi, const double *lo, i64 i) { return lo ? (REAL)hi[i] + (REAL)lo[i] : (REAL)hi[i]; }
static void st(double *hi, double *lo, i64 i, REAL v) {
    double h = (double)v;
    if (hi) hi[i] = h;
    if (lo) lo[i] = (double)(v - (REAL)h);
}
static REAL *ralloc(i64 n) { REAL *p = (REAL *)calloc((size_t)(n > 0 ? n : 1), sizeof(REAL)); return p; }
static REAL *rload(const double *hi, const double *lo, i64 n) {
    REAL *p = ralloc(n);
    for (i64 i = 0; i < n; i++) p[i] = ld(hi, lo, i);
    return p;
}
/* the same with nt more limb planes behind (hi, lo): tail[t * len + off + i], t < nt (the planes have the length of the whole array, `off` is where this
 * piece starts in it) */
static REAL *rload_t(const double *hi, const double *lo, const double *tail, int nt, i64 len, i64 off, i64 n) {
    REAL *p = rload(hi, lo, n);
    if (tail)
        for (int t = 0; t < nt; t++)
            for (i64 i = 0; i < n; i++) p[i] = p[i] + (REAL)tail[(i64)t * len + off + i];
    return p;
}

/* k-limb planar arrays: value[i] = sum_l p[l * len + i] (limb 0 = the value rounded to fp64, then the
 * rounded remainders): the interchange format of the multi-word HIP path (include/clrs_hip.h, clrs_mw_*) */
static REAL ldk(const double *p, i64 len, int k, i64 i) {
    REAL v = (REAL)p[i];
    for (int l = 1; l < k; l++) v = v + (REAL)p[(i64)l * len + i];
    return v;
}
static void stk(double *p, i64 len, int k, i64 i, REAL v) {
    for (int l = 0; l < k; l++) {
        double h = (double)v;
        p[(i64)l * len + i] = h;
        v = v - (REAL)h;
    }
}
static REAL *rloadk(const double *p, i64 len, int k) {
    REAL *r = ralloc(len);
    for (i64 i = 0; i < len; i++) r[i] = ldk(p, len, k, i);
    return r;
}

/* ------------------------------------------------------------------------------------------ */
/* dense kernels (src/tools.jl)                                                                */
/* ------------------------------------------------------------------------------------------ */

/* C(MxN) = A(MxK) B(KxN)           -- matmul_threaded!, src/tools.jl:175-209 (split over columns) */
static void gemm_nn(int M, int N, int K, const REAL *A, int lda, const REAL *B, int ldb, REAL *C, int ldc) {
#pragma omp parallel for schedule(static) if ((i64)M * N * K > 32768)
    for (int j = 0; j < N; j++) {
        REAL *c = C + (i64)j * ldc;
        for (int i = 0; i < M; i++) c[i] = 0;
        for (int k = 0; k < K; k++) {
            REAL bkj = B[k + (i64)j * ldb];
            const REAL *a = A + (i64)k * lda;
            for (int i = 0; i < M; i++) c[i] += a[i] * bkj;
        }
    }
}
/* C(MxN) = A(KxM)^T B(KxN) */
static void gemm_tn(int M, int N, int K, const REAL *A, int lda, const REAL *B, int ldb, REAL *C, int ldc) {
#pragma omp parallel for schedule(static) if ((i64)M * N * K > 32768)
    for (int j = 0; j < N; j++) {
        const REAL *b = B + (i64)j * ldb;
        for (int i = 0; i < M; i++) {
            const REAL *a = A + (i64)i * lda;
            REAL s = 0;
            for (int k = 0; k < K; k++) s += a[k] * b[k];
            C[i + (i64)j * ldc] = s;
        }
    }
}

/* approx_cholesky!, src/tools.jl:75-107: row-by-row lower Cholesky, strict upper zeroed,
 * returns 1 on success, 0 when a pivot is not positive. */
static int cholesky_lower(int n, REAL *A, int lda) {
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < i; j++) {
            REAL s = A[i + (i64)j * lda];
            for (int k = 0; k < j; k++) s -= A[i + (i64)k * lda] * A[j + (i64)k * lda];
            A[i + (i64)j * lda] = s / A[j + (i64)j * lda];
        }
        REAL d = A[i + (i64)i * lda];
        for (int k = 0; k < i; k++) d -= A[i + (i64)k * lda] * A[i + (i64)k * lda];
        if (!(d > 0)) return 0;
        A[i + (i64)i * lda] = RSQRT(d);
    }
    for (int i = 0; i < n; i++)
        for (int j = i + 1; j < n; j++) A[i + (i64)j * lda] = 0;
    return 1;
}
/* B <- L^-1 B    (Arblib.approx_solve_tril!, call sites src/solver.jl:1258,1538) */
static void trsm_lower(int n, int nrhs, const REAL *L, int ldl, REAL *B, int ldb) {
#pragma omp parallel for schedule(static) if ((i64)n * n * nrhs > 65536)
    for (int c = 0; c < nrhs; c++) {
        REAL *x = B + (i64)c * ldb;
        for (int i = 0; i < n; i++) {
            REAL s = x[i];
            for (int k = 0; k < i; k++) s -= L[i + (i64)k * ldl] * x[k];
            x[i] = s / L[i + (i64)i * ldl];
        }
    }
}
/* B <- L^-T B    (approx_solve_triu! on the transposed factor, src/solver.jl:1567-1572) */
static void trsm_lower_trans(int n, int nrhs, const REAL *L, int ldl, REAL *B, int ldb) {
#pragma omp parallel for schedule(static) if ((i64)n * n * nrhs > 65536)
    for (int c = 0; c < nrhs; c++) {
        REAL *x = B + (i64)c * ldb;
        for (int i = n - 1; i >= 0; i--) {
            REAL s = x[i];
            for (int k = i + 1; k < n; k++) s -= L[k + (i64)i * ldl] * x[k];
            x[i] = s / L[i + (i64)i * ldl];
        }
    }
}
/* B <- (L L^T)^-1 B   (Arblib.solve_cho_precomp!, src/solver.jl:1095,1507,1557,1605) */
static void potrs(int n, int nrhs, const REAL *L, int ldl, REAL *B, int ldb) {
    trsm_lower(n, nrhs, L, ldl, B, ldb);
    trsm_lower_trans(n, nrhs, L, ldl, B, ldb);
}
/* Xinv <- (L L^T)^-1  (Arblib.inv_cho_precomp!, src/solver.jl:1117) */
static void inv_from_chol(int n, const REAL *L, REAL *Xinv) {
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) Xinv[i + (i64)j * n] = (i == j) ? 1 : 0;
    potrs(n, n, L, n, Xinv, n);
    /* the result is symmetric up to rounding; Arb returns the symmetric inverse */
    for (int j = 0; j < n; j++)
        for (int i = j + 1; i < n; i++) {
            REAL v = (Xinv[i + (i64)j * n] + Xinv[j + (i64)i * n]) / 2;
            Xinv[i + (i64)j * n] = Xinv[j + (i64)i * n] = v;
        }
}
static REAL dotn(i64 n, const REAL *a, const REAL *b) {
    REAL s = 0;
    for (i64 i = 0; i < n; i++) s += a[i] * b[i];
    return s;
}

/* ------------------------------------------------------------------------------------------ */
/* context creation = precompute_matrices_bilinear_pairings (src/solver.jl:985-1059)           */
/* ------------------------------------------------------------------------------------------ */
static int vec_equal(const REAL *a, const REAL *b, int n) {
    for (int i = 0; i < n; i++)
        if (a[i] != b[i]) return 0;
    return 1;
}

octx *oracle_create(const oracle_sdp *d) {
    octx *o = (octx *)calloc(1, sizeof(octx));
    o->J = d->n_clusters; o->N = d->n_free; o->NB = d->n_blocks;
    o->maximize = d->maximize;
    o->constant = (REAL)d->constant + (REAL)d->constant_lo;
    o->P = (int *)malloc(sizeof(int) * (o->J + 1));
    o->coff = (i64 *)calloc(o->J + 1, sizeof(i64));
    o->Soff = (i64 *)calloc(o->J + 1, sizeof(i64));
    o->B = (REAL **)calloc(o->J + 1, sizeof(REAL *));
    o->LinvB = (REAL **)calloc(o->J + 1, sizeof(REAL *));
    i64 boff = 0, Blen_ = 0;
    for (int j = 0; j < o->J; j++) Blen_ += (i64)d->cluster_P[j] * o->N;      /* length of the whole B array (the plane length of its tail) */
    for (int j = 0; j < o->J; j++) {
        o->P[j] = d->cluster_P[j];
        o->coff[j + 1] = o->coff[j] + o->P[j];
        o->Soff[j + 1] = o->Soff[j] + (i64)o->P[j] * o->P[j];
        o->B[j] = rload_t(d->B + boff, d->B_lo ? d->B_lo + boff : NULL, d->B_tail, d->n_tail, Blen_, boff, (i64)o->P[j] * o->N);
        o->LinvB[j] = ralloc((i64)o->P[j] * o->N);
        boff += (i64)o->P[j] * o->N;
    }
    o->xlen = o->coff[o->J]; o->Slen = o->Soff[o->J];
    o->c = rload_t(d->c, d->c_lo, d->c_tail, d->n_tail, o->xlen, 0, o->xlen);
    o->b = rload_t(d->b, d->b_lo, d->b_tail, d->n_tail, o->N, 0, o->N);
    o->T = d->term_ptr[o->NB]; o->D = d->dense_ptr[o->NB];
    o->tp = (int *)malloc(sizeof(int) * (o->T + 1)); o->tr = (int *)malloc(sizeof(int) * (o->T + 1));
    o->ts = (int *)malloc(sizeof(int) * (o->T + 1)); o->tk = (int *)malloc(sizeof(int) * (o->T + 1));
    o->ridx = (int *)malloc(sizeof(int) * (o->T + 1)); o->lidx = (int *)malloc(sizeof(int) * (o->T + 1));
    o->partner = (i64 *)malloc(sizeof(i64) * (o->T + 1));
    o->tvptr = (i64 *)malloc(sizeof(i64) * (o->T + 1));
    memcpy(o->tp, d->term_p, sizeof(int) * o->T); memcpy(o->tr, d->term_r, sizeof(int) * o->T);
    memcpy(o->ts, d->term_s, sizeof(int) * o->T); memcpy(o->tk, d->term_rank, sizeof(int) * o->T);
    memcpy(o->tvptr, d->term_vec_ptr, sizeof(i64) * (o->T + 1));
    o->tlam = rload_t(d->term_lambda, d->term_lambda_lo, d->term_lambda_tail, d->n_tail, o->T, 0, o->T);
    o->tvs = rload_t(d->term_vs, d->term_vs_lo, d->term_vs_tail, d->n_tail, o->tvptr[o->T], 0, o->tvptr[o->T]);
    o->tws = rload_t(d->term_ws, d->term_ws_lo, d->term_ws_tail, d->n_tail, o->tvptr[o->T], 0, o->tvptr[o->T]);
    o->dp = (int *)malloc(sizeof(int) * (o->D + 1));
    memcpy(o->dp, d->dense_p, sizeof(int) * o->D);
    o->dAptr = (i64 *)malloc(sizeof(i64) * (o->D + 1));
    memcpy(o->dAptr, d->dense_A_ptr, sizeof(i64) * (o->D + 1));
    o->dA = rload_t(d->dense_A, d->dense_A_lo, d->dense_A_tail, d->n_tail, o->dAptr[o->D], 0, o->dAptr[o->D]);
    o->blk = (oblock *)calloc(o->NB + 1, sizeof(oblock));
    i64 off = 0;
    o->maxn = 1;
    for (int b = 0; b < o->NB; b++) {
        oblock *k = &o->blk[b];
        k->j = d->block_cluster[b]; k->m = d->block_m[b]; k->delta = d->block_delta[b];
        k->n = k->m * k->delta; k->kind = d->block_kind[b];
        k->off = off; off += (i64)k->n * k->n;
        k->t0 = d->term_ptr[b]; k->t1 = d->term_ptr[b + 1];
        k->d0 = d->dense_ptr[b]; k->d1 = d->dense_ptr[b + 1];
        if (k->n > o->maxn) o->maxn = k->n;
        if (k->kind != 0) continue;
        int m = k->m, dl = k->delta;
        k->UR = (int *)calloc(m, sizeof(int)); k->UL = (int *)calloc(m, sizeof(int));
        k->rightvecs = (REAL **)calloc(m, sizeof(REAL *)); k->leftvecs = (REAL **)calloc(m, sizeof(REAL *));
        i64 nt = k->t1 - k->t0;
        for (int r = 0; r < m; r++) {
            /* unique_idx (src/tools.jl:128-145): first occurrence wins, exact equality */
            i64 *uniqR = (i64 *)malloc(sizeof(i64) * (nt + 1)), *uniqL = (i64 *)malloc(sizeof(i64) * (nt + 1));
            int nR = 0, nL = 0;
            for (i64 t = k->t0; t < k->t1; t++) {
                if (o->tr[t] != r) continue;
                const REAL *v = o->tvs + o->tvptr[t], *w = o->tws + o->tvptr[t];
                int f = -1;
                for (int u = 0; u < nR; u++)
                    if (vec_equal(o->tvs + o->tvptr[uniqR[u]], v, dl)) { f = u; break; }
                if (f < 0) { uniqR[nR] = t; f = nR++; }
                o->ridx[t] = f;
                f = -1;
                for (int u = 0; u < nL; u++)
                    if (vec_equal(o->tws + o->tvptr[uniqL[u]], w, dl)) { f = u; break; }
                if (f < 0) { uniqL[nL] = t; f = nL++; }
                o->lidx[t] = f;
            }
            k->UR[r] = nR; k->UL[r] = nL;
            k->rightvecs[r] = ralloc((i64)dl * nR);
            k->leftvecs[r] = ralloc((i64)dl * nL);
            for (int u = 0; u < nR; u++) memcpy(k->rightvecs[r] + (i64)u * dl, o->tvs + o->tvptr[uniqR[u]], sizeof(REAL) * dl);
            for (int u = 0; u < nL; u++)
                for (int i = 0; i < dl; i++) k->leftvecs[r][u + (i64)i * nL] = o->tws[o->tvptr[uniqL[u]] + i];
            free(uniqR); free(uniqL);
        }
        /* partner term (p, s, r, k): the reference relies on A[r,s][p] = A[s,r][p]^T (src/solver.jl:1009) */
        for (i64 t = k->t0; t < k->t1; t++) {
            o->partner[t] = -1;
            for (i64 u = k->t0; u < k->t1; u++)
                if (o->tp[u] == o->tp[t] && o->tr[u] == o->ts[t] && o->ts[u] == o->tr[t] && o->tk[u] == o->tk[t]) { o->partner[t] = u; break; }
            if (o->partner[t] < 0) { fprintf(stderr, "oracle: block %d term %lld has no transposed partner\n", b, t); return NULL; }
        }
        k->bpY = (REAL **)calloc(m * m, sizeof(REAL *)); k->bpX = (REAL **)calloc(m * m, sizeof(REAL *));
        for (int s = 0; s < m; s++)
            for (int r = 0; r < m; r++) {
                k->bpY[s + r * m] = ralloc((i64)k->UL[s] * k->UR[r]);
                k->bpX[s + r * m] = ralloc((i64)k->UL[s] * k->UR[r]);
            }
    }
    o->xylen = off;
    o->C = rload_t(d->C, d->C_lo, d->C_tail, d->n_tail, o->xylen, 0, o->xylen);
    o->S = ralloc(o->Slen);
    o->Q = ralloc((i64)o->N * o->N);
    o->AY = ralloc(o->T);
    o->Xinv_tmp = ralloc(3 * (i64)o->maxn * o->maxn);
    return o;
}

void oracle_destroy(octx *o) {
    if (!o) return;
    for (int j = 0; j < o->J; j++) { free(o->B[j]); free(o->LinvB[j]); }
    for (int b = 0; b < o->NB; b++) {
        oblock *k = &o->blk[b];
        if (k->kind != 0) continue;
        for (int r = 0; r < k->m; r++) { free(k->rightvecs[r]); free(k->leftvecs[r]); }
        for (int i = 0; i < k->m * k->m; i++) { free(k->bpY[i]); free(k->bpX[i]); }
        free(k->UR); free(k->UL); free(k->rightvecs); free(k->leftvecs); free(k->bpY); free(k->bpX);
    }
    free(o->P); free(o->coff); free(o->Soff); free(o->B); free(o->LinvB); free(o->c); free(o->b); free(o->C);
    free(o->blk); free(o->tp); free(o->tr); free(o->ts); free(o->tk); free(o->tlam); free(o->tvptr);
    free(o->tvs); free(o->tws); free(o->ridx); free(o->lidx); free(o->partner); free(o->dp); free(o->dAptr);
    free(o->dA); free(o->S); free(o->Q); free(o->AY); free(o->Xinv_tmp);
    free(o);
}

/* unique-vector counts per (block, r), for tests of the de-duplication (a2) */
int oracle_unique_counts(const octx *o, int b, int *UR, int *UL) {
    const oblock *k = &o->blk[b];
    if (k->kind != 0) return 0;
    for (int r = 0; r < k->m; r++) { UR[r] = k->UR[r]; UL[r] = k->UL[r]; }
    return k->m;
}

/* ------------------------------------------------------------------------------------------ */
/* compute_S_integrated!  (src/solver.jl:1062-1226), REAL version on internal buffers          */
/* ------------------------------------------------------------------------------------------ */
/* The reference's `matmul_prec` (src/solver.jl:125, 304, 312-313): part_r and the bilinear pairings are matrices of matmul_prec bits and their products
 * run at matmul_prec (:1125-1143).  Here: while the pairing products run, the working precision of the multi-limb type is oracle_matmul_bits_ (0 = no change). */
static int oracle_matmul_bits_ = 0;
#if defined(ORACLE_MP)
#define MATMUL_PREC_BEGIN const int matmul_save_ = mpx_prec_bits; if (oracle_matmul_bits_ > 0) mpx_prec_bits = oracle_matmul_bits_;
#define MATMUL_PREC_END mpx_prec_bits = matmul_save_;
#else
#define MATMUL_PREC_BEGIN
#define MATMUL_PREC_END
#endif
static void schur_assemble_real(octx *o, const REAL *Xchol, const REAL *Y) {
    REAL *Xinv = o->Xinv_tmp, *T1 = Xinv + (i64)o->maxn * o->maxn, *T2 = T1 + (i64)o->maxn * o->maxn;
    memset(o->S, 0, sizeof(REAL) * o->Slen);                          /* :1082 */
    for (int b = 0; b < o->NB; b++) {
        oblock *k = &o->blk[b];
        int n = k->n, m = k->m, dl = k->delta, P = o->P[k->j];
        REAL *S = o->S + o->Soff[k->j];
        const REAL *Lx = Xchol + k->off, *Yb = Y + k->off;
        if (k->kind != 0) {                                           /* dense branch :1089-1104 */
            for (i64 e = k->d0; e < k->d1; e++) {
                int p = o->dp[e];
                memcpy(T1, o->dA + o->dAptr[e], sizeof(REAL) * n * n);
                potrs(n, n, Lx, n, T1, n);                            /* X^-1 A_p  :1095 */
                gemm_nn(n, n, n, T1, n, Yb, n, T2, n);                /* (X^-1 A_p) Y :1097 */
                for (i64 f = k->d0; f < k->d1; f++) {
                    int q = o->dp[f];
                    if (q < p) continue;                              /* upper triangle :1100 */
                    S[p + (i64)q * P] += dotn((i64)n * n, o->dA + o->dAptr[f], T2);   /* :1102 */
                }
            }
            continue;
        }
        inv_from_chol(n, Lx, Xinv);                                   /* method 3, :1117 */
        for (int r = 0; r < m; r++) {                                 /* :1121-1149 */
            int UR = k->UR[r];
            if (UR == 0) continue;
            REAL *part = ralloc((i64)n * UR);
            MATMUL_PREC_BEGIN
            for (int which = 0; which < 2; which++) {
                const REAL *M = which == 0 ? Yb : Xinv;
                gemm_nn(n, UR, dl, M + (i64)r * dl * n, n, k->rightvecs[r], dl, part, n);       /* :1125 / :1137 */
                for (int s = 0; s < m; s++) {
                    int UL = k->UL[s];
                    if (UL == 0) continue;
                    REAL *bp = (which == 0 ? k->bpY : k->bpX)[s + r * m];
                    gemm_nn(UL, UR, dl, k->leftvecs[s], UL, part + (i64)s * dl, n, bp, UL);     /* :1131 / :1143 */
                }
            }
            MATMUL_PREC_END
            free(part);
        }
        /* A_Y: w^T Y v per term (:1152-1170; here for every (r,s), the reference keeps s <= r) */
        for (i64 t = k->t0; t < k->t1; t++) {
            int r = o->tr[t], s = o->ts[t];
            o->AY[t] = k->bpY[r + s * m][o->lidx[t] + (i64)o->ridx[o->partner[t]] * k->UL[r]];
        }
        /* S accumulation (:1176-1212).  Terms are sorted by p, so each group of equal p owns row p of S
         * (the reference threads over p in the same way, :1179). */
        {
            i64 nt = k->t1 - k->t0, ng = 0;
            i64 *gs = (i64 *)malloc(sizeof(i64) * (nt + 2));
            for (i64 t = k->t0; t < k->t1; t++)
                if (t == k->t0 || o->tp[t] != o->tp[t - 1]) gs[ng++] = t;
            gs[ng] = k->t1;
#pragma omp parallel for schedule(dynamic, 4) if (nt > 256)
            for (i64 g = 0; g < ng; g++)
                for (i64 t1 = gs[g]; t1 < gs[g + 1]; t1++) {
                    int p = o->tp[t1], r1 = o->tr[t1], s1 = o->ts[t1];
                    int L1 = o->lidx[o->partner[t1]];    /* pointers_left[s1][(r1,p,rnk1)] */
                    int R1 = o->ridx[t1];                /* pointers_right[r1][(s1,p,rnk1)] */
                    for (i64 t2 = k->t0; t2 < k->t1; t2++) {
                        int q = o->tp[t2];
                        if (q < p) continue;                                  /* :1186 */
                        int r2 = o->tr[t2], s2 = o->ts[t2];
                        int L2 = o->lidx[o->partner[t2]], R2 = o->ridx[t2];
                        REAL bx = k->bpX[s1 + r2 * m][L1 + (i64)R2 * k->UL[s1]];
                        REAL by = k->bpY[s2 + r1 * m][L2 + (i64)R1 * k->UL[s2]];
                        S[p + (i64)q * P] += o->tlam[t1] * o->tlam[t2] * bx * by;   /* :1196-1199 */
                    }
                }
            free(gs);
        }
    }
    for (int j = 0; j < o->J; j++) {                                  /* symmetric!, :1222 */
        int P = o->P[j];
        REAL *S = o->S + o->Soff[j];
        for (int c = 0; c < P; c++)
            for (int r = c + 1; r < P; r++) S[r + (i64)c * P] = S[c + (i64)r * P];
    }
}

/* steps 3-4 of compute_T_decomposition! (src/solver.jl:1244-1279).
 * returns 0 ok; j+1 if S_j failed; -1 if Q failed. */
static int schur_factor_real(octx *o) {
    int fail = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int j = 0; j < o->J; j++) {
        int P = o->P[j];
        REAL *S = o->S + o->Soff[j];
        if (!cholesky_lower(P, S, P)) {                               /* :1246 */
#pragma omp critical
            { if (fail == 0 || j + 1 < fail) fail = j + 1; }
            continue;
        }
        memcpy(o->LinvB[j], o->B[j], sizeof(REAL) * (i64)P * o->N);
        trsm_lower(P, o->N, S, P, o->LinvB[j], P);                    /* :1258 */
    }
    if (fail) return fail;
    int N = o->N;
    memset(o->Q, 0, sizeof(REAL) * (i64)N * N);
    REAL *tmp = ralloc((i64)N * N);
    for (int j = 0; j < o->J; j++) {                                  /* Q = sum_j LinvB_j^T LinvB_j :1268-1269 */
        gemm_tn(N, N, o->P[j], o->LinvB[j], o->P[j], o->LinvB[j], o->P[j], tmp, N);
        for (i64 i = 0; i < (i64)N * N; i++) o->Q[i] += tmp[i];
    }
    free(tmp);
    if (N > 0 && !cholesky_lower(N, o->Q, N)) return -1;              /* :1274 */
    return 0;
}

/* solve stage of compute_search_direction! (src/solver.jl:1527-1582) */
static void schur_solve_real(octx *o, const REAL *rhs_x, const REAL *rhs_y, REAL *dx, REAL *dy) {
    int N = o->N;
    REAL *tx = ralloc(o->xlen);
    memcpy(tx, rhs_x, sizeof(REAL) * o->xlen);
    for (int i = 0; i < N; i++) dy[i] = rhs_y[i];                     /* :1550 */
    for (int j = 0; j < o->J; j++) {
        int P = o->P[j];
        REAL *t = tx + o->coff[j];
        trsm_lower(P, 1, o->S + o->Soff[j], P, t, P);                 /* :1538 */
        for (int i = 0; i < N; i++)                                   /* :1546, :1552 */
            dy[i] -= dotn(P, o->LinvB[j] + (i64)i * P, t);
    }
    if (N > 0) potrs(N, 1, o->Q, N, dy, N);                           /* :1557 */
    for (int j = 0; j < o->J; j++) {
        int P = o->P[j];
        REAL *t = tx + o->coff[j];
        for (int i = 0; i < N; i++) {                                 /* :1568-1569 */
            REAL a = dy[i];
            const REAL *col = o->LinvB[j] + (i64)i * P;
            for (int r = 0; r < P; r++) t[r] += col[r] * a;
        }
        trsm_lower_trans(P, 1, o->S + o->Soff[j], P, t, P);           /* :1571 */
        memcpy(dx + o->coff[j], t, sizeof(REAL) * P);
    }
    free(tx);
}

/* ------------------------------------------------------------------------------------------ */
/* public hot-path API (hi/lo doubles)                                                         */
/* ------------------------------------------------------------------------------------------ */
void oracle_dims(const octx *o, i64 *xylen, i64 *xlen, i64 *Slen, i64 *T) {
    *xylen = o->xylen; *xlen = o->xlen; *Slen = o->Slen; *T = o->T;
}

/* approx_cholesky!(X_inv_blk, X_blk) for all blocks (src/solver.jl:388-399).
 * returns 0 or b+1 of the first failing block. */
int oracle_cholesky_blocks(octx *o, const double *X, const double *X_lo, double *L, double *L_lo) {
    REAL *W = rload(X, X_lo, o->xylen);
    int fail = 0;
    for (int b = 0; b < o->NB && !fail; b++)
        if (!cholesky_lower(o->blk[b].n, W + o->blk[b].off, o->blk[b].n)) fail = b + 1;
    for (i64 i = 0; i < o->xylen; i++) st(L, L_lo, i, W[i]);
    free(W);
    return fail;
}

void oracle_schur_assemble(octx *o, const double *Xchol, const double *Xchol_lo, const double *Y, const double *Y_lo,
                           double *S, double *S_lo, double *AY, double *AY_lo) {
    REAL *L = rload(Xchol, Xchol_lo, o->xylen), *Yr = rload(Y, Y_lo, o->xylen);
    schur_assemble_real(o, L, Yr);
    if (S) for (i64 i = 0; i < o->Slen; i++) st(S, S_lo, i, o->S[i]);
    if (AY) for (i64 i = 0; i < o->T; i++) st(AY, AY_lo, i, o->AY[i]);
    free(L); free(Yr);
}

int oracle_schur_factor(octx *o) { return schur_factor_real(o); }

/* copy out L_j (concatenated like S), LinvB (concatenated per cluster, col-major P_j x N), L_Q */
void oracle_get_factor(octx *o, double *L, double *L_lo, double *LinvB, double *LinvB_lo, double *Q, double *Q_lo) {
    if (L) for (i64 i = 0; i < o->Slen; i++) st(L, L_lo, i, o->S[i]);
    if (LinvB) {
        i64 off = 0;
        for (int j = 0; j < o->J; j++) {
            i64 n = (i64)o->P[j] * o->N;
            for (i64 i = 0; i < n; i++) st(LinvB, LinvB_lo, off + i, o->LinvB[j][i]);
            off += n;
        }
    }
    if (Q) for (i64 i = 0; i < (i64)o->N * o->N; i++) st(Q, Q_lo, i, o->Q[i]);
}

void oracle_schur_solve(octx *o, const double *rx, const double *rx_lo, const double *ry, const double *ry_lo,
                        double *dx, double *dx_lo, double *dy, double *dy_lo) {
    REAL *a = rload(rx, rx_lo, o->xlen), *b = rload(ry, ry_lo, o->N);
    REAL *x = ralloc(o->xlen), *y = ralloc(o->N);
    schur_solve_real(o, a, b, x, y);
    for (i64 i = 0; i < o->xlen; i++) st(dx, dx_lo, i, x[i]);
    for (i64 i = 0; i < o->N; i++) st(dy, dy_lo, i, y[i]);
    free(a); free(b); free(x); free(y);
}


/* --- the same entry points on k-limb planar arrays (multi-word parity checks) ---------------- */
int oracle_cholesky_blocks_mw(octx *o, int k, const double *X, double *L) {
    REAL *W = rloadk(X, o->xylen, k);
    int fail = 0;
    for (int b = 0; b < o->NB && !fail; b++)
        if (!cholesky_lower(o->blk[b].n, W + o->blk[b].off, o->blk[b].n)) fail = b + 1;
    for (i64 i = 0; i < o->xylen; i++) stk(L, o->xylen, k, i, W[i]);
    free(W);
    return fail;
}
void oracle_schur_assemble_mw(octx *o, int k, const double *Xchol, const double *Y, double *S, double *AY) {
    REAL *L = rloadk(Xchol, o->xylen, k), *Yr = rloadk(Y, o->xylen, k);
    schur_assemble_real(o, L, Yr);
    if (S) for (i64 i = 0; i < o->Slen; i++) stk(S, o->Slen, k, i, o->S[i]);
    if (AY) for (i64 i = 0; i < o->T; i++) stk(AY, o->T, k, i, o->AY[i]);
    free(L); free(Yr);
}
/* load S (S layout) into the context, e.g. to factor a matrix another implementation assembled */
void oracle_set_S_mw(octx *o, int k, const double *S) {
    for (i64 i = 0; i < o->Slen; i++) o->S[i] = ldk(S, o->Slen, k, i);
}
void oracle_get_factor_mw(octx *o, int k, double *L, double *LinvB, double *Q) {
    if (L) for (i64 i = 0; i < o->Slen; i++) stk(L, o->Slen, k, i, o->S[i]);
    if (LinvB) {
        i64 off = 0, tot = (i64)o->xlen * o->N;
        for (int j = 0; j < o->J; j++) {
            i64 n = (i64)o->P[j] * o->N;
            for (i64 i = 0; i < n; i++) stk(LinvB, tot, k, off + i, o->LinvB[j][i]);
            off += n;
        }
    }
    if (Q) for (i64 i = 0; i < (i64)o->N * o->N; i++) stk(Q, (i64)o->N * o->N, k, i, o->Q[i]);
}
void oracle_schur_solve_mw(octx *o, int k, const double *rx, const double *ry, double *dx, double *dy) {
    REAL *a = rloadk(rx, o->xlen, k), *b = rloadk(ry, o->N, k);
    REAL *x = ralloc(o->xlen), *y = ralloc(o->N);
    schur_solve_real(o, a, b, x, y);
    for (i64 i = 0; i < o->xlen; i++) stk(dx, o->xlen, k, i, x[i]);
    for (i64 i = 0; i < o->N; i++) stk(dy, o->N, k, i, y[i]);
    free(a); free(b); free(x); free(y);
}
/* Backward error of a candidate solution (dx, dy) of the Newton system written at src/solver.jl:1527,
 *     [S -B; B^T 0] (dx; dy) = (rhs_x; rhs_y),
 * evaluated at the oracle's working precision from S given as k-limb planes (S layout; the matrix BEFORE its factorisation):
 *     out[0] = max_i |S dx - B dy - rhs_x|_i / max_i (|S| |dx| + |B| |dy| + |rhs_x|)_i
 *     out[1] = max_j |B^T dx - rhs_y|_j     / max_j (|B^T| |dx| + |rhs_y|)_j          (0 when N = 0)
 * Normwise and free of the conditioning of S: a forward-error comparison of (dx, dy) loses log2 cond(S) bits, this does not. */
void oracle_kkt_backward_error_mw(octx *o, int k, const double *S, const double *dx, const double *dy, const double *rx,
                                  const double *ry, double *out) {
    int N = o->N;
    REAL *Sr = rloadk(S, o->Slen, k), *x = rloadk(dx, o->xlen, k), *y = rloadk(dy, N, k);
    REAL *a = rloadk(rx, o->xlen, k), *b = rloadk(ry, N, k);
    REAL num_x = 0, den_x = 0, num_y = 0, den_y = 0;
    REAL *ry_acc = ralloc(N), *ry_abs = ralloc(N);
    for (int j = 0; j < o->J; j++) {
        int P = o->P[j];
        const REAL *Sj = Sr + o->Soff[j], *Bj = o->B[j], *xj = x + o->coff[j], *aj = a + o->coff[j];
        for (int i = 0; i < P; i++) {
            REAL r = 0, m = 0;
            for (int q = 0; q < P; q++) { REAL t = Sj[i + (i64)q * P] * xj[q]; r = r + t; m = m + RABS(t); }
            for (int c = 0; c < N; c++) { REAL t = Bj[i + (i64)c * P] * y[c]; r = r - t; m = m + RABS(t); }
            r = r - aj[i]; m = m + RABS(aj[i]);
            r = RABS(r);
            if (r > num_x) num_x = r;
            if (m > den_x) den_x = m;
        }
        for (int c = 0; c < N; c++)
            for (int i = 0; i < P; i++) { REAL t = Bj[i + (i64)c * P] * xj[i]; ry_acc[c] = ry_acc[c] + t; ry_abs[c] = ry_abs[c] + RABS(t); }
    }
    for (int c = 0; c < N; c++) {
        REAL r = RABS(ry_acc[c] - b[c]), m = ry_abs[c] + RABS(b[c]);
        if (r > num_y) num_y = r;
        if (m > den_y) den_y = m;
    }
    out[0] = den_x > (REAL)0 ? (double)(num_x / den_x) : 0.0;
    out[1] = den_y > (REAL)0 ? (double)(num_y / den_y) : 0.0;
    free(Sr); free(x); free(y); free(a); free(b); free(ry_acc); free(ry_abs);
}
/* snapshots of the next oracle_solvesdp: at the listed iterations (1-based, ascending) the iterate (X, Y) at the
 * top of the iteration and the predictor's right-hand sides (rhs_x, rhs_y) are stored as k-limb planar arrays,
 * snapshot s at offset s * k * len of each buffer.  n = 0 switches them off. */
void oracle_set_snapshots(octx *o, int n, const int *iters, int k, double *X, double *Y, double *rx, double *ry) {
    o->snap_n = n; o->snap_it = iters; o->snap_k = k; o->snap_count = 0;
    o->snap_X = X; o->snap_Y = Y; o->snap_rx = rx; o->snap_ry = ry;
}
int oracle_snapshot_count(const octx *o) { return o->snap_count; }
/* warm start (src/solver.jl:202-239: x, X from dualsol, y, Y from primalsol): the next oracle_solvesdp starts from these k-limb planar arrays
 * instead of x = 0, y = 0, X = omega_p I, Y = omega_d I.  k = 0 switches it off. */
void oracle_set_start(octx *o, int k, const double *x, const double *y, const double *X, const double *Y) {
    o->start_k = k; o->start_x = x; o->start_y = y; o->start_X = X; o->start_Y = Y;
}
/* dual objective, primal objective and duality gap the last oracle_solvesdp ended with, as k-limb planar numbers (out[l * 3 + i]) */
void oracle_last_objectives_mw(octx *o, int k, double *out) {
    for (int i = 0; i < 3; i++) stk(out, 3, k, i, o->last_obj[i]);
}

/* Dense restatement of S used as an independent structural check:
 * S[p,q] = sum_l Tr(A_p X^-1 A_q Y) with A_p = Matrix(::LowRankMat) (src/interface.jl:798-800). */
void oracle_schur_dense_check(octx *o, const double *Xchol, const double *Xchol_lo, const double *Y, const double *Y_lo,
                              double *S, double *S_lo) {
    REAL *L = rload(Xchol, Xchol_lo, o->xylen), *Yr = rload(Y, Y_lo, o->xylen);
    REAL *Sd = ralloc(o->Slen);
    for (int b = 0; b < o->NB; b++) {
        oblock *k = &o->blk[b];
        int n = k->n, dl = k->delta, P = o->P[k->j];
        REAL *Sj = Sd + o->Soff[k->j];
        REAL *A = ralloc((i64)P * n * n), *T = ralloc((i64)P * n * n), *T2 = ralloc((i64)n * n);
        char *has = (char *)calloc(P, 1);
        if (k->kind == 0) {
            for (i64 t = k->t0; t < k->t1; t++) {
                REAL *Ap = A + (i64)o->tp[t] * n * n;
                has[o->tp[t]] = 1;
                const REAL *v = o->tvs + o->tvptr[t], *w = o->tws + o->tvptr[t];
                for (int jj = 0; jj < dl; jj++)
                    for (int ii = 0; ii < dl; ii++)
                        Ap[(o->tr[t] * dl + ii) + (i64)(o->ts[t] * dl + jj) * n] += o->tlam[t] * v[ii] * w[jj];
            }
        } else {
            for (i64 e = k->d0; e < k->d1; e++) { memcpy(A + (i64)o->dp[e] * n * n, o->dA + o->dAptr[e], sizeof(REAL) * n * n); has[o->dp[e]] = 1; }
        }
        for (int p = 0; p < P; p++) {
            if (!has[p]) continue;
            memcpy(T2, A + (i64)p * n * n, sizeof(REAL) * n * n);
            potrs(n, n, L + k->off, n, T2, n);
            gemm_nn(n, n, n, T2, n, Yr + k->off, n, T + (i64)p * n * n, n);
        }
        for (int p = 0; p < P; p++)
            for (int q = 0; q < P; q++)
                if (has[p] && has[q]) {
                    /* Tr(A_q^T (X^-1 A_p Y)) */
                    Sj[p + (i64)q * P] += dotn((i64)n * n, A + (i64)q * n * n, T + (i64)p * n * n);
                }
        free(A); free(T); free(T2); free(has);
    }
    for (i64 i = 0; i < o->Slen; i++) st(S, S_lo, i, Sd[i]);
    free(L); free(Yr); free(Sd);
}

/* ------------------------------------------------------------------------------------------ */
/* the solver loop around the path (src/solver.jl:100-744) -- needed to pin the oracle          */
/* end-to-end against the reference's known answers                                            */
/* ------------------------------------------------------------------------------------------ */

/* sum_i a_i A_i per block (compute_weighted_A!, src/solver.jl:1410-1470) */
static void weighted_A(octx *o, const REAL *a, REAL *M) {
    memset(M, 0, sizeof(REAL) * o->xylen);
    for (int b = 0; b < o->NB; b++) {
        oblock *k = &o->blk[b];
        int n = k->n, dl = k->delta;
        REAL *Mb = M + k->off;
        const REAL *aj = a + o->coff[k->j];
        if (k->kind != 0) {
            for (i64 e = k->d0; e < k->d1; e++) {                                 /* :1424-1426 */
                REAL w = aj[o->dp[e]];
                const REAL *A = o->dA + o->dAptr[e];
                for (i64 i = 0; i < (i64)n * n; i++) Mb[i] += w * A[i];
            }
            continue;
        }
        for (i64 t = k->t0; t < k->t1; t++) {                                     /* :1433-1459, s <= r */
            int r = o->tr[t], s = o->ts[t];
            if (s > r) continue;
            REAL w = aj[o->tp[t]] * o->tlam[t];
            const REAL *v = o->tvs + o->tvptr[t], *ws = o->tws + o->tvptr[t];
            for (int jj = 0; jj < dl; jj++)
                for (int ii = 0; ii < dl; ii++) Mb[(r * dl + ii) + (i64)(s * dl + jj) * n] += w * v[ii] * ws[jj];
        }
        if (k->m > 1)                                                             /* symmetric!(:L) :1462-1465 */
            for (int c = 0; c < n; c++)
                for (int r = c + 1; r < n; r++) Mb[c + (i64)r * n] = Mb[r + (i64)c * n];
    }
}

/* <A_*, Z> (trace_A, src/solver.jl:1290-1366); Z symmetric */
static void trace_A(octx *o, const REAL *Z, REAL *res) {
    memset(res, 0, sizeof(REAL) * o->xlen);
    for (int b = 0; b < o->NB; b++) {
        oblock *k = &o->blk[b];
        int n = k->n, dl = k->delta;
        const REAL *Zb = Z + k->off;
        REAL *rj = res + o->coff[k->j];
        if (k->kind != 0) {
            for (i64 e = k->d0; e < k->d1; e++) rj[o->dp[e]] += dotn((i64)n * n, o->dA + o->dAptr[e], Zb);   /* :1304 */
            continue;
        }
        for (i64 t = k->t0; t < k->t1; t++) {
            int r = o->tr[t], s = o->ts[t];
            if (s > r) continue;                                                  /* :1310 */
            const REAL *v = o->tvs + o->tvptr[t], *w = o->tws + o->tvptr[t];
            REAL acc = 0;                                                          /* ones * (W o (Z[r,s] V)) :1334-1341 */
            for (int jj = 0; jj < dl; jj++) {
                REAL zc = 0;
                for (int ii = 0; ii < dl; ii++) zc += w[ii] * Zb[(r * dl + ii) + (i64)(s * dl + jj) * n];
                acc += zc * v[jj];
            }
            acc *= o->tlam[t];
            if (r != s) acc *= 2;                                                  /* :1354-1356 */
            rj[o->tp[t]] += acc;
        }
    }
}

/* <A_*, Y> from the stored pairings (trace_A with (Y, A_Y), src/solver.jl:1368-1407) */
static void trace_A_from_AY(octx *o, const REAL *Y, REAL *res) {
    memset(res, 0, sizeof(REAL) * o->xlen);
    for (int b = 0; b < o->NB; b++) {
        oblock *k = &o->blk[b];
        int n = k->n;
        REAL *rj = res + o->coff[k->j];
        if (k->kind != 0) {
            for (i64 e = k->d0; e < k->d1; e++) rj[o->dp[e]] += dotn((i64)n * n, o->dA + o->dAptr[e], Y + k->off);
            continue;
        }
        for (i64 t = k->t0; t < k->t1; t++) {
            if (o->ts[t] > o->tr[t]) continue;
            REAL v = o->AY[t] * o->tlam[t];
            if (o->tr[t] != o->ts[t]) v *= 2;
            rj[o->tp[t]] += v;
        }
    }
}

static REAL bdot(octx *o, const REAL *A, const REAL *B) { return dotn(o->xylen, A, B); }
static REAL maxabs(i64 n, const REAL *a) {
    REAL m = 0;
    for (i64 i = 0; i < n; i++) { REAL v = RABS(a[i]); if (v > m) m = v; }
    return m;
}

/* smallest eigenvalue of a symmetric n x n matrix in double (cyclic Jacobi).  The reference uses a
 * Float64 Lanczos with tol 1e-5 (src/solver.jl:1659); Jacobi is a deterministic stand-in. */
static double min_eig_sym(int n, double *A) {
    if (n == 1) return A[0];
    for (int sweep = 0; sweep < 60; sweep++) {
        double offd = 0, diag = 0;
        for (int j = 0; j < n; j++)
            for (int i = 0; i < n; i++) { if (i != j) offd += A[i + (i64)j * n] * A[i + (i64)j * n]; else diag += A[i + (i64)j * n] * A[i + (i64)j * n]; }
        if (offd <= 1e-30 * (diag + 1e-300)) break;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double apq = A[p + (i64)q * n];
                if (apq == 0.0) continue;
                double app = A[p + (i64)p * n], aqq = A[q + (i64)q * n];
                double theta = (aqq - app) / (2 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k2 = 0; k2 < n; k2++) {
                    double akp = A[k2 + (i64)p * n], akq = A[k2 + (i64)q * n];
                    A[k2 + (i64)p * n] = c * akp - s * akq;
                    A[k2 + (i64)q * n] = s * akp + c * akq;
                }
                for (int k2 = 0; k2 < n; k2++) {
                    double apk = A[p + (i64)k2 * n], aqk = A[q + (i64)k2 * n];
                    A[p + (i64)k2 * n] = c * apk - s * aqk;
                    A[q + (i64)k2 * n] = s * apk + c * aqk;
                }
            }
    }
    double mn = A[0];
    for (int i = 1; i < n; i++) if (A[i + (i64)i * n] < mn) mn = A[i + (i64)i * n];
    return mn;
}

/* compute_step_length (src/solver.jl:1620-1693); returns -1 on Cholesky failure */
static REAL step_length(octx *o, const REAL *M, const REAL *dM, REAL gamma, int unsafe_step) {
    REAL min_eig = 0;
    int first = 1;
    REAL *L = ralloc((i64)o->maxn * o->maxn), *W = ralloc((i64)o->maxn * o->maxn);
    double *Wd = (double *)malloc(sizeof(double) * (size_t)o->maxn * o->maxn);
    for (int b = 0; b < o->NB; b++) {
        oblock *k = &o->blk[b];
        int n = k->n;
        REAL e;
        if (n == 1) {
            e = dM[k->off] / M[k->off];                                           /* :1637-1641 */
        } else {
            memcpy(L, M + k->off, sizeof(REAL) * n * n);
            if (!cholesky_lower(n, L, n)) { free(L); free(W); free(Wd); return -1; }   /* :1644-1646 */
            memcpy(W, dM + k->off, sizeof(REAL) * n * n);
            trsm_lower(n, n, L, n, W, n);                                         /* :1651 */
            for (int c = 0; c < n; c++)                                           /* transpose :1652 */
                for (int r = c + 1; r < n; r++) { REAL t = W[r + (i64)c * n]; W[r + (i64)c * n] = W[c + (i64)r * n]; W[c + (i64)r * n] = t; }
            trsm_lower(n, n, L, n, W, n);                                         /* :1655 */
            for (int c = 0; c < n; c++)
                for (int r = 0; r < n; r++) Wd[r + (i64)c * n] = (double)((W[r + (i64)c * n] + W[c + (i64)r * n]) / 2);
            e = (REAL)min_eig_sym(n, Wd) - (REAL)1e-5;                            /* :1659-1662 */
        }
        if (first || e < min_eig) { min_eig = e; first = 0; }
    }
    free(L); free(W); free(Wd);
    if (min_eig > -gamma && !unsafe_step) return 1;                               /* :1688-1689 */
    return -gamma / min_eig;                                                      /* :1691 */
}

typedef struct {
    int maxiterations;
    double beta_infeasible, beta_feasible, gamma, omega_p, omega_d;
    double duality_gap_threshold, dual_error_threshold, primal_error_threshold;
    double max_complementary_gap, step_length_threshold;
    int need_dual_feasible, need_primal_feasible, safe_step, verbose;
    int correctoronly;      /* src/solver.jl:121, 370-374, 945 */
} oracle_params;

void oracle_default_params(oracle_params *p) {       /* src/solver.jl:103-126 */
    p->maxiterations = 500; p->beta_infeasible = 0.3; p->beta_feasible = 0.1; p->gamma = 0.9;
    p->omega_p = 1e10; p->omega_d = 1e10; p->duality_gap_threshold = 1e-15;
    p->dual_error_threshold = 1e-30; p->primal_error_threshold = 1e-30;
    p->max_complementary_gap = 1e100; p->step_length_threshold = 1e-7;
    p->need_dual_feasible = 0; p->need_primal_feasible = 0; p->safe_step = 1; p->verbose = 0;
    p->correctoronly = 0;
}

/* history row: iter, mu, d_obj, p_obj, gap, P-error, p-error, d-error, alpha_d, alpha_p, beta_c */
#define HIST_COLS 11

/* solvesdp (src/solver.jl:100-744).  Returns error_code (0 ok, 1 solver failure, 2 max iterations,
 * 3 mu too large, 4 step too short).  out[0..5] = d_obj, p_obj, gap, dual_error, primal_error, pd_feas */
int oracle_solvesdp(octx *o, const oracle_params *prm, int *iters_out, double *out, double *hist, int hist_rows,
                    double *x_out, double *y_out, double *X_out, double *Y_out) {
    i64 nxy = o->xylen, nx = o->xlen;
    int N = o->N;
    REAL *x = ralloc(nx), *y = ralloc(N), *X = ralloc(nxy), *Y = ralloc(nxy);
    REAL *R = ralloc(nxy), *Xc = ralloc(nxy), *Pm = ralloc(nxy), *dX = ralloc(nxy), *dY = ralloc(nxy), *tmp = ralloc(nxy);
    REAL *d = ralloc(nx), *pv = ralloc(N), *dx = ralloc(nx), *dy = ralloc(N), *rhsx = ralloc(nx), *tr = ralloc(nx);
    REAL gamma = (REAL)prm->gamma;
    i64 K = 0;
    for (int b = 0; b < o->NB; b++) {                                             /* :187-201 */
        oblock *k = &o->blk[b];
        K += k->n;
        for (int i = 0; i < k->n; i++) { X[k->off + i + (i64)i * k->n] = (REAL)prm->omega_p; Y[k->off + i + (i64)i * k->n] = (REAL)prm->omega_d; }
    }
    if (o->start_k > 0) {                                                         /* :202-239 */
        for (i64 i = 0; i < nx; i++) x[i] = ldk(o->start_x, nx, o->start_k, i);
        for (int i = 0; i < N; i++) y[i] = ldk(o->start_y, N, o->start_k, i);
        for (i64 i = 0; i < nxy; i++) { X[i] = ldk(o->start_X, nxy, o->start_k, i); Y[i] = ldk(o->start_Y, nxy, o->start_k, i); }
    }
    int iter = 1, error_code = 0, pd_feas = 0;
    REAL d_obj, p_obj, gap, dual_error, primal_error, mu = 0, alpha_p = 0, alpha_d = 0, beta_c = 0;
    REAL sgn = o->maximize ? 1 : -1;

#define OBJECTIVES() do { \
        d_obj = sgn * dotn(nx, o->c, x) + o->constant;                 /* :793-799 */ \
        p_obj = bdot(o, o->C, Y) + dotn(N, o->b, y) + o->constant;     /* :802-804 */ \
        REAL den_ = RABS(d_obj + p_obj); if (den_ < 1) den_ = 1; \
        gap = RABS(d_obj - p_obj) / den_;                              /* :844-847 */ \
    } while (0)
#define RESIDUALS(use_AY) do { \
        weighted_A(o, x, Pm);                                          /* :884 */ \
        for (i64 i_ = 0; i_ < nxy; i_++) Pm[i_] = Pm[i_] - X[i_] - sgn * o->C[i_];   /* :886-891 */ \
        if (use_AY) trace_A_from_AY(o, Y, tr); else trace_A(o, Y, tr); \
        for (int j_ = 0; j_ < o->J; j_++) {                            /* d = c - <A,Y> - By :863-879 */ \
            int P_ = o->P[j_]; \
            for (int r_ = 0; r_ < P_; r_++) { \
                REAL s_ = o->c[o->coff[j_] + r_] - tr[o->coff[j_] + r_]; \
                for (int i_ = 0; i_ < N; i_++) s_ -= o->B[j_][r_ + (i64)i_ * P_] * y[i_]; \
                d[o->coff[j_] + r_] = s_; } } \
        for (int i_ = 0; i_ < N; i_++) {                               /* p = +-b - B^T x :899-916 */ \
            REAL s_ = sgn * o->b[i_]; \
            for (int j_ = 0; j_ < o->J; j_++) s_ -= dotn(o->P[j_], o->B[j_] + (i64)i_ * o->P[j_], x + o->coff[j_]); \
            pv[i_] = s_; } \
        { REAL e1_ = maxabs(N, pv), e2_ = maxabs(nxy, Pm); dual_error = e1_ > e2_ ? e1_ : e2_; }  /* :828-832 */ \
        primal_error = maxabs(nx, d); \
    } while (0)

    OBJECTIVES();
    RESIDUALS(0);
    pd_feas = (dual_error < (REAL)prm->dual_error_threshold) && (primal_error < (REAL)prm->primal_error_threshold);
    while (1) {
        /* terminate (:921-950) */
        int dual_feas = dual_error < (REAL)prm->dual_error_threshold, primal_feas = primal_error < (REAL)prm->primal_error_threshold;
        if (prm->need_dual_feasible && dual_feas) break;
        if (prm->need_primal_feasible && primal_feas) break;
        if (!prm->correctoronly && dual_feas && primal_feas && gap < (REAL)prm->duality_gap_threshold) break;      /* :945 */
        if (iter > prm->maxiterations) { error_code = 2; break; }                  /* :362-366 */
        mu = bdot(o, X, Y) / (REAL)K;                                             /* :369 */
        REAL mu_p = prm->correctoronly ? mu : pd_feas ? (REAL)0 : (REAL)prm->beta_infeasible * mu;          /* :370-374 */
        if (mu > (REAL)prm->max_complementary_gap) { error_code = 3; break; }     /* :376-380 */
        /* R = mu_p I - X Y  (:961-970) */
        for (int b = 0; b < o->NB; b++) {
            oblock *k = &o->blk[b];
            gemm_nn(k->n, k->n, k->n, X + k->off, k->n, Y + k->off, k->n, R + k->off, k->n);
            for (i64 i = 0; i < (i64)k->n * k->n; i++) R[k->off + i] = -R[k->off + i];
            for (int i = 0; i < k->n; i++) R[k->off + i + (i64)i * k->n] += mu_p;
        }
        /* Cholesky of X (:388-399) */
        memcpy(Xc, X, sizeof(REAL) * nxy);
        int fail = 0;
        for (int b = 0; b < o->NB && !fail; b++) if (!cholesky_lower(o->blk[b].n, Xc + o->blk[b].off, o->blk[b].n)) fail = 1;
        if (fail) { error_code = 1; break; }
        /* decomposition (:406-408) */
        schur_assemble_real(o, Xc, Y);
        if (schur_factor_real(o) != 0) { error_code = 1; break; }
        RESIDUALS(1);                                                              /* :415 */
        REAL xy = bdot(o, X, Y);
        for (int pass = 0; pass < 2; pass++) {
            if (pass == 1) {
                /* corrector (:429-447) */
                REAL r = (xy + bdot(o, X, dY) + bdot(o, dX, Y) + bdot(o, dX, dY)) / (mu * (REAL)K);
                REAL beta = r < 1 ? r * r : r;
                if (pd_feas) { beta_c = beta > (REAL)prm->beta_feasible ? beta : (REAL)prm->beta_feasible; if (beta_c > 1) beta_c = 1; }
                else beta_c = beta > (REAL)prm->beta_infeasible ? beta : (REAL)prm->beta_infeasible;
                REAL mu_c = beta_c * mu;
                for (int b = 0; b < o->NB; b++) {                                 /* R = mu_c I - XY - dX dY :972-983 */
                    oblock *k = &o->blk[b];
                    gemm_nn(k->n, k->n, k->n, X + k->off, k->n, Y + k->off, k->n, R + k->off, k->n);
                    gemm_nn(k->n, k->n, k->n, dX + k->off, k->n, dY + k->off, k->n, tmp + k->off, k->n);
                    for (i64 i = 0; i < (i64)k->n * k->n; i++) R[k->off + i] = -R[k->off + i] - tmp[k->off + i];
                    for (int i = 0; i < k->n; i++) R[k->off + i + (i64)i * k->n] += mu_c;
                }
                pd_feas = (dual_error < (REAL)prm->dual_error_threshold) && (primal_error < (REAL)prm->primal_error_threshold);
            }
            /* compute_search_direction! (:1474-1616) */
            for (int b = 0; b < o->NB; b++) {                                     /* Z = sym(X^-1 (P Y - R)) :1501-1514 */
                oblock *k = &o->blk[b];
                int n = k->n;
                gemm_nn(n, n, n, Pm + k->off, n, Y + k->off, n, dY + k->off, n);
                for (i64 i = 0; i < (i64)n * n; i++) dY[k->off + i] -= R[k->off + i];
                potrs(n, n, Xc + k->off, n, dY + k->off, n);
                for (int c = 0; c < n; c++)
                    for (int r = c; r < n; r++) { REAL v = (dY[k->off + r + (i64)c * n] + dY[k->off + c + (i64)r * n]) / 2; dY[k->off + r + (i64)c * n] = dY[k->off + c + (i64)r * n] = v; }
            }
            trace_A(o, dY, tr);
            for (i64 i = 0; i < nx; i++) rhsx[i] = -d[i] - tr[i];                 /* :1522-1523 */
            if (pass == 0 && o->snap_count < o->snap_n && o->snap_it[o->snap_count] == iter) {
                int k_ = o->snap_k, s_ = o->snap_count++;
                for (i64 i = 0; i < nxy; i++) { stk(o->snap_X + (i64)s_ * k_ * nxy, nxy, k_, i, X[i]); stk(o->snap_Y + (i64)s_ * k_ * nxy, nxy, k_, i, Y[i]); }
                for (i64 i = 0; i < nx; i++) stk(o->snap_rx + (i64)s_ * k_ * nx, nx, k_, i, rhsx[i]);
                for (int i = 0; i < N; i++) stk(o->snap_ry + (i64)s_ * k_ * (N > 0 ? N : 1), N, k_, i, pv[i]);
            }
            schur_solve_real(o, rhsx, pv, dx, dy);                                /* :1527-1582 */
            weighted_A(o, dx, dX);                                                /* :1588 */
            for (i64 i = 0; i < nxy; i++) dX[i] += Pm[i];                         /* :1591 */
            for (int b = 0; b < o->NB; b++) {                                     /* dY = sym(X^-1 (R - dX Y)) :1598-1612 */
                oblock *k = &o->blk[b];
                int n = k->n;
                gemm_nn(n, n, n, dX + k->off, n, Y + k->off, n, dY + k->off, n);
                for (i64 i = 0; i < (i64)n * n; i++) dY[k->off + i] = R[k->off + i] - dY[k->off + i];
                potrs(n, n, Xc + k->off, n, dY + k->off, n);
                for (int c = 0; c < n; c++)
                    for (int r = c; r < n; r++) { REAL v = (dY[k->off + r + (i64)c * n] + dY[k->off + c + (i64)r * n]) / 2; dY[k->off + r + (i64)c * n] = dY[k->off + c + (i64)r * n] = v; }
            }
        }
        alpha_d = step_length(o, X, dX, gamma, pd_feas && !prm->safe_step);       /* :462-463 */
        alpha_p = step_length(o, Y, dY, gamma, pd_feas && !prm->safe_step);
        if (alpha_d < 0 || alpha_p < 0) { error_code = 1; break; }
        if (hist && iter <= hist_rows) {
            double *h = hist + (i64)(iter - 1) * HIST_COLS;
            h[0] = iter; h[1] = (double)mu; h[2] = (double)d_obj; h[3] = (double)p_obj; h[4] = (double)gap;
            h[5] = (double)maxabs(nxy, Pm); h[6] = (double)maxabs(N, pv); h[7] = (double)maxabs(nx, d);
            h[8] = (double)alpha_d; h[9] = (double)alpha_p; h[10] = (double)beta_c;
        }
        if (prm->verbose)
            printf("%5d %11.3e %11.3e %11.3e %10.2e %10.2e %10.2e %10.2e %10.2e %10.2e %10.2e\n", iter, (double)mu, (double)d_obj,
                   (double)p_obj, (double)gap, (double)maxabs(nxy, Pm), (double)maxabs(N, pv), (double)maxabs(nx, d), (double)alpha_d,
                   (double)alpha_p, (double)beta_c);
        REAL amin = alpha_d < alpha_p ? alpha_d : alpha_p;
        if (amin < (REAL)prm->step_length_threshold) { error_code = 4; break; }   /* :470-475 */
        if (pd_feas && prm->safe_step) { alpha_p = amin; alpha_d = amin; }        /* :480-483 */
        for (i64 i = 0; i < nx; i++) x[i] += alpha_d * dx[i];                     /* :485-495 */
        for (int i = 0; i < N; i++) y[i] += alpha_p * dy[i];
        for (i64 i = 0; i < nxy; i++) { X[i] += alpha_d * dX[i]; Y[i] += alpha_p * dY[i]; }
        OBJECTIVES();                                                             /* :586-588 */
        iter++;
    }
    OBJECTIVES();
    *iters_out = iter - 1;
    out[0] = (double)d_obj; out[1] = (double)p_obj; out[2] = (double)gap; out[3] = (double)dual_error;
    out[4] = (double)primal_error; out[5] = pd_feas;
    o->last_obj[0] = d_obj; o->last_obj[1] = p_obj; o->last_obj[2] = gap;
    if (x_out) for (i64 i = 0; i < nx; i++) x_out[i] = (double)x[i];
    if (y_out) for (int i = 0; i < N; i++) y_out[i] = (double)y[i];
    if (X_out) for (i64 i = 0; i < nxy; i++) X_out[i] = (double)X[i];
    if (Y_out) for (i64 i = 0; i < nxy; i++) Y_out[i] = (double)Y[i];
    free(x); free(y); free(X); free(Y); free(R); free(Xc); free(Pm); free(dX); free(dY); free(tmp);
    free(d); free(pv); free(dx); free(dy); free(rhsx); free(tr);
    return error_code;
}

/* one scalar operation of the oracle's arithmetic on k-limb operands (unit tests of REAL itself, tests/test_oracle_cpu.py):
 * op 0 add, 1 sub, 2 mul, 3 div, 4 sqrt(a) */
void oracle_real_op(int op, int k, i64 n, const double *a, const double *b, double *out) {
    for (i64 i = 0; i < n; i++) {
        REAL x = ldk(a, n, k, i), y = ldk(b, n, k, i), r;
        switch (op) {
        case 0: r = x + y; break;
        case 1: r = x - y; break;
        case 2: r = x * y; break;
        case 3: r = x / y; break;
        default: r = RSQRT(x); break;
        }
        stk(out, n, k, i, r);
    }
}

#if defined(ORACLE_MP)
int oracle_real_bits(void) { return mpx_prec_bits > 0 && mpx_prec_bits < 64 * ORACLE_MP ? mpx_prec_bits : 64 * ORACLE_MP; }
/* working precision of every later operation (1..64*limbs bits; 0 = all limbs): the reference's `prec` */
void oracle_set_precision_bits(int bits) { mpx_prec_bits = bits; }
#elif defined(ORACLE_QUAD)
int oracle_real_bits(void) { return 113; }
void oracle_set_precision_bits(int bits) { (void)bits; }
#else
int oracle_real_bits(void) { return 53; }
void oracle_set_precision_bits(int bits) { (void)bits; }
#endif
/* the reference's matmul_prec (bits; 0 = the working precision); effective in the multi-limb builds */
void oracle_set_matmul_precision_bits(int bits) { oracle_matmul_bits_ = bits; }
void oracle_set_num_threads(int n) {
    if (n > 0) omp_set_num_threads(n);
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
