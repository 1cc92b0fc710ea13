// clrs_mw_ipm.hip.h -- the interior-point iteration around the path at the reference's working precision (included by
// clrs_mw.hip).  SURVEY.md section 8f rows 1-2 in multi-word fp64: x, y, X, Y and every intermediate stay in HBM as planar
// limbs; one iteration is the loop body of solvesdp (src/solver.jl:348-589):
//   mu (:369) -> R = mu_p I - X Y (:961-970) -> chol X (:388-399) -> decomposition (:406-408, the path) -> residuals P, p, d
//   (:863-918) -> predictor (:423) -> beta_c, mu_c (:429-434) -> R (:972-983) -> corrector (:454) -> step lengths (:462-463,
//   1620-1693) -> update (:485-495) -> objectives (:793-804, 844-847).
// The scalar control flow runs in one-thread kernels between the stages; the host reads one record per iteration and decides
// termination (src/solver.jl:921-950).  The smallest eigenvalue of L^-1 dM L^-T is taken in fp64 (Householder + Sturm
// multisection) from the multi-word congruence rounded to fp64 -- the reference, too, hands a Float64 matrix to its Lanczos
// (KrylovKit, tol 1e-5, src/solver.jl:1659) -- everything else carries K limbs.

enum { MSC_MU = 0, MSC_MUS, MSC_XY, MSC_XdY, MSC_dXY, MSC_dXdY, MSC_CY, MSC_DOBJ, MSC_POBJ, MSC_GAP, MSC_COUNT = 16 };
enum { MREC_ITER = 0, MREC_MU, MREC_DOBJ, MREC_POBJ, MREC_GAP, MREC_DERR, MREC_PERR, MREC_AD, MREC_AP, MREC_BETA, MREC_MAXP, MREC_MAXp, MREC_MAXd,
       MREC_PDFEAS, MREC_ERR, MREC_FSTAT, MREC_XSTAT, MREC_REFB, MREC_COUNT = 24 };

struct MwIpmDev {
    double *x, *y, *X, *Y, *dx, *dy, *dX, *dY, *R, *Xc, *Pm, *d, *rhsx, *pv, *coef;   // planar limbs
    double *Yi;                        // chol(Y)^-1 per block (xy layout), formed beside chol(X) at the start of the iteration
    int *yfail;                        // [NB] 1 = Y_b is not positive definite
    int wBn;                           // leading dimension of wB: the largest side of a low-rank block (0: not used)
    double *Zs;                        // unsymmetrised X^-1 (...) of k_mwi_Zi (xy layout)
    double *Zt;                        // second scratch of the tiled form of the same products (k_mwi_bmm; xy layout)
    double *wB;                        // [T * wBn] K limbs: coefficient times right vector of every term, for blocks with many terms (k_mwi_wB)
    int *zcnt;                         // [NB] workgroups of a block that have delivered their panel
    unsigned long long *stamps;        // diagnostic (clrs_mw_debug_exact_stamps): wall_clock64 at the phase boundaries of k_mwi_Zi, wave 0 of the first workgroup, or null
    double *dtr;                       // [xlen] dense part of the row traces <A_*, M> (k_mwi_rows_dn), when some dense block has n > 1
    double *sc, *part;                 // planar scalars [MSC_COUNT]; partial dot products [5][NB]
    unsigned long long *fmax;          // bit patterns of non-negative doubles: [0] max|P|, [1] max|d|, [2] max|p|
    double *eig;                       // fp64 [2][NB]: smallest eigenvalue of L^-1 dM L^-T per block (X then Y)
    double *Wd;                        // fp64 [2][xylen]: the congruences L^-1 dM L^-T rounded to fp64, written by column panels (k_mwi_step)
    int *wcnt;                         // [2 NB] workgroups of a (block, which) that have delivered their panel
    double *rec;                       // fp64 record of the iteration
    int *flags;                        // [0] pd_feas, [1] error_code, [2] Cholesky failure inside the step length
    int klow, pad4;                    // limbs of the arithmetic of THIS launch of k_mwi_wA / k_mwi_Zi / k_mwi_dots: K, or (the predictor's dX, dY and the dot products
                                       // that lead to beta_c, when the factor stage runs reduced: clrs_mw_ipm_host.inc) mw_kf_of(K) -- the predictor's (dx, dy) come from
                                       // factors of that many limbs, unrefined, and feed beta_c and the second-order term of the corrector only
    double *tau, *ttau, *utau;         // the corrector's right-hand side is affine in mu_c (mw_ipm_enqueue): tau_g = <A_g, X^-1> (xlen; k_mwi_rows mode 0 forms it beside d),
                                       // t_tau = Si tau (xlen), u_tau = LB^T t_tau (J x N slabs; both ride on the Cholesky of Q); null: the corrector waits for mu_c
    int *sync;                         // [3] "the dot products behind the predictor are complete" (stored by the corrector's first launch), [4] "mu_c is there" (side stream); [0] the number of the last iteration whose side-stream work (everything the predictor's solve reads) is complete (k_mwi_mark)
    const double *C, *c, *b;           // problem data, DK limbs planar (sdp.C xy layout, sdp.c x layout, sdp.b [N])
    const int *row_clu;                // [xlen] cluster of each constraint row
    double sgn, constant;
    int Ktot, pad;
    double beta_infeasible, beta_feasible, gamma, dual_thr, primal_thr, max_gap, step_thr;
    int safe_step, corrector_only;     // (corrector_only: the reference's correctoronly keyword, src/solver.jl:370-374, 945)
    // termination test of the loop (src/solver.jl:921-950) evaluated ON THE DEVICE at the end of an iteration when stop_on != 0
    // (clrs_mw_ipm_solve: the host runs one iteration ahead of the records it reads; an iteration that follows the last one must
    // not move the iterate): flags[6] = 1 makes k_mwi_update a no-op
    double gap_thr;
    int stop_on, need_dual, need_primal, pad3;
    // cluster sharding (q.world > 1): the sums / maxima / minima over ALL clusters that the scalar stages need (mu :369, errors :441-442,
    // p = +-b - B^T x :899-916, beta_c :429, step lengths :1684-1686, objectives :793-804) travel as one small record per rank and stage:
    // k_mwi_gpack writes this rank's slot, the library all-gathers it, every rank reduces the slots in rank order -- identical bits everywhere.
    // Two buffers, one per stream that exchanges.  Slot layout: see MWG_* below.
    double *gsM, *gsS;                 // [world][GL]
    int GL, Jglob;                     // slot length; clusters of the whole problem (status code of a failed Q: Jglob + 1)
    const int *clu_gid, *blk_gid;      // global number of every local cluster / PSD block (failure codes are global), or null = local numbers
};
#define MWG_S1(K, N) 0                       /* K limbs: <X,Y> (stage 0) | <X,dY>+<dX,Y>+<dX,dY> (stage 2) | <C,Y> (stage 4) */
#define MWG_S2(K, N) (K)                     /* K limbs: <c,x> (stage 4) */
#define MWG_BX(K, N) (2 * (K))               /* K N limbs, planar with plane N: -B^T x of this rank's rows (stage 1) */
#define MWG_D(K, N) (2 * (K) + (K) * (N))    /* doubles: max|P|, max|d|, min eig X, min eig Y, factor status, Cholesky status, step-length failure, - */
#define MWG_XY(K, N) (MWG_D(K, N) + 8)        /* K limbs: <X,Y> of the iterate the last update produced (stage 15) */
#define MWG_LEN(K, N) (MWG_D(K, N) + 8 + (K))

namespace mwk {

// The scalar stages are control flow between a handful of K-limb operations; inlined, every division is thousands of instructions
// and a stage function grows to hundreds of kilobytes (585 KB at 10 limbs, where the device then hangs in it).  Their arithmetic
// goes through calls instead: one copy of each operation per K.
template <int K> __device__ __noinline__ mw<K> s_div(const mw<K> a, const mw<K> b) { return div<K>(a, b); }
template <int K> __device__ __noinline__ mw<K> s_mul(const mw<K> a, const mw<K> b) { return mul<K>(a, b); }
template <int K> __device__ __noinline__ mw<K> s_mul_d(const mw<K> a, double b) { return mul_d<K>(a, b); }
template <int K> __device__ __noinline__ mw<K> s_add(const mw<K> a, const mw<K> b) { return add<K>(a, b); }
template <int K> __device__ __noinline__ mw<K> s_sub(const mw<K> a, const mw<K> b) { return sub<K>(a, b); }
template <int K> __device__ __noinline__ mw<K> s_result(const acc<K> s) { return acc_result<K>(s); }
template <int K> __device__ __noinline__ mw<K> s_lanes64(const mw<K> v) { return lanes_sum<K, 64>(v); }
template <int K> __device__ __forceinline__ bool s_less(const mw<K> &a, const mw<K> &b) { return s_sub<K>(a, b).l[0] < 0.0; }

// sum over the workgroup (256 threads) of one multi-word value per thread; red: LDS, K * 256 doubles; result in every thread
template <int K>
__device__ __noinline__ mw<K> wg_reduce_sum(const mw<K> &v, lds_d *red, int tid) {
    mw<K> w = lanes_sum<K, 64>(v);                     // within the wave by shuffles, across the four waves through LDS
    if ((tid & 63) == 0) stx<K>(red, MW_NT / 64, tid >> 6, w);
    __syncthreads();
    acc<K> s;
    acc_zero<K>(s);
#pragma unroll
    for (int i = 0; i < MW_NT / 64; i++) acc_add<K, K>(s, ldx<K>(red, MW_NT / 64, i));
    __syncthreads();
    return acc_result<K>(s);
}
__device__ __forceinline__ void atomic_max_abs(unsigned long long *slot, double v) {
    atomicMax(slot, (unsigned long long)__double_as_longlong(__builtin_fabs(v)));
}

// sum of one double per thread over the workgroup: shuffles inside the waves, one LDS slot per wave, one barrier
__device__ __forceinline__ double wg_sum_d(double v, lds_d *red, int tid) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = 0;
#pragma unroll
    for (int w = 0; w < MW_NT / 64; w++) s += red[w];
    return s;
}

// Smallest eigenvalue of the symmetric n x n fp64 matrix A (LDS, full storage, leading dimension n; destroyed).
// Householder tridiagonalisation by the workgroup, then Sturm-count multisection (256 shifts per round).
// work: LDS, at least 3 n + 2 * MW_NT doubles.
__device__ double wg_min_eig(lds_d *A, int n, lds_d *work, int tid) {
    if (n == 1) return A[0];
    lds_d *dd = work, *ee = work + n, *v = work + 2 * n, *red = work + 3 * n, *pq = work + 3 * n + MW_NT;   // pq: n doubles inside the second MW_NT chunk
    for (int k = 0; k < n - 2; k++) {
        const int m = n - k - 1;           // x = A[k+1 .. n-1, k]
        double part = 0;
        for (int i = tid; i < m; i += MW_NT) { double t = A[(k + 1 + i) + k * n]; part += t * t; }
        const double nrm2 = wg_sum_d(part, red, tid);
        const double x0 = A[(k + 1) + k * n];
        const double alpha = (x0 > 0 ? -1.0 : 1.0) * __builtin_sqrt(nrm2);
        if (tid == 0) { dd[k] = A[k + k * n]; ee[k] = (nrm2 == 0.0) ? 0.0 : alpha; }
        if (nrm2 == 0.0 || nrm2 == x0 * x0) {            // column already tridiagonal
            if (tid == 0) ee[k] = x0;
            __syncthreads();
            continue;
        }
        for (int i = tid; i < m; i += MW_NT) v[i] = A[(k + 1 + i) + k * n] - (i == 0 ? alpha : 0.0);
        __syncthreads();
        const double vtv = nrm2 - 2.0 * alpha * x0 + alpha * alpha;   // |x - alpha e1|^2
        const double beta = 2.0 / vtv;
        // p = beta * A22 v, four lanes per row
        part = 0;
        for (int i0 = 0; i0 < m; i0 += MW_NT / 4) {
            const int i = i0 + (tid >> 2);
            double sm = 0;
            if (i < m)
                for (int j = tid & 3; j < m; j += 4) sm += A[(k + 1 + i) + (k + 1 + j) * n] * v[j];
            sm += __shfl_xor(sm, 1, 64);
            sm += __shfl_xor(sm, 2, 64);
            if (i < m && (tid & 3) == 0) { pq[i] = beta * sm; part += v[i] * beta * sm; }
        }
        const double Kc = 0.5 * beta * wg_sum_d(part, red + MW_NT / 64, tid);      // its own slots: the first reduction may still be read
        for (int e = tid; e < m * m; e += MW_NT) {
            const int i = e % m, j = e / m;
            A[(k + 1 + i) + (k + 1 + j) * n] -= v[i] * (pq[j] - Kc * v[j]) + (pq[i] - Kc * v[i]) * v[j];
        }
        __syncthreads();
    }
    if (tid == 0) {
        dd[n - 2] = A[(n - 2) + (n - 2) * n];
        ee[n - 2] = A[(n - 1) + (n - 2) * n];
        dd[n - 1] = A[(n - 1) + (n - 1) * n];
    }
    __syncthreads();
    // Gershgorin interval
    double lo = dd[0], hi = dd[0];
    for (int i = 0; i < n; i++) {
        const double r = (i > 0 ? __builtin_fabs(ee[i - 1]) : 0.0) + (i < n - 1 ? __builtin_fabs(ee[i]) : 0.0);
        lo = fmin(lo, dd[i] - r);
        hi = fmax(hi, dd[i] + r);
    }
    const double scale = fmax(__builtin_fabs(lo), __builtin_fabs(hi));
    if (scale == 0.0) return 0.0;
    lo -= 1e-14 * scale;
    hi += 1e-14 * scale;
    int *first = (int *)(double *)red;               // generic pointer: atomicMin has no overload for address-space-qualified ones
    for (int round = 0; round < 8; round++) {
        // shift of thread t: lo + (t + 1) * h; count the eigenvalues below it
        const double h = (hi - lo) / MW_NT;
        const double sg = lo + (tid + 1) * h;
        int cnt = 0;
        double qv = dd[0] - sg;
        if (qv < 0) cnt++;
        for (int i = 1; i < n; i++) {
            if (qv == 0.0) qv = 1e-300;
            qv = dd[i] - sg - ee[i - 1] * ee[i - 1] / qv;
            if (qv < 0) cnt++;
        }
        if (tid == 0) *first = MW_NT - 1;
        __syncthreads();
        if (cnt >= 1) atomicMin(first, tid);
        __syncthreads();
        const int f = *first;
        __syncthreads();
        const double nlo = lo + f * h, nhi = lo + (f + 1) * h;
        lo = nlo;
        hi = nhi;
        if (hi - lo <= 4e-16 * scale) break;
    }
    return 0.5 * (lo + hi);
}


// ---- the same for n <= 64 (wg_min_eig32: named after its first version) with the tridiagonalisation in the REGISTERS of one wave and a division-free Sturm count ------------------
// (the scheme of the fp64 loop's k_ipm_step, clrs_ipm.hip.h: lane r holds row r, the column loop is fully unrolled so that every
// register index is static, v and w reach the other lanes through v_readlane, norms and dot products through DPP row shifts; the
// Sturm sequence runs in product form p_(i+1) = (d_i - s) p_i - e_i^2 p_(i-1) with a power-of-two renormalisation every four steps,
// 257 sections per round with ONE barrier).  The LDS form above spends a barrier and strided row reads on each of its n - 2
// dependent steps and an IEEE division on each of the n steps of its Sturm chains: 45 of the 72 us of k_mwi_step at n = 16.
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
__device__ __forceinline__ double wave_sum_d(double v) {      // the same value (bitwise) in every lane
    v += dpp_mov_zero<0x111>(v);   // row_shr:1
    v += dpp_mov_zero<0x112>(v);   // row_shr:2
    v += dpp_mov_zero<0x114>(v);   // row_shr:4
    v += dpp_mov_zero<0x118>(v);   // row_shr:8   -> lane 15 of every row holds the row total
    return (readlane_f64(v, 15) + readlane_f64(v, 31)) + (readlane_f64(v, 47) + readlane_f64(v, 63));
}
template <int NC>
__device__ __forceinline__ void householder_regs(lds_d *A, int n, lds_d *dd, lds_d *ee, int lane) {
    double a[NC];
    const int r = lane < n ? lane : 0;
#pragma unroll
    for (int j = 0; j < NC; j++) a[j] = A[r * n + (j < n ? j : 0)];      // row r = column r: both triangles are stored
    const bool rowok = lane < n;
#pragma unroll
    for (int j = 0; j < NC; j++) a[j] = (rowok && j < n) ? a[j] : 0.0;
#pragma unroll
    for (int c = 0; c < NC - 2; c++) {
        if (c < n - 2) {                                    // uniform
            const double xi = (lane > c) ? a[c] : 0.0;      // column c below the diagonal (rows >= n hold zeros)
            const double ss = wave_sum_d(xi * xi);
            const double x0 = readlane_f64(xi, c + 1);
            if (ss == 0.0) {
                if (lane == 0) ee[c] = 0.0;
            } else {
                // H = I - tau v v^T is orthogonal for ANY alpha as long as tau = 2 / v^T v for the v in use: |x| and tau by rsq / rcp
                // with Newton steps (the IEEE sqrt and division are ~60 dependent instructions per step)
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
                const double kk = wave_sum_d(pi * vi);
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
// n = 33 .. 64 (a wave has 64 lanes: one row each; 96-128 VGPRs of matrix): a call, not inlined -- inlined, the register allocation of the whole
// step kernel followed this path and the n <= 32 path of the named problems paid for it (34.5 -> 37.4 us)
__device__ __noinline__ void householder_regs_big(lds_d *A, int n, lds_d *dd, lds_d *ee, int lane) {
    if (n <= 48) householder_regs<48>(A, n, dd, ee, lane);
    else householder_regs<64>(A, n, dd, ee, lane);
}
#ifdef CLRS_MW_STAMPS            // diagnostic builds only: phase stamps of the workgroup of k_mwi_step that ends the launch (scripts/step_stamps.py)
__device__ unsigned long long g_mws_stamp[256][8];
#define MWS_STAMP(i) do { if (threadIdx.x == 0) mwk::g_mws_stamp[blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)][i] = wall_clock64(); } while (0)
#else
#define MWS_STAMP(i) do { } while (0)
#endif
// work: LDS, at least 2 n + 160 doubles (3 n + 2 MW_NT are there); 2 <= n <= 64; NT threads (NT + 1 sections per round)
template <int NT = MW_NT>
__device__ double wg_min_eig32(lds_d *A, int n, lds_d *work, int tid) {
    __shared__ int zc[2][NT / 64];
    lds_d *dd = work, *ee = work + n, *d2 = work + 2 * n, *e2s = d2 + 80;
    const int wave = tid >> 6, lane = tid & 63;
    if (wave == 0) {
        if (n <= 16) householder_regs<16>(A, n, dd, ee, lane);
        else if (n <= 24) householder_regs<24>(A, n, dd, ee, lane);
        else if (n <= 32) householder_regs<32>(A, n, dd, ee, lane);
        else householder_regs_big(A, n, dd, ee, lane);
    }
    __syncthreads();
    MWS_STAMP(6);
    double lo = dd[0], hi = dd[0];                          // Gershgorin interval, by every thread
    for (int i = 0; i < n; i++) {
        const double rr = (i > 0 ? __builtin_fabs(ee[i - 1]) : 0.0) + (i < n - 1 ? __builtin_fabs(ee[i]) : 0.0);
        lo = fmin(lo, dd[i] - rr);
        hi = fmax(hi, dd[i] + rr);
    }
    const double scale = fmax(fmax(__builtin_fabs(lo), __builtin_fabs(hi)), 1e-290);
    const int sexp = __builtin_amdgcn_frexp_exp(scale);           // scale < 2^sexp: |d - s| <= 2, e^2 <= 1 after the exact scaling
    if (tid < 80) {
        d2[tid] = (tid < n) ? ldexp(dd[tid], -sexp) : 0.0;
        const double es = (tid >= 1 && tid < n) ? ldexp(ee[tid - 1], -sexp) : 0.0;
        e2s[tid] = es * es;
    }
    lo = ldexp(lo, -sexp) - 1e-3;
    hi = ldexp(hi, -sexp) + 1e-3;
    __syncthreads();
    MWS_STAMP(7);
    constexpr int ROUNDS = NT >= 1024 ? 6 : 7;      // 257^7, 1025^6 > 1e16: the bracket shrinks to rounding level
    for (int round = 0; round < ROUNDS; round++) {
        const double h = (hi - lo) * (1.0 / (NT + 1));
        const double sft = lo + h * (tid + 1);
        double pm = 1.0, pc = (d2[0] - sft) + 1e-300;     // p_0, p_1
        auto hi32 = [](double v) { return (unsigned)(__double_as_longlong(v) >> 32); };
        unsigned cnt = hi32(pc) >> 31;                     // sign changes counted on the sign bits
        auto step = [&](double tvu, double e2u) {
            // + 1e-300 off the dependent chain: an exact zero becomes a tiny positive p, from which the recurrence continues correctly
            const double pn = __builtin_fma(tvu, pc, __builtin_fma(-e2u, pm, 1e-300));
            cnt += (hi32(pn) ^ hi32(pc)) >> 31;
            pm = pc;
            pc = pn;
        };
        int i0 = 1;
        for (; i0 + 4 <= n; i0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; u++) step(d2[i0 + u] - sft, e2s[i0 + u]);
            const int ex = __builtin_amdgcn_frexp_exp(pc);
            pc = ldexp(pc, -ex);
            pm = ldexp(pm, -ex);
        }
        for (; i0 < n; i0++) step(d2[i0] - sft, e2s[i0]);
        // the counts are monotone in the shift: the number of shifts with count 0 is the index of the sub-interval with the smallest eigenvalue
        const unsigned long long zero = __ballot(cnt == 0u);
        if (lane == 0) zc[round & 1][wave] = __popcll(zero);
        __syncthreads();
        int idx = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; w++) idx += zc[round & 1][w];
        const double nlo = lo + h * idx;
        hi = (idx == NT) ? hi : lo + h * (idx + 1);
        lo = nlo;
    }
    return ldexp(0.5 * (lo + hi), sexp);
}

}  // namespace mwk

// ---- scalar stages (one thread) ------------------------------------------------------------------------------------
template <int K>
__device__ __noinline__ mwa::mw<K> mwi_sum_part(const MwDev &q, const MwIpmDev &p, int slot) {
    using namespace mwk;
    acc<K> s;
    acc_zero<K>(s);
    for (int b = 0; b < q.NB; b++) acc_add<K, K>(s, ldx<K>(p.part, 5L * q.NB, (long)slot * q.NB + b));
    return acc_result<K>(s);
}
// Two scalar stages (1: errors, 3: step lengths) run at the tail of the kernel that produces their inputs: the workgroup that
// finishes last (a counter in flags[4]; every workgroup publishes its results with a fence before it counts itself) executes
// the stage.  (The stages behind the block dot products stay separate launches: the fences cost those kernels more than the
// launch saves -- measured 31 us against 10 + 9.5.)  `mwi_last_block` is uniform over the workgroup.
__device__ __forceinline__ bool mwi_last_block(int *counter, unsigned total) { return mwk::wg_last_block(counter, total); }
// sum over the ranks, in rank order, of the K-limb number at offset `off` of every slot of a gather buffer
template <int K>
__device__ __noinline__ mwa::mw<K> mwi_gsum(const MwDev &q, const double *gs, int GL, int off) {
    using namespace mwk;
    acc<K> s;
    acc_zero<K>(s);
    for (int r = 0; r < q.world; r++) {
        mw<K> v;
#pragma unroll
        for (int l = 0; l < K; l++) v.l[l] = gs[(long)r * GL + off + l];
        acc_add<K, K>(s, v);
    }
    return acc_result<K>(s);
}
// step lengths (:1684-1691, 470-483), by one lane.  A function of its own, inlined into k_mwi_step: through the call of mwi_scalar_stage (not inlined: its other
// stages are chains of K-limb operations) the kernel's argument structures had their addresses taken, so the kernel opened with a copy of both to scratch
// (80 stores per lane) and read every field of them from there, and the stage ran behind a call with a stack frame -- 12 of the 45 us of k_mwi_step.
template <int K>
__device__ __forceinline__ void mwi_scalar_stage3(const MwDev &q, const MwIpmDev &p) {
    using namespace mwk;
    // this stage ends k_mwi_step, on the critical path of the iteration: everything it reads is asked for at once, in front of the arithmetic and
    // of the first store (one round trip to memory instead of one per dependent group of loads; 10 -> ~4 us)
    const bool pre = q.world <= 1 && q.NB <= 8;
    double pev[2][8];
#pragma unroll
    for (int w = 0; w < 2; w++)
#pragma unroll
        for (int b = 0; b < 8; b++) pev[w][b] = (pre && b < q.NB) ? p.eig[(long)w * q.NB + b] : 1e300;
    const int f0 = p.flags[0];
    int f1 = p.flags[1], f2 = p.flags[2];
    unsigned long long rs[4] = {0ull, 0ull, 0ull, 0ull};
    if (q.refstat) {
        if (q.world > 1) {                             // the largest over the ranks (k_mwi_gpack stage 3): the same number on every rank
#pragma unroll
            for (int i = 0; i < 4; i++) {
                double m = 0.0;
                for (int r = 0; r < q.world; r++) m = fmax(m, p.gsM[(long)r * p.GL + MWG_S1(K, q.N) + i]);
                rs[i] = (unsigned long long)__double_as_longlong(m);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) rs[i] = q.refstat[i];
        }
    }
    double al[2];
    for (int w = 0; w < 2; w++) {
        double mn;
        if (q.world > 1) {
            mn = p.gsM[MWG_D(K, q.N) + 2 + w];
            for (int r = 1; r < q.world; r++) mn = fmin(mn, p.gsM[(long)r * p.GL + MWG_D(K, q.N) + 2 + w]);
            for (int r = 0; r < q.world; r++) if (p.gsM[(long)r * p.GL + MWG_D(K, q.N) + 6] != 0.0) f2 = 1;
        } else if (pre) {
            mn = pev[w][0];
#pragma unroll
            for (int b = 1; b < 8; b++) mn = fmin(mn, pev[w][b]);
        } else {
            mn = p.eig[(long)w * q.NB];
            for (int b = 1; b < q.NB; b++) mn = fmin(mn, p.eig[(long)w * q.NB + b]);
        }
        const bool unsafe = f0 && !p.safe_step;
        al[w] = (mn > -p.gamma && !unsafe) ? 1.0 : -p.gamma / mn;
    }
    if (f2 && f1 == 0) f1 = 1;
    double amin = fmin(al[0], al[1]);
    if (!(amin >= p.step_thr) && f1 == 0) f1 = 4;
    // how good the first pass of the corrector's refined solve was: -log2(max|correction| / max|solution|) over dx and dy (k_mw_solve_bwd MODE 2 left the
    // four maxima in q.refstat).  The host returns the factor stage to all K limbs when this falls to one limb plus a margin (mw_kf_of, clrs_mw_ipm_host.inc).
    double refb = 0.0;
    if (q.refstat) {
        const double c0 = __longlong_as_double((long long)rs[0]), v0 = __longlong_as_double((long long)rs[1]);
        const double c1 = __longlong_as_double((long long)rs[2]), v1 = __longlong_as_double((long long)rs[3]);
        double ratio = 0.0;
        if (v0 > 0.0) ratio = fmax(ratio, c0 / v0);
        if (v1 > 0.0) ratio = fmax(ratio, c1 / v1);
        refb = (v0 > 0.0 || v1 > 0.0) ? (ratio > 0.0 ? fmin(1023.0, fmax(1.0, -log2(ratio))) : 1023.0) : 0.0;
    }
    p.flags[2] = f2;
    p.flags[1] = f1;
    p.rec[MREC_AD] = al[0]; p.rec[MREC_AP] = al[1];           // the table row shows the values before they are equalised
    if (f0 && p.safe_step) al[0] = al[1] = amin;
    p.sc[MSC_COUNT * 0 + 10] = al[0];                          // slots 10 / 11 of limb plane 0: alpha_d / alpha_p as plain doubles
    p.sc[MSC_COUNT * 0 + 11] = al[1];
    p.rec[MREC_ERR] = f1;
    if (q.refstat) {
        p.rec[MREC_REFB] = refb;
        q.refstat[0] = q.refstat[1] = q.refstat[2] = q.refstat[3] = 0ull;
    }
}
// called by the first wave of a workgroup (threadIdx.x < 64); stages 0-3 use its first lane only
template <int K, int DK>
__device__ void mwi_scalar_stage(const MwDev &q, const MwIpmDev &p, int stage, int iter) {
    using namespace mwk;
    const long SP = MSC_COUNT;
    if (stage == 4) {                              // objectives of the new iterate (:793-804, 844-847): the two dot products over one wave
        const int lane = threadIdx.x;
        acc<K> s;
        acc_zero<K>(s);
        for (long i = lane; i < q.xlen; i += 64) acc_fma<K, K, DK>(s, ldx<K>(p.x, q.xlen, i), ldx<DK>(p.c, q.xlen, i), p.sgn);
        mw<K> cx = s_lanes64<K>(s_result<K>(s));
        acc_zero<K>(s);
        for (int a = lane; a < q.N; a += 64) acc_fma<K, K, DK>(s, ldx<K>(p.y, q.N, a), ldx<DK>(p.b, q.N, a));
        mw<K> by = s_lanes64<K>(s_result<K>(s));
        if (lane != 0) return;
        if (q.world > 1) cx = mwi_gsum<K>(q, p.gsS, p.GL, MWG_S2(K, q.N));       // this rank's rows of <c,x> were gathered (k_mwi_gpack)
        mw<K> dobj = s_add<K>(cx, from_double<K>(p.constant));
        acc_zero<K>(s);
        acc_add<K, K>(s, q.world > 1 ? mwi_gsum<K>(q, p.gsS, p.GL, MWG_S1(K, q.N)) : mwi_sum_part<K>(q, p, 4));
        acc_add<K, K>(s, by);
        acc_add_d<K>(s, p.constant);
        mw<K> pobj = s_result<K>(s);
        mw<K> den = abs<K>(s_add<K>(dobj, pobj));
        if (s_less<K>(den, from_double<K>(1.0))) den = from_double<K>(1.0);
        mw<K> gap = s_div<K>(abs<K>(s_sub<K>(dobj, pobj)), den);
        stx<K>(p.sc, SP, MSC_DOBJ, dobj); stx<K>(p.sc, SP, MSC_POBJ, pobj); stx<K>(p.sc, SP, MSC_GAP, gap);
        p.rec[MREC_DOBJ] = dobj.l[0]; p.rec[MREC_POBJ] = pobj.l[0]; p.rec[MREC_GAP] = gap.l[0];
        p.rec[MREC_ERR] = p.flags[1];
        p.rec[MREC_PDFEAS] = p.flags[0];
        if (p.stop_on) {                           // the test of :921-950 on the values the host will read from this record
            const bool df = p.rec[MREC_DERR] < p.dual_thr, pf = p.rec[MREC_PERR] < p.primal_thr;
            if (p.flags[1] != 0 || (p.need_dual && df) || (p.need_primal && pf) || (!p.corrector_only && df && pf && gap.l[0] < p.gap_thr)) p.flags[6] = 1;      // (:945)
        }
        return;
    }
    if (threadIdx.x != 0) return;
    if (stage == 5) {                              // warm start (clrs_mw_ipm_set): the feasibility of the STARTING iterate, which the reference knows before its
        const double maxP = __longlong_as_double((long long)p.fmax[0]), maxd = __longlong_as_double((long long)p.fmax[1]),      // loop (src/solver.jl:322-333)
                     maxp = __longlong_as_double((long long)p.fmax[2]);      // and which decides mu_p and beta_c of the first iteration (:373, :429-434)
        p.flags[0] = (q.info[1] == MW_INFO_NONE && fmax(maxp, maxP) < p.dual_thr && maxd < p.primal_thr) ? 1 : 0;
        p.fmax[0] = p.fmax[1] = p.fmax[2] = 0ull;
        q.info[0] = MW_INFO_NONE;
        q.info[1] = MW_INFO_NONE;
        return;
    }
    const bool xy_merged = stage == 10;            // stage 0 whose <X,Y> travelled with the previous iteration's objectives (k_mwi_gpack stage 15: MWG_XY)
    if (xy_merged) stage = 0;
    if (stage == 0) {                              // start of the iteration: mu, mu_p  (src/solver.jl:369-380)
        mw<K> xy = q.world > 1 ? mwi_gsum<K>(q, p.gsS, p.GL, xy_merged ? MWG_XY(K, q.N) : MWG_S1(K, q.N)) : mwi_sum_part<K>(q, p, 0);
        mw<K> mu = s_div<K>(xy, from_double<K>((double)p.Ktot));
        stx<K>(p.sc, SP, MSC_XY, xy);
        stx<K>(p.sc, SP, MSC_MU, mu);
        stx<K>(p.sc, SP, MSC_MUS, p.corrector_only ? mu : p.flags[0] ? zero<K>() : s_mul_d<K>(mu, p.beta_infeasible));      // mu_p (:370-374)
        p.flags[1] = 0;
        p.flags[2] = 0;
        p.fmax[0] = p.fmax[1] = p.fmax[2] = 0ull;
        for (int i = 0; i < MREC_COUNT; i++) p.rec[i] = 0.0;
        p.rec[MREC_ITER] = iter;
        p.rec[MREC_MU] = mu.l[0];
        if (mu.l[0] > p.max_gap) p.flags[1] = 3;
    }
    // stage 2 sits on the main stream between predictor and corrector: everything it reads is asked for at once, here, in front of the (non-inlined) K-limb
    // operations -- one round trip to memory instead of one per dependent group of loads (13.5 -> ~9 us)
    mw<K> pre_xy = zero<K>(), pre_mu = zero<K>(), pre_sum = zero<K>();
    int pre_flag0 = 0;
    if (stage == 2) {
        pre_xy = ldx<K>(p.sc, SP, MSC_XY);
        pre_mu = ldx<K>(p.sc, SP, MSC_MU);
        pre_flag0 = p.flags[0];
        if (q.world <= 1 && q.NB <= 8) {
            mw<K> part[8];
#pragma unroll
            for (int b = 0; b < 8; b++) part[b] = b < q.NB ? ldx<K>(p.part, 5L * q.NB, 1L * q.NB + b) : zero<K>();
            acc<K> s;
            acc_zero<K>(s);
#pragma unroll
            for (int b = 0; b < 8; b++) if (b < q.NB) acc_add<K, K>(s, part[b]);
            pre_sum = acc_result<K>(s);
        }
    }
    if (stage == 1 || stage == 2) {                // after the residuals: errors (:441-447 use them), failures of the decomposition
        const double maxP = __longlong_as_double((long long)p.fmax[0]), maxp = __longlong_as_double((long long)p.fmax[2]);
        double maxd = __longlong_as_double((long long)p.fmax[1]);
        if (q.world > 1 && stage == 2) {           // max|d| over the ranks (their stage-2 records)
            for (int r = 0; r < q.world; r++) maxd = fmax(maxd, p.gsM[(long)r * p.GL + MWG_D(K, q.N) + 1]);
        }
        // max|P| of the NEXT iterate is accumulated by the tail of this iteration when the solve is sharded (stage 15), before the next stage 0 runs
        p.fmax[0] = 0ull;
        p.rec[MREC_MAXP] = maxP; p.rec[MREC_MAXd] = maxd; p.rec[MREC_MAXp] = maxp;
        p.rec[MREC_DERR] = fmax(maxp, maxP);       // :828-832
        p.rec[MREC_PERR] = maxd;
        int fs = q.info[0], xs = q.info[1];
        if (q.world > 1) {                         // the first failure in rank order, as every rank sees it (codes are global numbers)
            fs = xs = MW_INFO_NONE;
            for (int r = q.world - 1; r >= 0; r--) {
                const double *D = p.gsM + (long)r * p.GL + MWG_D(K, q.N);
                if ((int)D[4] != MW_INFO_NONE) fs = (int)D[4];
                if ((int)D[5] != MW_INFO_NONE) xs = (int)D[5];
            }
        }
        p.rec[MREC_FSTAT] = fs == MW_INFO_NONE ? 0 : fs;
        p.rec[MREC_XSTAT] = xs == MW_INFO_NONE ? 0 : xs;
        if ((fs != MW_INFO_NONE || xs != MW_INFO_NONE) && p.flags[1] == 0) p.flags[1] = (fs == MW_INFO_TIMEOUT || xs == MW_INFO_TIMEOUT) ? 5 : 1;      // 5: a wait between the two streams timed out
        q.info[0] = MW_INFO_NONE;                  // re-armed for the next decomposition (the iteration issues no memsets)
        q.info[1] = MW_INFO_NONE;
    }
    if (stage == 2) {                              // between predictor and corrector: beta_c, mu_c (:429-434), then pd_feas (:441-447)
        acc<K> s;
        acc_zero<K>(s);
        acc_add<K, K>(s, pre_xy);
        acc_add<K, K>(s, q.world > 1 ? mwi_gsum<K>(q, p.gsM, p.GL, MWG_S1(K, q.N)) : q.NB <= 8 ? pre_sum : mwi_sum_part<K>(q, p, 1));       // <X,dY> + <dX,Y> + <dX,dY>
        mw<K> mu = pre_mu;
        mw<K> r = s_div<K>(s_result<K>(s), s_mul_d<K>(mu, (double)p.Ktot));
        mw<K> beta = s_less<K>(r, from_double<K>(1.0)) ? s_mul<K>(r, r) : r;
        mw<K> beta_c;
        if (pre_flag0) {                           // the feasibility of the PREVIOUS iteration decides (:429-434 come before :441-447)
            beta_c = s_less<K>(from_double<K>(p.beta_feasible), beta) ? beta : from_double<K>(p.beta_feasible);
            if (s_less<K>(from_double<K>(1.0), beta_c)) beta_c = from_double<K>(1.0);
        } else {
            beta_c = s_less<K>(from_double<K>(p.beta_infeasible), beta) ? beta : from_double<K>(p.beta_infeasible);
        }
        stx<K>(p.sc, SP, MSC_MUS, s_mul<K>(beta_c, mu));
        p.rec[MREC_BETA] = beta_c.l[0];
        p.flags[0] = (p.rec[MREC_DERR] < p.dual_thr && p.rec[MREC_PERR] < p.primal_thr) ? 1 : 0;
        p.rec[MREC_PDFEAS] = p.flags[0];
    }
    if (stage == 3) mwi_scalar_stage3<K>(q, p);
}
template <int K, int DK>
__global__ void k_mwi_scalar(const MwDev q, const MwIpmDev p, int stage, int iter) {
    if (blockIdx.x == 0 && threadIdx.x < 64) mwi_scalar_stage<K, DK>(q, p, stage, iter);
}


// ---- R' = - X Y [- dX dY]: compute_residual_R! (src/solver.jl:961-983) without its mu_s I, which the consumers (k_mwi_Zi, k_mwi_Z)
// add on the diagonal of their first product -- the products do not depend on mu_p / mu_c, so this kernel runs beside the scalar
// stages that produce them instead of behind them -------------------------------------------------------------------------
#define MWI_EW 4              // lanes per matrix entry in the block products of the iteration
template <int K, int EW>
__device__ __forceinline__ void mwi_R_body(const MwDev &q, const MwIpmDev &p, int corrector, int bx, int by) {
    using namespace mwk;
    const MwBlk &k = q.blk[by];
    const int n = k.n;
    if (bx * (MW_NT / EW) >= n * n) return;
    const int e = bx * (MW_NT / EW) + threadIdx.x / EW, sub = threadIdx.x % EW;
    const bool live = e < n * n;
    const int ee = live ? e : 0, i = ee % n, c = ee / n;
    acc<K> s;
    acc_zero<K>(s);
    for (int kk = sub; kk < n; kk += EW) acc_fma<K, K, K>(s, ldx<K>(p.X + k.xyoff, q.xylen, i + (long)kk * n), ldx<K>(p.Y + k.xyoff, q.xylen, kk + (long)c * n), -1.0);
    if (corrector)
        for (int kk = sub; kk < n; kk += EW) acc_fma<K, K, K>(s, ldx<K>(p.dX + k.xyoff, q.xylen, i + (long)kk * n), ldx<K>(p.dY + k.xyoff, q.xylen, kk + (long)c * n), -1.0);
    const mw<K> v = lanes_sum<K, EW>(acc_result<K>(s));
    if (live && sub == 0) stx<K>(p.R + k.xyoff, q.xylen, e, v);
}
template <int K>
__global__ __launch_bounds__(MW_NT) void k_mwi_R(const MwDev q, const MwIpmDev p, int corrector) { mwi_R_body<K, MWI_EW>(q, p, corrector, blockIdx.x, blockIdx.y); }

// ---- block dot products: partial sums per PSD block ------------------------------------------------------------------
// sel bit 0: <X,Y>; bit 1: <X,dY>, <dX,Y>, <dX,dY>; bit 2: <C,Y>
// r_tiles > 0: the launch carries r_tiles * NB more workgroups that form the corrector's R' = -XY - dX dY (mwi_R_body): it depends on the
// predictor's dX, dY only, like the dot products, and not on the beta_c they lead to -- one launch, side by side, instead of two in a row
template <int K, int KA, int DK>
__device__ __forceinline__ void mwi_dots_body(const MwDev &q, const MwIpmDev &p, int sel) {
    using namespace mwk;
    const MwBlk &k = q.blk[blockIdx.x];
    const int tid = threadIdx.x;
    const long nn = (long)k.n * k.n;
    acc<KA> a0, a1, a4;
    acc_zero<KA>(a0); acc_zero<KA>(a1); acc_zero<KA>(a4);
    for (long i = tid; i < nn; i += MW_NT) {
        const long e = k.xyoff + i;
        mw<KA> Y = ldx<KA>(p.Y, q.xylen, e);
        if (sel & 1) acc_fma<KA, KA, KA>(a0, ldx<KA>(p.X, q.xylen, e), Y);
        if (sel & 2) {
            mw<KA> X = ldx<KA>(p.X, q.xylen, e), dX = ldx<KA>(p.dX, q.xylen, e), dY = ldx<KA>(p.dY, q.xylen, e);
            acc_fma<KA, KA, KA>(a1, X, dY);                   // only the sum of the three is ever used (:429): one accumulator, one reduction
            acc_fma<KA, KA, KA>(a1, dX, Y);
            acc_fma<KA, KA, KA>(a1, dX, dY);
        }
        if (sel & 4) acc_fma<KA, KA, DK>(a4, Y, ldx<DK>(p.C, q.xylen, e));
    }
    lds_d *red = MW_LDS;
    if (sel & 1) { mw<KA> r = wg_reduce_sum<KA>(acc_result<KA>(a0), red, tid); if (tid == 0) stx<K>(p.part, 5L * q.NB, 0L * q.NB + blockIdx.x, cvt<K, KA>(r)); }
    if (sel & 2) {
        mw<KA> r = wg_reduce_sum<KA>(acc_result<KA>(a1), red, tid); if (tid == 0) stx<K>(p.part, 5L * q.NB, 1L * q.NB + blockIdx.x, cvt<K, KA>(r));
    }
    if (sel & 4) { mw<KA> r = wg_reduce_sum<KA>(acc_result<KA>(a4), red, tid); if (tid == 0) stx<K>(p.part, 5L * q.NB, 4L * q.NB + blockIdx.x, cvt<K, KA>(r)); }
}
template <int K, int DK>
__global__ __launch_bounds__(MW_NT) void k_mwi_dots(const MwDev q, const MwIpmDev p, int sel, int r_tiles) {
    using namespace mwk;
    if ((int)blockIdx.x >= q.NB) {
        const int w = blockIdx.x - q.NB;
        if (r_tiles < 0) mwi_R_body<K, 16>(q, p, 1, w % (-r_tiles), w / (-r_tiles));      // (negative: sixteen lanes per entry, small problems)
        else mwi_R_body<K, MWI_EW>(q, p, 1, w % r_tiles, w / r_tiles);
        return;
    }
    if constexpr (mw_kf_of(K) < K) { if (p.klow < K) { mwi_dots_body<K, mw_kf_of(K), DK>(q, p, sel); return; } }
    mwi_dots_body<K, K, DK>(q, p, sel);
}


// ---- coefficients a_p * lambda_t of the sorted terms -------------------------------------------------------------------
template <int K, int DK>
__global__ __launch_bounds__(MW_NT) void k_mwi_coef(const MwDev q, const MwIpmDev p, const double *__restrict__ a) {
    using namespace mwk;
    const mwi64 t = (mwi64)blockIdx.x * MW_NT + threadIdx.x;
    if (t >= q.T) return;
    const int b = q.ay_blk[t];           // block of the term range (sorted and original order share it)
    if (b < 0) return;
    const MwClu &c = q.clu[q.blk[b].j];
    stx<K>(p.coef, q.T, t, mulx<K, K, DK>(ldx<K>(a, q.xlen, c.coff + q.st_p[t]), ldx<DK>(q.st_lam, q.lamp, t)));
}

// ---- sum_i a_i A_i per block (compute_weighted_A!, src/solver.jl:1410-1470) plus the rest of P or dX -------------------
// mode 0: P = sum x_i A_i - X -+ C (:882-893), max|P|;  mode 1: dX = sum dx_i A_i + P (:1585-1594)
// coef_lds = 1: the coefficients a_p lambda_t of the block's terms are formed here, into LDS (every workgroup of a block repeats the few
// products: cheaper than the launch of k_mwi_coef in front of this kernel); 0: read from p.coef (blocks with more terms than LDS holds)
// EW lanes per entry: four, or sixteen for blocks with hundreds of terms (every term is a chain of dependent gathers: its flag, its two vector
// indices, the vectors' entries -- the lanes of an entry walk their share of the terms one after the other)
template <int K, int KA, int DK, int EW>
__device__ __forceinline__ void mwi_wA_body_ka(const MwDev &q, const MwIpmDev &p, int mode, int coef_lds, int use_B) {
    using namespace mwk;
    const MwBlk &k = q.blk[blockIdx.y];
    const int n = k.n;
    if (blockIdx.x * (MW_NT / EW) >= n * n) return;
    lds_d *cf = MW_LDS;
    int t_first = 0, t_cnt = 0;
    if (coef_lds && k.kind == 0) {                        // uniform over the workgroup
        const int *tp0 = q.tptr + k.tptr_off;
        t_first = tp0[0]; t_cnt = tp0[k.P] - tp0[0];
        const double *av = mode == 0 ? p.x : p.dx;
        const MwClu &cl0 = q.clu[k.j];
        for (int t = threadIdx.x; t < t_cnt; t += MW_NT)
            stx<KA>(cf, t_cnt, t, mulx<KA, KA, DK>(ldx<KA>(av, q.xlen, cl0.coff + q.st_p[t_first + t]), ldx<DK>(q.st_lam, q.lamp, t_first + t)));
        __syncthreads();
    }
    const int e = blockIdx.x * (MW_NT / EW) + threadIdx.x / EW, sub = threadIdx.x % EW;
    const int ee = e < n * n ? e : 0, i = ee % n, c = ee / n;
    const double *a = mode == 0 ? p.x : p.dx;
    const bool mirror = (k.kind == 0 && k.m > 1);
    const bool live = e < n * n && !(mirror && c > i);     // the lower triangle is computed and mirrored (symmetric!(:L), :1462-1465)
    acc<KA> s;
    acc_zero<KA>(s);
    if (k.kind == 0) {
        const double *V = q.V + k.v_off;
        const int *tp = q.tptr + k.tptr_off;
        for (int t = tp[0] + sub; t < tp[k.P]; t += EW) {
            if (!(q.st_flag[t] & 1)) continue;     // s <= r only (:1433)
            if (use_B) {                           // coefficient times right vector formed once per term and column (k_mwi_wB): a K x DK product per entry is left
                const mw<DK> vi = ldx<DK>(V, q.Vp, i + (long)q.st_war[t] * n);
                if (vi.l[0] == 0.0) continue;
                acc_fma<KA, KA, DK>(s, ldx<KA>(p.wB, q.T * (long)p.wBn, (long)t * p.wBn + c), vi);
                continue;
            }
            const mw<DK> vi = ldx<DK>(V, q.Vp, i + (long)q.st_war[t] * n), vc = ldx<DK>(V, q.Vp, c + (long)q.st_wac[t] * n);
            if (vi.l[0] == 0.0 || vc.l[0] == 0.0) continue;
            constexpr int LL = (2 * DK + 1 < KA) ? 2 * DK + 1 : KA;
            acc_fma<KA, KA, LL>(s, coef_lds ? ldx<KA>(cf, t_cnt, t - t_first) : ldx<KA>(p.coef, q.T, t), mulx<LL, DK, DK>(vi, vc));
        }
    } else {
        const MwClu &cl = q.clu[k.j];
        const long nn = (long)n * n;
        for (int en = sub; en < k.cnt; en += EW)
            acc_fma<KA, KA, DK>(s, ldx<KA>(a, q.xlen, cl.coff + q.dense_p[k.d0 + en]), ldx<DK>(q.dA, q.dAp, k.a_off + en * nn + ee));
    }
    if (sub == 0) {
        if (mode == 0) {
            acc_add<KA, KA>(s, ldx<KA>(p.X + k.xyoff, q.xylen, ee), -1.0);
            acc_add<KA, DK>(s, ldx<DK>(p.C, q.xylen, k.xyoff + ee), -p.sgn);
        } else {
            acc_add<KA, KA>(s, ldx<KA>(p.Pm + k.xyoff, q.xylen, ee));
        }
    }
    const mw<K> v = cvt<K, KA>(lanes_sum<KA, EW>(acc_result<KA>(s)));
    if (!live || sub != 0) return;
    if (mode == 0) {
        atomic_max_abs(&p.fmax[0], v.l[0]);
        stx<K>(p.Pm + k.xyoff, q.xylen, e, v);
        // P is symmetric as a whole: -X -+ C are, and the weighted sum is mirrored
        if (mirror && c != i) stx<K>(p.Pm + k.xyoff, q.xylen, c + (long)i * n, v);
    } else {
        stx<K>(p.dX + k.xyoff, q.xylen, e, v);
        if (mirror && c != i) stx<K>(p.dX + k.xyoff, q.xylen, c + (long)i * n, v);
    }
}
template <int K, int DK, int EW>
__device__ __forceinline__ void mwi_wA_body(const MwDev &q, const MwIpmDev &p, int mode, int coef_lds, int use_B) {
    if constexpr (mw_kf_of(K) < K) { if (p.klow < K) { mwi_wA_body_ka<K, mw_kf_of(K), DK, EW>(q, p, mode, coef_lds, use_B); return; } }
    mwi_wA_body_ka<K, K, DK, EW>(q, p, mode, coef_lds, use_B);
}
template <int K, int DK>
__global__ __launch_bounds__(MW_NT) void k_mwi_wA(const MwDev q, const MwIpmDev p, int mode, int coef_lds, int ew) {
    if (ew == 17) mwi_wA_body<K, DK, 16>(q, p, mode, 0, 1);       // sixteen lanes per entry, products with the vectors' right halves from k_mwi_wB
    else if (ew == 16) mwi_wA_body<K, DK, 16>(q, p, mode, coef_lds, 0);
    else mwi_wA_body<K, DK, MWI_EW>(q, p, mode, coef_lds, 0);
}
// wB[t][c] = a_p(t) lambda_t V[c, wac(t)] for every term (s <= r) of every low-rank block: with hundreds of terms per block (the three-point bound:
// 204-408) forming this once per term and column leaves sum a_i A_i a K x DK product per term and entry instead of a DK x DK and a K x (2 DK + 1) one
template <int K, int DK>
__global__ __launch_bounds__(MW_NT) void k_mwi_wB(const MwDev q, const MwIpmDev p, int mode) {
    using namespace mwk;
    const mwi64 e = (mwi64)blockIdx.x * MW_NT + threadIdx.x;
    const mwi64 t = e / p.wBn;
    const int c = (int)(e % p.wBn);
    if (t >= q.T) return;
    const int b = q.ay_blk[t];
    if (b < 0 || !(q.st_flag[t] & 1)) return;
    const MwBlk &k = q.blk[b];
    if (c >= k.n) return;
    const MwClu &cl = q.clu[k.j];
    const double *a = mode == 0 ? p.x : p.dx;
    const mw<K> cf = mulx<K, K, DK>(ldx<K>(a, q.xlen, cl.coff + q.st_p[t]), ldx<DK>(q.st_lam, q.lamp, t));
    const mw<DK> vc = ldx<DK>(q.V + k.v_off, q.Vp, c + (long)q.st_wac[t] * k.n);
    stx<K>(p.wB, q.T * (long)p.wBn, t * p.wBn + c, mulx<K, K, DK>(cf, vc));
}

// ---- T = M V for the low-rank blocks (first half of trace_A, src/solver.jl:1334-1341) -----------------------------------
template <int K, int DK>
__global__ __launch_bounds__(MW_NT) void k_mwi_MV(const MwDev q, const double *__restrict__ M) {
    using namespace mwk;
    const MwBlk &k = q.blk[q.lr_list[blockIdx.y]];
    const int n = k.n, dl = k.delta;
    if (blockIdx.x * (MW_NT / MWI_EW) >= n * k.U) return;
    const int e = blockIdx.x * (MW_NT / MWI_EW) + threadIdx.x / MWI_EW, sub = threadIdx.x % MWI_EW;
    const bool live = e < n * k.U;
    const int ee = live ? e : 0, i = ee % n, c = ee / n;
    const double *V = q.V + k.v_off;
    const int r0 = q.vrow[k.vrow_off + c];
    acc<K> s;
    acc_zero<K>(s);
    for (int kk = r0 + sub; kk < r0 + dl; kk += MWI_EW) acc_fma<K, K, DK>(s, ldx<K>(M + k.xyoff, q.xylen, i + (long)kk * n), ldx<DK>(V, q.Vp, kk + (long)c * n));
    const mw<K> v = lanes_sum<K, MWI_EW>(acc_result<K>(s));
    if (live && sub == 0) stx<K>(q.Tm + k.z_off, q.zlen, e, v);
}

// ---- per constraint row: d = c - <A_*,Y> - B y (:863-879) or rhs_x = -d - <A_*,Z> (:1518-1523) --------------------------
// Dense part of the row traces, <A_g, M> summed over the dense matrices of constraint row g: nn-term dot products, ONE WAVE per
// row (the per-row lane group of k_mwi_rows would walk them with eight lanes, and rows of one wave touch different blocks).
template <int K, int DK>
__global__ __launch_bounds__(MW_NT) void k_mwi_rows_dn(const MwDev q, const MwIpmDev p, int mode) {
    using namespace mwk;
    const mwi64 g = (mwi64)blockIdx.x * (MW_NT / 64) + (threadIdx.x >> 6);
    if (g >= q.xlen) return;                            // uniform over the wave
    const int lane = threadIdx.x & 63;
    const double *M = mode == 0 ? p.Y : p.dY;
    acc<K> s;
    acc_zero<K>(s);
    for (int t = q.drow_ptr[g]; t < q.drow_ptr[g + 1]; t++) {
        const MwBlk &k = q.blk[q.drow_blk[t]];
        const long nn = (long)k.n * k.n, a0 = k.a_off + (long)q.drow_en[t] * nn;
        for (long i = lane; i < nn; i += 64) acc_fma<K, K, DK>(s, ldx<K>(M + k.xyoff, q.xylen, i), ldx<DK>(q.dA, q.dAp, a0 + i));
    }
    const mw<K> v = lanes_sum<K, 64>(acc_result<K>(s));
    if (lane == 0) stx<K>(p.dtr, q.xlen, g, v);
}

#define MWI_RW 8              // lanes per constraint row
// BG groups of MWI_RW lanes per row, each group taking every BG-th PSD block of the row's cluster: a cluster of dozens of blocks (the three-point
// bound: 43) is otherwise a chain of as many dependent dot products per row, on seven workgroups
template <int K, int DK, int BG>
__device__ __forceinline__ void mwi_rows_body(const MwDev &q, const MwIpmDev &p, int mode, mwi64 g0, bool live) {
    using namespace mwk;
    constexpr int RWT = MWI_RW * BG;                        // lanes per row
    const int sub = threadIdx.x % MWI_RW, grp = (threadIdx.x % RWT) / MWI_RW, sub_all = threadIdx.x % RWT;
    const mwi64 g = live ? g0 : 0;
    const int j = p.row_clu[g];
    const MwClu &cl = q.clu[j];
    const int pp = (int)(g - cl.coff);
    const double *M = mode == 0 ? p.Y : p.dY;        // Z is kept in the dY buffer, as the reference does (:1501-1514)
    // every lane accumulates its share of  -(trace term) [- (B y)_g] ; the lanes are summed at the end
    acc<K> s;
    acc_zero<K>(s);
    // mode 0 with p.tau: tau_g = <A_g, X^-1> beside d_g -- the same sums with the pairings w^T X^-1 v of the assembly (q.AX) in place of w^T Y v: what mu_c
    // multiplies in the corrector's right-hand side (Z = sym(X^-1 (P Y - R)) = Z0 - mu_c X^-1, rhs_x = -d - <A, Z>)
    const bool with_tau = mode == 0 && p.tau != nullptr && q.AX != nullptr;
    acc<K> st;
    acc_zero<K>(st);
    for (int b = cl.b0 + grp; b < cl.b1; b += BG) {
        const MwBlk &k = q.blk[b];
        const int n = k.n;
        if (k.kind == 0) {
            const int *tp = q.tptr + k.tptr_off;
            if (mode == 0) {
                for (int t = tp[pp] + sub; t < tp[pp + 1]; t += MWI_RW) {
                    const int fl = q.st_flag[t];
                    if (!(fl & 1)) continue;                                           // s <= r (:1310)
                    const mw<DK> w = mul_pow2<DK>(ldx<DK>(q.st_lam, q.lamp, t), (fl & 2) ? 2.0 : 1.0);      // off-diagonal sub-blocks count twice (:1354-1356)
                    acc_fma<K, K, DK>(s, ldx<K>(q.AY, q.T, q.st_orig[t]), w, -1.0);     // trace_A with (Y, A_Y), :1368-1407
                    if (with_tau) acc_fma<K, K, DK>(st, ldx<K>(q.AX, q.T, q.st_orig[t]), w);
                }
            } else {
                const double *V = q.V + k.v_off;
                for (int t = tp[pp]; t < tp[pp + 1]; t++) {
                    const int fl = q.st_flag[t];
                    if (!(fl & 1)) continue;
                    const mw<DK> w = mul_pow2<DK>(ldx<DK>(q.st_lam, q.lamp, t), (fl & 2) ? 2.0 : 1.0);
                    const int l = q.st_trl[t], dcol = q.st_trd[t], r0 = q.vrow[k.vrow_off + l];
                    acc<K> z;
                    acc_zero<K>(z);
                    for (int ii = r0 + sub; ii < r0 + k.delta; ii += MWI_RW) acc_fma<K, K, DK>(z, ldx<K>(q.Tm + k.z_off, q.zlen, ii + (long)dcol * n), ldx<DK>(V, q.Vp, ii + (long)l * n));
                    acc_fma<K, K, DK>(s, acc_result<K>(z), w, -1.0);
                }
            }
        } else if (!q.dn_big) {
            const int en = q.dmap[k.dmap_off + pp];
            if (en >= 0) {
                const long nn = (long)n * n;
                for (long i = sub; i < nn; i += MWI_RW) acc_fma<K, K, DK>(s, ldx<K>(M + k.xyoff, q.xylen, i), ldx<DK>(q.dA, q.dAp, k.a_off + en * nn + i), -1.0);
                if (with_tau && sub == 0) {                                              // (1 x 1 dense blocks only: the host asks for tau only then) X^-1 = Xi^2
                    const mw<K> xi = ldx<K>(q.Xi + k.xyoff, q.xylen, 0);
                    acc_fma<K, K, DK>(st, mul<K>(xi, xi), ldx<DK>(q.dA, q.dAp, k.a_off + en * nn));
                }
            }
        }
    }
    if (q.dn_big && sub_all == 0) acc_add<K, K>(s, ldx<K>(p.dtr, q.xlen, g), -1.0);      // the dense matrices of the row: k_mwi_rows_dn
    if (mode == 0) {
        if (sub_all == 0) acc_add<K, DK>(s, ldx<DK>(p.c, q.xlen, g));
        for (int a = sub_all; a < q.N; a += RWT) acc_fma<K, K, DK>(s, ldx<K>(p.y, q.N, a), ldx<DK>(q.B, q.Bp, g + (long)a * q.xlen), -1.0);
        const mw<K> dv = lanes_sum<K, RWT>(acc_result<K>(s));
        if (live && sub_all == 0) {
            atomic_max_abs(&p.fmax[1], dv.l[0]);
            stx<K>(p.d, q.xlen, g, dv);
        }
        if (with_tau) {
            const mw<K> tv = lanes_sum<K, RWT>(acc_result<K>(st));
            if (live && sub_all == 0) stx<K>(p.tau, q.xlen, g, tv);
        }
    } else {
        if (sub_all == 0) acc_add<K, K>(s, ldx<K>(p.d, q.xlen, g), -1.0);
        const mw<K> r = lanes_sum<K, RWT>(acc_result<K>(s));
        if (live && sub_all == 0) stx<K>(p.rhsx, q.xlen, g, r);
    }
}
template <int K, int DK>
__global__ __launch_bounds__(MW_NT) void k_mwi_rows(const MwDev q, const MwIpmDev p, int mode, int groups) {
    if (groups == 8) { const mwi64 g0 = (mwi64)blockIdx.x * (MW_NT / (MWI_RW * 8)) + threadIdx.x / (MWI_RW * 8); mwi_rows_body<K, DK, 8>(q, p, mode, g0, g0 < q.xlen); }
    else { const mwi64 g0 = (mwi64)blockIdx.x * (MW_NT / MWI_RW) + threadIdx.x / MWI_RW; mwi_rows_body<K, DK, 1>(q, p, mode, g0, g0 < q.xlen); }
}
// rhs_x of the corrector and the first product pair of its solve in one launch, per cluster: the rows of a cluster are all its workgroup of the
// solve needs (small unsharded systems; one launch less on the chain of the iteration)
template <int K, int DK>
__global__ __launch_bounds__(MW_NT) void k_mwi_rows_fwd(const MwDev q, const MwIpmDev p) {
    const MwClu &c = q.clu[blockIdx.x];
    for (int r0 = 0; r0 < c.P; r0 += MW_NT / MWI_RW) {
        const int r = r0 + threadIdx.x / MWI_RW;
        mwi_rows_body<K, DK, 1>(q, p, 1, c.coff + r, r < c.P);
    }
    __threadfence_block();
    __syncthreads();
    mw_solve_fwd_cluster<K>(q, blockIdx.x, p.rhsx);
}

// ---- "the side stream's work of iteration `iter` is done": one thread, behind that work in stream order.  The workgroups that ride on the Cholesky of Q
// (the first products of the predictor's solve) wait for this word INSIDE their launch instead of the main stream waiting for an event in front of it:
// an event awaited on the main stream costs it a bubble of 6-15 us (measured, profiles/r03, r04), while those workgroups have the 80 us of the launch.
template <int UNIT>      // (a template only so that the units that include this header do not each define the symbol)
__global__ void k_mwi_mark(int *word, int iter) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(word, iter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- the other direction: the side stream waits, in a kernel of its own (one wave, bounded polling), for a word that a launch of the main stream stores
// as it starts (MwDev::mark_word) -- an event RECORDED on the main stream costs it a bubble of ~5 us (measured: profiles/r04/r4_host_vs_gpu), a word costs it nothing
template <int UNIT>
__global__ void k_mwi_wait(const int *word, int value, int *info, int code) {
    (void)code;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned spins = 0;
        unsigned long long t0 = 0;
        while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < value) {      // (it may wait half an iteration: a poll per ~1 us)
            __builtin_amdgcn_s_sleep(32);
            if ((++spins & 1023u) == 0) {                      // bounded by WALL CLOCK (MW_WAIT_TICKS), not by a poll count: a long wait is not a failure
                const unsigned long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                else if (now - t0 > MW_WAIT_TICKS) { atomicMin(info, MW_INFO_TIMEOUT); break; }      // the iteration reports error code 5, never numbers
            }
        }
    }
}

// ---- this rank's slot of a gather buffer (cluster sharding): one workgroup, the first wave --------------------------------------
template <int K, int DK>
__global__ void k_mwi_gpack(const MwDev q, const MwIpmDev p, int stage) {
    using namespace mwk;
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    const int lane = threadIdx.x, N = q.N;
    double *slot = ((stage == 2 || stage == 3) ? p.gsM : p.gsS) + (long)q.rank * p.GL, *D = slot + MWG_D(K, N);
    // stage 15: ONE record for everything the side stream needs of the iterate the last update produced -- the objectives (stage 4), <X,Y> (stage 0 of the
    // next iteration) and what depends on the iterate alone among the residuals (stage 1: -B^T x, which k_mwi_pv has written to the slot, and max|P|; max|d|
    // needs the pairings of the next assembly and travels with stage 2)
    const bool with_xy = stage == 15;
    if (with_xy) stage = 4;
    if (stage == 4) {                              // <c,x> over this rank's rows, by the wave
        acc<K> s;
        acc_zero<K>(s);
        for (long i = lane; i < q.xlen; i += 64) acc_fma<K, K, DK>(s, ldx<K>(p.x, q.xlen, i), ldx<DK>(p.c, q.xlen, i), p.sgn);
        const mw<K> cx = s_lanes64<K>(s_result<K>(s));
        if (lane == 0) {
#pragma unroll
            for (int l = 0; l < K; l++) slot[MWG_S2(K, N) + l] = cx.l[l];
        }
    }
    if (lane != 0) return;
    if (stage == 0 || stage == 2 || stage == 4) {
        const mw<K> v = mwi_sum_part<K>(q, p, stage == 0 ? 0 : stage == 2 ? 1 : 4);
#pragma unroll
        for (int l = 0; l < K; l++) slot[MWG_S1(K, N) + l] = v.l[l];
    }
    if (with_xy) {
        const mw<K> v = mwi_sum_part<K>(q, p, 0);
#pragma unroll
        for (int l = 0; l < K; l++) slot[MWG_XY(K, N) + l] = v.l[l];
        D[0] = __longlong_as_double((long long)p.fmax[0]);
        D[1] = 0.0;
    }
    if (stage == 1) {
        D[0] = __longlong_as_double((long long)p.fmax[0]);
        D[1] = __longlong_as_double((long long)p.fmax[1]);
    }
    if (stage == 2) {                              // max|d| of this rank's rows; the decomposition's status words, as global numbers
        D[1] = __longlong_as_double((long long)p.fmax[1]);
        const int fs = q.info[0], xs = q.info[1];
        D[4] = fs == MW_INFO_NONE ? MW_INFO_NONE : fs > q.J ? p.Jglob + 1 : p.clu_gid ? p.clu_gid[fs - 1] + 1 : fs;
        D[5] = xs == MW_INFO_NONE ? MW_INFO_NONE : p.blk_gid ? p.blk_gid[xs - 1] + 1 : xs;
    }
    if (stage == 3) {
        for (int w = 0; w < 2; w++) {
            double mn = p.eig[(long)w * q.NB];
            for (int b = 1; b < q.NB; b++) mn = fmin(mn, p.eig[(long)w * q.NB + b]);
            D[2 + w] = mn;
        }
        D[6] = p.flags[2];
        // the first-pass accuracy of the corrector's solve must be judged alike on every rank (the factor stage's limb count follows from it, and Q is factored
        // by all of them): this rank's four maxima travel with the step lengths, in the K-limb slots this record does not use
        if (q.refstat) {
#pragma unroll
            for (int i = 0; i < 4; i++) slot[MWG_S1(K, N) + i] = __longlong_as_double((long long)q.refstat[i]);
        }
    }
}
// sharded: p = +-b + sum over the ranks (rank order) of their -B^T x, max|p|; and the maxima of |P|, |d| over the ranks
template <int K, int DK>
__global__ __launch_bounds__(MW_NT) void k_mwi_pvfin(const MwDev q, const MwIpmDev p) {
    using namespace mwk;
    const int N = q.N;
    for (int a = blockIdx.x * MW_NT + threadIdx.x; a < N; a += gridDim.x * MW_NT) {
        acc<K> s;
        acc_zero<K>(s);
        acc_add<K, DK>(s, ldx<DK>(p.b, N, a), p.sgn);
        for (int r = 0; r < q.world; r++) acc_add<K, K>(s, ldx<K>(p.gsS + (long)r * p.GL + MWG_BX(K, N), N, a));
        const mw<K> v = acc_result<K>(s);
        atomic_max_abs(&p.fmax[2], v.l[0]);
        stx<K>(p.pv, N, a, v);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double mP = 0, md = 0;
        for (int r = 0; r < q.world; r++) {
            const double *D = p.gsS + (long)r * p.GL + MWG_D(K, N);
            mP = fmax(mP, D[0]); md = fmax(md, D[1]);
        }
        p.fmax[0] = (unsigned long long)__double_as_longlong(mP);
        p.fmax[1] = (unsigned long long)__double_as_longlong(md);
    }
}

// p = +-b - B^T x  (:899-916); sharded (q.world > 1): only -B^T x over this rank's rows, into its gather slot (k_mwi_pvfin completes it)
template <int K, int DK>
__global__ __launch_bounds__(MW_NT) void k_mwi_pv(const MwDev q, const MwIpmDev p, int iter) {
    using namespace mwk;
    const int a = blockIdx.x * (MW_NT / 8) + threadIdx.x / 8, sub = threadIdx.x % 8;      // eight lanes per free variable
    const bool live = a < q.N;
    const int aa = live ? a : 0;
    acc<K> s;
    acc_zero<K>(s);
    const bool part = q.world > 1;
    if (sub == 0 && !part) acc_add<K, DK>(s, ldx<DK>(p.b, q.N, aa), p.sgn);
    for (long g = sub; g < q.xlen; g += 8) acc_fma<K, K, DK>(s, ldx<K>(p.x, q.xlen, g), ldx<DK>(q.B, q.Bp, g + (long)aa * q.xlen), -1.0);
    mw<K> v = lanes_sum<K, 8>(acc_result<K>(s));
    if (live && sub == 0) {
        if (part) stx<K>(p.gsS + (long)q.rank * p.GL + MWG_BX(K, q.N), q.N, a, v);
        else {
            atomic_max_abs(&p.fmax[2], v.l[0]);
            stx<K>(p.pv, q.N, a, v);
        }
    }
    (void)iter;      // the errors of the residuals (scalar stage 1) are taken at the head of stage 2, when the decomposition has ended as well
}

// ---- which 0: Z = sym(X^-1 (P Y - R)) (:1501-1514);  which 1: dY = sym(X^-1 (R - dX Y)) (:1597-1613); both into dY --------
template <int K>
__global__ __launch_bounds__(MW_NT) void k_mwi_Z(const MwDev q, const MwIpmDev p, int which, int lds_L) {
    using namespace mwk;
    const MwBlk &k = q.blk[blockIdx.x];
    if (k.inv) return;                                  // k_mwi_Zi
    const int n = k.n, tid = threadIdx.x;
    const long nn = (long)n * n;
    lds_d *M = MW_LDS;
    const double *A = (which == 0 ? p.Pm : p.dX) + k.xyoff;
    const double sg = which == 0 ? 1.0 : -1.0;
    for (int e = tid; e < nn; e += MW_NT) {
        const int i = e % n, c = e / n;
        acc<K> s;
        acc_zero<K>(s);
        for (int kk = 0; kk < n; kk++) acc_fma<K, K, K>(s, ldx<K>(A, q.xylen, i + (long)kk * n), ldx<K>(p.Y + k.xyoff, q.xylen, kk + (long)c * n), sg);
        acc_add<K, K>(s, ldx<K>(p.R + k.xyoff, q.xylen, e), -sg);
        if (i == c) acc_add<K, K>(s, ldx<K>(p.sc, MSC_COUNT, MSC_MUS), -sg);          // R = mu_s I + R'
        stx<K>(M, nn, e, acc_result<K>(s));
    }
    // X^-1 M with the scaled triangles of chol(X) (this iteration's k_mw_potrf_x left them in the context)
    if (lds_L) {
        lds_d *Lf = MW_LDS + (long)K * nn, *Lb = Lf + (long)K * nn;
        wg_copy<K>(Lf, nn, n, q.Xf + k.xyoff, q.xylen, n, n, n, tid);
        wg_copy<K>(Lb, nn, n, q.Xb + k.xyoff, q.xylen, n, n, n, tid);
        __syncthreads();
        wg_trsm_f<K>(Lf, nn, n, q.xrd + k.rd_off, q.xrdlen, n, M, nn, n, n, tid);
        wg_trsm_b<K>(Lb, nn, n, q.xrd + k.rd_off, q.xrdlen, n, M, nn, n, n, tid);
    } else {
        __syncthreads();
        wg_trsm_f<K>(q.Xf + k.xyoff, q.xylen, n, q.xrd + k.rd_off, q.xrdlen, n, M, nn, n, n, tid);
        wg_trsm_b<K>(q.Xb + k.xyoff, q.xylen, n, q.xrd + k.rd_off, q.xrdlen, n, M, nn, n, n, tid);
    }
    for (int e = tid; e < nn; e += MW_NT) {
        const int i = e % n, c = e / n;
        if (c > i) continue;
        mw<K> v = mul_pow2<K>(add<K>(ldx<K>(M, nn, i + (long)c * n), ldx<K>(M, nn, c + (long)i * n)), 0.5);
        stx<K>(p.dY + k.xyoff, q.xylen, i + (long)c * n, v);
        stx<K>(p.dY + k.xyoff, q.xylen, c + (long)i * n, v);
    }
}

// The same with the explicit inverse Xi = chol(X)^-1 that k_mw_potrf_x leaves beside the factor: X^-1 M = Xi^T (Xi M), three
// block products.  They are independent column by column, and one compute unit issues them no faster than its four SIMDs
// allow, so a block is split over MWI_ZS or more workgroups (panels of at most eight columns) by column panels (eight lanes per entry, MW_PT threads); the panels go to
// a scratch matrix and the workgroup of a block that finishes last symmetrises it (a counter per block).
#define MWI_ZS 4
#ifndef MWI_ZL
#define MWI_ZL 8          // (16 lanes per entry with two-column panels -- one term per lane and product -- is slower: 0.4178 against 0.4150 ms per iteration)
#endif
// KA <= K: limbs of the three products (MwIpmDev::klow); the panels in the scratch matrix and the symmetrised result carry K planes, the upper ones zero
template <int K, int KA>
__device__ __forceinline__ void mwi_Zi_body(const MwDev &q, const MwIpmDev &p, int which) {
    using namespace mwk;
    const bool nomu = which == 2;                        // which 2: which 0 without the mu_s I of R (the corrector's Z0: mw_ipm_enqueue adds mu_c <A, X^-1> to the traces instead)
    if (nomu) which = 0;
    const MwBlk &k = q.blk[blockIdx.x];
    if (!k.inv) return;
    const int n = k.n, tid = threadIdx.x, sub = tid % MWI_ZL;
    const int zs = gridDim.y, pc0 = (n + zs - 1) / zs, c0 = blockIdx.y * pc0, pc = max(0, min(pc0, n - c0));     // this workgroup's columns
    const long np = (long)n * pc0;
    lds_d *M = MW_LDS, *M2 = M + (long)K * np;
    const double *A = (which == 0 ? p.Pm : p.dX) + k.xyoff, *Xi = q.Xi + k.xyoff;
    const double sg = which == 0 ? 1.0 : -1.0;
#ifdef CLRS_MW_STAMPS            // diagnostic builds only (CLRS_MW_STAMPS=1 python -c "... _lib.build()": scripts/zi_stamps.py): the product carries no run-time probe
    const bool stamp = p.stamps && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0;
    int nst = 0;
#define MWZ_STAMP() do { if (stamp) p.stamps[nst++] = wall_clock64(); } while (0)
#else
#define MWZ_STAMP() do { } while (0)
#endif
    MWZ_STAMP();
    for (int e0 = 0; e0 < n * pc; e0 += MW_PT / MWI_ZL) {  // M = sg (A Y - R)
        const int e = e0 + tid / MWI_ZL;
        const bool live = e < n * pc;
        const int ee = live ? e : 0, i = ee % n, c = c0 + ee / n;
        acc<KA> s;
        acc_zero<KA>(s);
        for (int kk = sub; kk < n; kk += MWI_ZL) acc_fma<KA, KA, KA>(s, ldx<KA>(A, q.xylen, i + (long)kk * n), ldx<KA>(p.Y + k.xyoff, q.xylen, kk + (long)c * n), sg);
        if (sub == 0) acc_add<KA, KA>(s, ldx<KA>(p.R + k.xyoff, q.xylen, i + (long)c * n), -sg);
        if (sub == 1 && i == c && !nomu) acc_add<KA, KA>(s, ldx<KA>(p.sc, MSC_COUNT, MSC_MUS), -sg);      // R = mu_s I + R'
        const mw<KA> v = lanes_sum<KA, MWI_ZL>(acc_result<KA>(s));
        if (live && sub == 0) stx<KA>(M, np, ee, v);
    }
    __syncthreads();
    MWZ_STAMP();
    for (int e0 = 0; e0 < n * pc; e0 += MW_PT / MWI_ZL) {  // M2 = Xi M
        const int e = e0 + tid / MWI_ZL;
        const bool live = e < n * pc;
        const int ee = live ? e : 0, i = ee % n, cl = ee / n;
        acc<KA> s;
        acc_zero<KA>(s);
        for (int r = sub; r <= i; r += MWI_ZL) acc_fma<KA, KA, KA>(s, ldx<KA>(Xi, q.xylen, i + (long)r * n), ldx<KA>(M, np, r + (long)cl * n));
        const mw<KA> v = lanes_sum<KA, MWI_ZL>(acc_result<KA>(s));
        if (live && sub == 0) stx<KA>(M2, np, ee, v);
    }
    __syncthreads();
    MWZ_STAMP();
    for (int e0 = 0; e0 < n * pc; e0 += MW_PT / MWI_ZL) {  // Xi^T M2 -> scratch
        const int e = e0 + tid / MWI_ZL;
        const bool live = e < n * pc;
        const int ee = live ? e : 0, i = ee % n, cl = ee / n;
        acc<KA> s;
        acc_zero<KA>(s);
        for (int r = i + sub; r < n; r += MWI_ZL) acc_fma<KA, KA, KA>(s, ldx<KA>(Xi, q.xylen, r + (long)i * n), ldx<KA>(M2, np, r + (long)cl * n));
        const mw<KA> v = lanes_sum<KA, MWI_ZL>(acc_result<KA>(s));
        if (live && sub == 0) stx<K>(p.Zs + k.xyoff, q.xylen, i + (long)(c0 + cl) * n, cvt<K, KA>(v));
    }
    MWZ_STAMP();
    if (!mwi_last_block(&p.zcnt[blockIdx.x], zs)) return;
    MWZ_STAMP();
    for (int e = tid; e < n * n; e += MW_PT) {
        const int i = e % n, c = e / n;
        if (c > i) continue;
        const mw<K> v = cvt<K, KA>(mul_pow2<KA>(add<KA>(ldx<KA>(p.Zs + k.xyoff, q.xylen, i + (long)c * n), ldx<KA>(p.Zs + k.xyoff, q.xylen, c + (long)i * n)), 0.5));
        stx<K>(p.dY + k.xyoff, q.xylen, i + (long)c * n, v);
        stx<K>(p.dY + k.xyoff, q.xylen, c + (long)i * n, v);
    }
    MWZ_STAMP();
#undef MWZ_STAMP
}
template <int K>
__global__ __launch_bounds__(MW_PT) void k_mwi_Zi(const MwDev q, const MwIpmDev p, int which) {
    mw_mark(q);
    if constexpr (mw_kf_of(K) < K) { if (p.klow < K) { mwi_Zi_body<K, mw_kf_of(K)>(q, p, which); return; } }
    mwi_Zi_body<K, K>(q, p, which);
}

// The same three products for LARGE blocks (sides beyond ~24), as launches of their own over 8 x 8 output tiles, four lanes per entry: a column panel
// of k_mwi_Zi walks whole matrices with 64 entries at a time (21 passes of seven multiply-adds per lane at n = 54, every workgroup reading all of A
// and Xi), a tile reads eight rows and eight columns and its lanes take n / 4 multiply-adds.  op 0: Zt = sg (A Y - R' - mu_s I); 1: Zs = Xi Zt;
// 2: Zt = Xi^T Zs; 3: dY = sym(Zt).  Blocks without an inverse factor stay with k_mwi_Z.
#ifndef MWI_BT
#define MWI_BT 8             // (4 x 4 tiles with sixteen lanes per entry: Nsphere_packing N = 3 1.888 -> 1.877 ms per iteration, but SDPA x64 -- 64 blocks of 32 rows -- 1.64 -> 1.91)
#endif
template <int K>
__global__ __launch_bounds__(MW_NT) void k_mwi_bmm(const MwDev q, const MwIpmDev p, int op, int which) {
    using namespace mwk;
    const MwBlk &k = q.blk[blockIdx.y];
    if (!k.inv) return;
    const int n = k.n, nt = (n + MWI_BT - 1) / MWI_BT;
    if ((int)blockIdx.x >= nt * nt) return;
    const int ti = blockIdx.x % nt, tj = blockIdx.x / nt;
    if (op == 3) {
        if (tj > ti) return;
        const int e = threadIdx.x;
        if (e >= MWI_BT * MWI_BT) return;
        const int i = ti * MWI_BT + e % MWI_BT, c = tj * MWI_BT + e / MWI_BT;
        if (i >= n || c > i) return;
        mw<K> v = mul_pow2<K>(add<K>(ldx<K>(p.Zt + k.xyoff, q.xylen, i + (long)c * n), ldx<K>(p.Zt + k.xyoff, q.xylen, c + (long)i * n)), 0.5);
        stx<K>(p.dY + k.xyoff, q.xylen, i + (long)c * n, v);
        stx<K>(p.dY + k.xyoff, q.xylen, c + (long)i * n, v);
        return;
    }
    constexpr int LW = MW_NT / (MWI_BT * MWI_BT);          // lanes per entry: 4
    const int e = threadIdx.x / LW, sub = threadIdx.x % LW;
    const int i0 = ti * MWI_BT + e % MWI_BT, c0 = tj * MWI_BT + e / MWI_BT;
    const bool live = i0 < n && c0 < n;
    const int wh = blockIdx.z;
    if (op >= 4) {
        // the congruences of the step lengths (compute_step_length, :1647-1659), both matrices in one launch (blockIdx.z: 0 = X, 1 = Y):
        // 4: U = dM Li^T into Zs / Zt;  5: W = Li U, lower triangle, rounded to fp64 into Wd -- the matrix k_mwi_step (inv_path 3) takes the eigenvalue of
        if (n == 1 || (wh == 1 && p.yfail[blockIdx.y])) return;
        if (op == 5 && tj > ti) return;
    }
    // The operands of the tile through LDS: eight rows of the left matrix and eight columns of the right one over the tile's range of k
    // (read straight from memory, every lane fetched its own two K-limb numbers per multiply-add: 164 KB per workgroup, 170 MB per launch
    // at 64 blocks of 32 x 32 -- the launches ran at the L2's bandwidth, not at the pipe's).  Triangular factors are stored with their zeros,
    // so the range of k is the tile's, not the entry's.
    const double *Lp, *Rp;
    int l_si, l_sk, r_sk, r_sc, k_lo = 0, k_hi = n;
    const double *Xi = q.Xi + k.xyoff;
    if (op == 0) { Lp = (which == 0 ? p.Pm : p.dX) + k.xyoff; l_si = 1; l_sk = n; Rp = p.Y + k.xyoff; r_sk = 1; r_sc = n; }
    else if (op == 1) { Lp = Xi; l_si = 1; l_sk = n; Rp = p.Zt + k.xyoff; r_sk = 1; r_sc = n; k_hi = min(n, ti * MWI_BT + MWI_BT); }
    else if (op == 2) { Lp = Xi; l_si = n; l_sk = 1; Rp = p.Zs + k.xyoff; r_sk = 1; r_sc = n; k_lo = ti * MWI_BT; }
    else if (op == 4) { Lp = (wh == 0 ? p.dX : p.dY) + k.xyoff; l_si = 1; l_sk = n; Rp = (wh == 0 ? q.Xi : p.Yi) + k.xyoff; r_sk = n; r_sc = 1; k_hi = min(n, tj * MWI_BT + MWI_BT); }
    else { Lp = (wh == 0 ? q.Xi : p.Yi) + k.xyoff; l_si = 1; l_sk = n; Rp = (wh == 0 ? p.Zs : p.Zt) + k.xyoff; r_sk = 1; r_sc = n; k_hi = min(n, ti * MWI_BT + MWI_BT); }
    const int klen = k_hi - k_lo;
    const long lp = (long)MWI_BT * n;                       // plane of the staged operands (klen <= n)
    lds_d *Ls = MW_LDS, *Rs = Ls + (long)K * lp;
    for (int idx = threadIdx.x; idx < MWI_BT * klen; idx += MW_NT) {
        const int ii = l_si == 1 ? idx % MWI_BT : idx / klen, kk = l_si == 1 ? idx / MWI_BT : idx % klen;     // the contiguous index runs fastest
        const int gi = ti * MWI_BT + ii;
        stx<K>(Ls, lp, kk * MWI_BT + ii, gi < n ? ldx<K>(Lp, q.xylen, (long)gi * l_si + (long)(k_lo + kk) * l_sk) : zero<K>());
    }
    for (int idx = threadIdx.x; idx < MWI_BT * klen; idx += MW_NT) {
        const int cc = r_sc == 1 ? idx % MWI_BT : idx / klen, kk = r_sc == 1 ? idx / MWI_BT : idx % klen;
        const int gc = tj * MWI_BT + cc;
        stx<K>(Rs, lp, kk * MWI_BT + cc, gc < n ? ldx<K>(Rp, q.xylen, (long)(k_lo + kk) * r_sk + (long)gc * r_sc) : zero<K>());
    }
    __syncthreads();
    const int i = live ? i0 : 0, c = live ? c0 : 0, ii = e % MWI_BT, cc = e / MWI_BT;
    acc<K> s;
    acc_zero<K>(s);
    const double sg = (op == 0 && which != 0) ? -1.0 : 1.0;
    if (op != 5 || i >= c)
        for (int kk = sub; kk < klen; kk += LW) acc_fma<K, K, K>(s, ldx<K>(Ls, lp, kk * MWI_BT + ii), ldx<K>(Rs, lp, kk * MWI_BT + cc), sg);
    if (op == 0) {
        if (sub == 0) acc_add<K, K>(s, ldx<K>(p.R + k.xyoff, q.xylen, i + (long)c * n), -sg);
        if (sub == 1 && i == c) acc_add<K, K>(s, ldx<K>(p.sc, MSC_COUNT, MSC_MUS), -sg);      // R = mu_s I + R'
    }
    const mw<K> v = lanes_sum<K, LW>(acc_result<K>(s));
    if (!live || sub != 0) return;
    if (op == 5) {
        if (i >= c) p.Wd[(long)wh * q.xylen + k.xyoff + i + (long)c * n] = v.l[0];
    } else if (op == 4) stx<K>((wh == 0 ? p.Zs : p.Zt) + k.xyoff, q.xylen, i + (long)c * n, v);
    else stx<K>((op == 1 ? p.Zs : p.Zt) + k.xyoff, q.xylen, i + (long)c * n, v);
}

// W = Li dM Li^T with the explicit inverse Li of the factor (two block products), symmetrised and rounded to fp64: the matrix of :1659
template <int K, class PI>
__device__ __forceinline__ void mwi_step_congruence_inv(PI Li, long iplane, int n, const double *dMg, long gplane, mwk::lds_d *T1, mwk::lds_d *Wd, int tid) {
    using namespace mwk;
    const long nn = (long)n * n;
    for (int e = tid; e < nn; e += MW_NT) {                // T1 = Li dM
        const int i = e % n, c = e / n;
        acc<K> s;
        acc_zero<K>(s);
        for (int r = 0; r <= i; r++) acc_fma<K, K, K>(s, ldx<K>(Li, iplane, i + (long)r * n), ldx<K>(dMg, gplane, r + (long)c * n));
        stx<K>(T1, nn, e, acc_result<K>(s));
    }
    __syncthreads();
    for (int e = tid; e < nn; e += MW_NT) {                // W = T1 Li^T, lower triangle, mirrored
        const int i = e % n, c = e / n;
        if (c > i) continue;
        acc<K> s;
        acc_zero<K>(s);
        for (int r = 0; r <= c; r++) acc_fma<K, K, K>(s, ldx<K>(T1, nn, i + (long)r * n), ldx<K>(Li, iplane, c + (long)r * n));
        const double w = acc_result<K>(s).l[0];
        Wd[i + (long)c * n] = w;
        Wd[c + (long)i * n] = w;
    }
}

// W = L^-1 dM L^-T through two forward substitutions and a transposition (:1651-1655), then its fp64 symmetrisation
template <int K, class PF, class PR, class PW>
__device__ __forceinline__ void mwi_step_congruence(PF F, PR rd, PW W, long wplane, long nn, int n, const double *dMg, long gplane, mwk::lds_d *Wd, int tid) {
    using namespace mwk;
    wg_copy<K>(W, wplane, n, dMg, gplane, n, n, n, tid);
    __syncthreads();
    wg_trsm_f<K>(F, nn, n, rd, n, n, W, wplane, n, n, tid);                              // :1651
    for (int e = tid; e < nn; e += MW_NT) {                                              // transpose :1652
        const int i = e % n, c = e / n;
        if (c >= i) continue;
        mw<K> a = ldx<K>(W, wplane, i + (long)c * n), b2 = ldx<K>(W, wplane, c + (long)i * n);
        stx<K>(W, wplane, i + (long)c * n, b2);
        stx<K>(W, wplane, c + (long)i * n, a);
    }
    __syncthreads();
    wg_trsm_f<K>(F, nn, n, rd, n, n, W, wplane, n, n, tid);                              // :1655
    for (int e = tid; e < nn; e += MW_NT) {
        const int i = e % n, c = e / n;
        Wd[e] = 0.5 * ((double)W[i + (long)c * n] + (double)W[c + (long)i * n]);         // heads of the limbs: the Float64 matrix of :1659
    }
}

// The congruence by COLUMN PANELS over gridDim.z workgroups per (block, which), for blocks whose inverse factors are both in memory:
// W[:, c] = Li (dM Li^T[:, c]) is independent column by column (two products, four lanes per entry, rows i >= c only: the lower triangle);
// the panels go to p.Wd as fp64 heads, the workgroup of a (block, which) that arrives last mirrors them in LDS and takes the eigenvalue.
// One workgroup issues the two 16-term products of a 16 x 16 block no faster than its four SIMDs allow (14 us); four share them.
#define MWI_SW 4
// KA: limbs of the two products (the factor stage's count while the refined solves run on reduced factors: the congruence is rounded to fp64, and the
// inverse factor it is taken with was computed in KA limbs); SW: lanes per entry -- 16 with one-column panels of blocks of at most 16 rows, where a
// product is then ONE term per lane and phase instead of up to four (12 -> 6 us of the 46 of k_mwi_step on the named problem)
template <int K, int KA, int SW>
__device__ __forceinline__ bool mwi_step_panels_ka(const MwDev &q, const MwIpmDev &p, const MwBlk &k, int which, const double *Li, const double *dMg) {
    using namespace mwk;
    const int n = k.n, tid = threadIdx.x, sub = tid % SW;
    const int zs = gridDim.z, pc0 = (n + zs - 1) / zs, c0 = blockIdx.z * pc0, pc = max(0, min(pc0, n - c0));
    const long np = (long)n * pc0;
    lds_d *Us = MW_LDS;                                   // U = dM Li^T[:, panel], n x pc0, KA limbs planar
    for (int e0 = 0; e0 < n * pc; e0 += MW_NT / SW) {
        const int e = e0 + tid / SW;
        const bool live = e < n * pc;
        const int ee = live ? e : 0, r = ee % n, c = c0 + ee / n;
        acc<KA> s;
        acc_zero<KA>(s);
        for (int t = sub; t <= c; t += SW) acc_fma<KA, KA, KA>(s, ldx<KA>(dMg, q.xylen, r + (long)t * n), ldx<KA>(Li, q.xylen, c + (long)t * n));
        const mw<KA> v = lanes_sum<KA, SW>(acc_result<KA>(s));
        if (live && sub == 0) stx<KA>(Us, np, ee, v);
    }
    __syncthreads();
    MWS_STAMP(1);
    double *Wg = p.Wd + (long)which * q.xylen + k.xyoff;
    for (int e0 = 0; e0 < n * pc; e0 += MW_NT / SW) {
        const int e = e0 + tid / SW;
        const bool live = e < n * pc;
        const int ee = live ? e : 0, i = ee % n, cl = ee / n, c = c0 + cl;
        acc<KA> s;
        acc_zero<KA>(s);
        if (i >= c)
            for (int r = sub; r <= i; r += SW) acc_fma<KA, KA, KA>(s, ldx<KA>(Li, q.xylen, i + (long)r * n), ldx<KA>(Us, np, r + (long)cl * n));
        const mw<KA> v = lanes_sum<KA, SW>(acc_result<KA>(s));
        if (live && sub == 0 && i >= c) Wg[i + (long)c * n] = v.l[0];
    }
    MWS_STAMP(2);
    return mwi_last_block(&p.wcnt[which * q.NB + blockIdx.x], zs);
}
template <int K>
__device__ __forceinline__ bool mwi_step_panels(const MwDev &q, const MwIpmDev &p, const MwBlk &k, int which, const double *Li, const double *dMg) {
    const bool wide = (int)gridDim.z >= k.n && k.n <= 16;        // one column per workgroup
    if constexpr (mw_kf_of(K) < K) {
        if (q.kf < K) return wide ? mwi_step_panels_ka<K, mw_kf_of(K), 16>(q, p, k, which, Li, dMg) : mwi_step_panels_ka<K, mw_kf_of(K), MWI_SW>(q, p, k, which, Li, dMg);
    }
    return wide ? mwi_step_panels_ka<K, K, 16>(q, p, k, which, Li, dMg) : mwi_step_panels_ka<K, K, MWI_SW>(q, p, k, which, Li, dMg);
}

// ---- compute_step_length (:1620-1693) per block: smallest eigenvalue of L^-1 dM L^-T, L = chol(M) -----------------------
// which 0: (X, dX) with the factors of this iteration; which 1: (Y, dY), factored here
template <int K>
__device__ __forceinline__ bool mwi_step_body(const MwDev &q, const MwIpmDev &p, int w_in_lds, int inv_path, int which_base) {
    using namespace mwk;
    const int which = which_base + blockIdx.y;          // both step lengths in one launch (grid.y = 2), or one launch each
    MWS_STAMP(0);
    const MwBlk &k = q.blk[blockIdx.x];
    const int n = k.n, tid = threadIdx.x;
    const long nn = (long)n * n;
    const double *Mg = (which == 0 ? p.X : p.Y) + k.xyoff, *dMg = (which == 0 ? p.dX : p.dY) + k.xyoff;
    // launched with column panels (gridDim.z > 1): only the inverse-factor path of blocks with n > 1 is split; everything else is workgroup z = 0's
    if (blockIdx.z != 0 && (n == 1 || inv_path != 2 || (which == 1 && p.yfail[blockIdx.x]))) return false;
    if (n == 1) {
        if (tid == 0) {
            mw<K> m = ldx<K>(Mg, q.xylen, 0);
            if (!(m.l[0] > 0.0)) { p.flags[2] = 1; p.eig[(long)which * q.NB + blockIdx.x] = 0.0; }
            else p.eig[(long)which * q.NB + blockIdx.x] = div<K>(ldx<K>(dMg, q.xylen, 0), m).l[0];     // :1637-1641
        }
        return true;
    }
    if (inv_path >= 2) {
        // both inverse factors are in memory (Xi from k_mw_potrf_x, Yi from its second half): LDS holds T1, the fp64 matrix, the eigenvalue work space
        lds_d *T1 = MW_LDS, *Wd = T1 + (long)K * nn, *work = Wd + nn;
        if (which == 1 && p.yfail[blockIdx.x]) {                                         // :1644-1646
            if (tid == 0) { p.flags[2] = 1; p.eig[(long)which * q.NB + blockIdx.x] = 0.0; }
            return true;
        }
        if (gridDim.z > 1 || inv_path == 3) {             // 3: the congruence is in Wd already (tiled launches of k_mwi_bmm, large blocks)
            if (inv_path == 2 && !mwi_step_panels<K>(q, p, k, which, (which == 0 ? q.Xi : p.Yi) + k.xyoff, dMg)) return false;      // not the last workgroup of this (block, which)
            MWS_STAMP(3);
            lds_d *Wl = MW_LDS, *wk = Wl + nn;
            const double *Wg = p.Wd + (long)which * q.xylen + k.xyoff;
            for (int e = tid; e < nn; e += MW_NT) {
                const int i = e % n, c = e / n;
                Wl[e] = i >= c ? Wg[e] : Wg[c + (long)i * n];
            }
            __syncthreads();
            MWS_STAMP(4);
            const double ev = n <= 64 ? wg_min_eig32(Wl, n, wk, tid) : wg_min_eig(Wl, n, wk, tid);
            if (tid == 0) p.eig[(long)which * q.NB + blockIdx.x] = ev - 1e-5;            // :1662
            MWS_STAMP(5);
            return true;
        }
        mwi_step_congruence_inv<K>((which == 0 ? q.Xi : p.Yi) + k.xyoff, q.xylen, n, dMg, q.xylen, T1, Wd, tid);
        __syncthreads();
        const double ev = n <= 64 ? wg_min_eig32(Wd, n, work, tid) : wg_min_eig(Wd, n, work, tid);
        if (tid == 0) p.eig[(long)which * q.NB + blockIdx.x] = ev - 1e-5;                // :1662
        return true;
    }
    if (inv_path && k.inv == 1) {
        // LDS: Li (inverse factor; which 0 reads Xi from memory instead), T1, [Y and its factor], rd, the fp64 matrix, eigenvalue work space, scratch
        lds_d *Li = MW_LDS, *T1 = Li + (long)K * nn, *Mf = T1 + (long)K * nn;
        lds_d *rd = Mf + (long)K * nn, *Wd = rd + (long)K * n, *work = Wd + nn, *scr = work + 3 * n + 2 * MW_NT;
        if (which == 0) {
            mwi_step_congruence_inv<K>(q.Xi + k.xyoff, q.xylen, n, dMg, q.xylen, T1, Wd, tid);
        } else {
            wg_copy<K>(Mf, nn, n, Mg, q.xylen, n, n, n, tid);
            __syncthreads();
            if (!wg_potrf<K, true, MW_NT, false>(Mf, nn, n, n, rd, n, Li, nn, n, scr, tid)) {                // :1644-1646
                if (tid == 0) { p.flags[2] = 1; p.eig[(long)which * q.NB + blockIdx.x] = 0.0; }
                return true;
            }
            mwi_step_congruence_inv<K>(Li, nn, n, dMg, q.xylen, T1, Wd, tid);
        }
        __syncthreads();
        const double ev = n <= 64 ? wg_min_eig32(Wd, n, work, tid) : wg_min_eig(Wd, n, work, tid);
        if (tid == 0) p.eig[(long)which * q.NB + blockIdx.x] = ev - 1e-5;                // :1662
        return true;
    }
    // LDS: F (row-scaled factor), [W], rd, the fp64 matrix and the eigenvalue work space, the broadcast slot of the factorisation.
    // Blocks too large for two multi-word matrices in LDS keep W in global memory (the R and P buffers are dead at this point of the iteration).
    lds_d *F = MW_LDS;
    lds_d *Wl = F + (long)K * nn;
    lds_d *rd = w_in_lds ? Wl + (long)K * nn : Wl;
    lds_d *Wd = rd + (long)K * n, *work = Wd + nn, *bc = work + 3 * n + 2 * MW_NT;
    if (which == 0) {
        wg_copy<K>(F, nn, n, q.Xf + k.xyoff, q.xylen, n, n, n, tid);
        for (int i = tid; i < n; i += MW_NT) stx<K>(rd, n, i, ldx<K>(q.xrd + k.rd_off, q.xrdlen, i));
        __syncthreads();
    } else {
        wg_copy<K>(F, nn, n, Mg, q.xylen, n, n, n, tid);
        __syncthreads();
        if (!wg_potrf<K, false>(F, nn, n, n, rd, n, F, 0, 0, bc, tid)) {                                 // :1644-1646
            if (tid == 0) { p.flags[2] = 1; p.eig[(long)which * q.NB + blockIdx.x] = 0.0; }
            return true;
        }
        for (int e = tid; e < nn; e += MW_NT) {                                          // row-scaled strict lower triangle, in place
            const int i = e % n, c = e / n;
            if (i > c) stx<K>(F, nn, e, mul<K>(ldx<K>(F, nn, e), ldx<K>(rd, n, i)));
        }
        __syncthreads();
    }
    if (w_in_lds) mwi_step_congruence<K>(F, rd, Wl, nn, nn, n, dMg, q.xylen, Wd, tid);
    else mwi_step_congruence<K>(F, rd, (which == 0 ? p.R : p.Pm) + k.xyoff, q.xylen, nn, n, dMg, q.xylen, Wd, tid);    // R and P are dead here; one each
    __syncthreads();
    const double ev = n <= 64 ? wg_min_eig32(Wd, n, work, tid) : wg_min_eig(Wd, n, work, tid);
    if (tid == 0) p.eig[(long)which * q.NB + blockIdx.x] = ev - 1e-5;                    // :1662
    return true;
}

// ---- x, X += alpha_d (dx, dX); y, Y += alpha_p (dy, dY)  (:485-495); skipped when the iteration ended with an error ------
template <int K>
__global__ __launch_bounds__(MW_NT) void k_mwi_update(const MwDev q, const MwIpmDev p) {
    using namespace mwk;
    if (p.flags[1] != 0 || p.flags[6] != 0) return;
    const double ad = p.sc[10], ap = p.sc[11];
    const mwi64 tot = q.xylen + q.xlen + q.N;
    for (mwi64 i = (mwi64)blockIdx.x * MW_NT + threadIdx.x; i < tot; i += (mwi64)gridDim.x * MW_NT) {
        if (i < q.xylen) {
            acc<K> s;
            acc_zero<K>(s);
            acc_add<K, K>(s, ldx<K>(p.X, q.xylen, i));
            acc_fma_d<K, K>(s, ldx<K>(p.dX, q.xylen, i), ad);
            stx<K>(p.X, q.xylen, i, acc_result<K>(s));
            acc_zero<K>(s);
            acc_add<K, K>(s, ldx<K>(p.Y, q.xylen, i));
            acc_fma_d<K, K>(s, ldx<K>(p.dY, q.xylen, i), ap);
            stx<K>(p.Y, q.xylen, i, acc_result<K>(s));
        } else if (i < q.xylen + q.xlen) {
            const mwi64 g = i - q.xylen;
            acc<K> s;
            acc_zero<K>(s);
            acc_add<K, K>(s, ldx<K>(p.x, q.xlen, g));
            acc_fma_d<K, K>(s, ldx<K>(p.dx, q.xlen, g), ad);
            stx<K>(p.x, q.xlen, g, acc_result<K>(s));
        } else {
            const mwi64 a = i - q.xylen - q.xlen;
            acc<K> s;
            acc_zero<K>(s);
            acc_add<K, K>(s, ldx<K>(p.y, q.N, a));
            acc_fma_d<K, K>(s, ldx<K>(p.dy, q.N, a), ap);
            stx<K>(p.y, q.N, a, acc_result<K>(s));
        }
    }
}

template <int K, int DK>
__global__ __launch_bounds__(MW_NT) void k_mwi_step(const MwDev q, const MwIpmDev p, int w_in_lds, int inv_path, int iter, int which_base) {
    if (!mwi_step_body<K>(q, p, w_in_lds, inv_path, which_base)) return;      // a column panel that was not the last one of its (block, which)
    // the workgroup that arrives last -- of the 2 NB that end a (block, which), in one launch or two -- takes the step lengths
    // (moving the iterate here as well, by this one workgroup, is slower than the launch of k_mwi_update it saves: 45 against 35 + 6 us)
    if (q.world > 1) return;                            // sharded: the minima travel first (k_mwi_gpack, all-gather, k_mwi_scalar stage 3)
#ifdef CLRS_MW_STAMPS
    if (mwi_last_block(&p.flags[4], 2u * q.NB)) {
        if (threadIdx.x == 0) mwi_scalar_stage3<K>(q, p);
        if (threadIdx.x == 0 && p.stamps) {
            const int me = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
            unsigned long long t0 = ~0ull;
            const int nw = gridDim.x * gridDim.y * gridDim.z;
            for (int w = 0; w < nw && w < 256; w++) t0 = mwk::g_mws_stamp[w][0] < t0 ? mwk::g_mws_stamp[w][0] : t0;
            p.stamps[8] = t0;
            { const int sel[6] = {0, 2, 4, 6, 7, 5}; for (int i = 0; i < 6; i++) p.stamps[9 + i] = mwk::g_mws_stamp[me][sel[i]]; }
            p.stamps[15] = wall_clock64();
        }
    }
    return;
#endif
    if (mwi_last_block(&p.flags[4], 2u * q.NB) && threadIdx.x == 0) mwi_scalar_stage3<K>(q, p);
}

// X = omega_p I, Y = omega_d I, x = y = 0 (:187-201)
template <int K>
__global__ __launch_bounds__(MW_NT) void k_mwi_init(const MwDev q, const MwIpmDev p, double omega_p, double omega_d) {
    const MwBlk &k = q.blk[blockIdx.x];
    const int n = k.n;
    for (int e = threadIdx.x; e < n * n; e += MW_NT) {
        const int i = e % n, c = e / n;
        for (int l = 0; l < K; l++) {
            p.X[(long)l * q.xylen + k.xyoff + e] = (l == 0 && i == c) ? omega_p : 0.0;
            p.Y[(long)l * q.xylen + k.xyoff + e] = (l == 0 && i == c) ? omega_d : 0.0;
        }
    }
}
