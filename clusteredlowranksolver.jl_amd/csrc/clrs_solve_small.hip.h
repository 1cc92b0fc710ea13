// clrs_solve_small.hip.h -- k_solve_small2: the whole solve stage of compute_search_direction! (src/solver.jl:1527-1582) in one
// workgroup, latency first.
//
// For the named configurations (a handful of clusters, P_j <= 128, N <= 128) the stage is a chain of dependent steps on ~40 KB
// of factors: t_j = L_j^-1 rhs_x[j];  u = sum_j LinvB_j^T t_j;  dy = Q^-1 (rhs_y - u);  dx_j = L_j^-T (t_j + LinvB_j dy).
// k_solve_small (clrs_fused.hip.h) walks it with one global -> LDS load phase per cluster and phase (five round trips to memory
// on cohnelkies(8,15)) and all four waves on every triangular solve.  Here
//   * every factor (L_j, L_Q, LinvB) and both right-hand sides are loaded ONCE, all loads in flight together;
//   * a triangular solve with one right-hand side is a single-wave job: the clusters are dealt to the waves (they run
//     concurrently), 16 unknowns at a time live in the 16 lanes of a DPP row, and one elimination is ONE instruction
//     (v_fmac_f64_dpp row_newbcast on the row-scaled factor, see clrs_assemble_w3.hip.h); the rows below / above a finished
//     panel are updated by the 64 lanes with 16 multiply-adds each;
//   * the two matrix-vector products are split over all 256 threads with a fixed-order LDS reduction (deterministic).
// Five workgroup barriers in all.
#pragma once
#include <algorithm>
#include <cstring>
#include <vector>
#include "clrs_fused.hip.h"

namespace clrs {

// (trsv16_chain_fwd / _bwd and wave_trsv_fwd / _bwd, the single-wave triangular solves, live in clrs_wave.hip.h)

#ifdef CLRS_W3_STAMPS
__device__ unsigned long long g_ss2_stamps[16];
#define SS2_STAMP(i) do { if (tid == 0) g_ss2_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SS2_STAMP(i) do { } while (0)
#endif

// ---- one-round-trip staging of many small matrices: a job table in the kernel arguments ---------------------------------
// A kernel that stages its operands matrix by matrix (load tile, store to LDS, next matrix) pays one trip to memory per
// matrix: the stores of one matrix sit between its loads and the loads of the next.  Here the host lists every 16 x 16 tile /
// 256-entry vector piece to stage as a job in the kernel arguments; all loads are issued (straight-line code, one register per
// job and thread), then all stores.  Thread (r = tid & 15, c = tid >> 4) of the 256 moves entry (r, c) of a tile.
struct StageJob {
    const double *src;     // tile origin
    unsigned dst_ldd;      // bits 0-15: destination offset in LDS (doubles); 16-24: destination leading dimension; 25-29: columns written
    unsigned meta;         // bits 0-15: source leading dimension; 16-20: valid rows; 21-25: valid columns.  Entry (r, c), r, c < 16, is loaded
                           // when r < rows and c < cols; the destination gets it, or zero, for every c < columns written (all 16 rows).
                           // A piece of a vector is a "tile" with both leading dimensions 16 (entry index r + 16 c)
};
constexpr int STAGE_MAX_JOBS = 48;
struct StageJobs {
    StageJob j[STAGE_MAX_JOBS];
    int n, pad;
};
static inline StageJob stage_rect(const double *src, int ldg, int rows, int cols, int dst, int ldd, int wcols = 16) {
    StageJob b;
    b.src = src; b.dst_ldd = (unsigned)dst | ((unsigned)ldd << 16) | ((unsigned)wcols << 25); b.meta = (unsigned)ldg | ((unsigned)rows << 16) | ((unsigned)cols << 21);
    return b;
}
// `count` consecutive entries from src to LDS offset dst, zeros up to `write` (a multiple of 16) entries: whole 16-entry columns as
// pieces of up to 256, the remainder (and the zero fill) as one more piece with a single partly valid column
static inline void stage_flat(std::vector<StageJob> &v, const double *src, int count, int write, int dst) {
    const int full = count / 16, rem = count % 16;
    for (int c0 = 0; c0 < full; c0 += 16) {
        const int nc = std::min(16, full - c0);
        v.push_back(stage_rect(src ? src + 16 * c0 : nullptr, 16, 16, nc, dst + 16 * c0, 16, nc));
    }
    for (int c0 = full; c0 < write / 16; c0 += 16) {                 // at most the first of these has a valid (partial) column
        const int nc = std::min(16, write / 16 - c0);
        const bool part = c0 == full && rem > 0;                       // a pure zero-fill piece reads (and ignores) the first entry of the vector
        v.push_back(stage_rect(src ? (part ? src + 16 * c0 : src) : nullptr, 16, part ? rem : 0, part ? 1 : 0, dst + 16 * c0, 16, nc));
    }
}
// The table itself must not be read entry by entry (scalar loads from the kernel-argument segment, or LDS reads of a copy: each
// entry's use then waits for its own fetch, 400 cycles per job measured): lane t of every wave fetches job t with ONE 16-byte
// vector load from the kernel-argument segment, and the unrolled loops below pick job t out of lane t with v_readlane
// (compile-time lane, a few cycles, no memory access).
// NJ (compile time) jobs are processed without any branch: the host pads the table with empty jobs (nothing valid, nothing written).
// A branch per job would make the compiler drain vmcnt before every v_readlane (it cannot see that the table's load, waited for
// on the path through the previous job, is complete on every path) -- one trip to memory per job again.
template <int NJ>
__device__ __forceinline__ void stage_all(const StageJobs &kjobs, double *lds, int tid) {
    const int r = tid & 15, c = tid >> 4, lane = tid & 63;
    static_assert(NJ <= STAGE_MAX_JOBS && STAGE_MAX_JOBS <= 64, "one job per lane");
    const StageJob mine = kjobs.j[lane < NJ ? lane : 0];
    const unsigned long long msrc = (unsigned long long)mine.src;
    const int m_lo = (int)(unsigned)msrc, m_hi = (int)(unsigned)(msrc >> 32), m_dl = (int)mine.dst_ldd, m_meta = (int)mine.meta;
    // validity as integer arithmetic on the VALU (sign bits), not compares: a compare writes an SGPR pair that the scalar unit then
    // combines and hands back through VCC -- four VALU <-> SALU hand-offs per job were most of the 215 cycles a job cost
    double v[NJ];
#pragma unroll
    for (int t = 0; t < NJ; t++) {
        const unsigned meta = (unsigned)__builtin_amdgcn_readlane(m_meta, t);
        const unsigned long long sp = (unsigned long long)(unsigned)__builtin_amdgcn_readlane(m_lo, t) | ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(m_hi, t) << 32);
        // a GLOBAL pointer (address space 1): through a generic pointer this would be a flat_load, which completes out of order, and the
        // compiler would drain vmcnt AND lgkmcnt before every later v_readlane of the table -- one trip to memory per job again
        const __attribute__((address_space(1))) double *src = (const __attribute__((address_space(1))) double *)sp;
        const int rows = (meta >> 16) & 31, cols = (meta >> 21) & 31;
        const int mask = ((r - rows) & (c - cols)) >> 31;                 // all ones: r < rows and c < cols
        v[t] = src[(r + c * (int)(meta & 0xffff)) & mask];
    }
#ifdef CLRS_W3_STAMPS
    if (tid == 0) g_ss2_stamps[8] = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) g_ss2_stamps[9] = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int t = 0; t < NJ; t++) {
        const unsigned meta = (unsigned)__builtin_amdgcn_readlane(m_meta, t), dl = (unsigned)__builtin_amdgcn_readlane(m_dl, t);
        const int rows = (meta >> 16) & 31, cols = (meta >> 21) & 31;
        const long long mask = (long long)(((r - rows) & (c - cols)) >> 31);
        const double val = __longlong_as_double(__double_as_longlong(v[t]) & mask);
        if (c < (int)((dl >> 25) & 31)) lds[(int)(dl & 0xffff) + r + c * (int)((dl >> 16) & 511)] = val;
    }
}

// LDS doubles k_solve_small2 needs (the host uses the same formula to decide whether the kernel applies)
static inline size_t solve_small2_lds_doubles(const int *P, int J, int N, long long xlen) {
    size_t tot = 0;
    for (int j = 0; j < J; j++) {
        const size_t P16 = (size_t)((P[j] + 15) & ~15);
        tot += (P16 + 2) * P16 + 2 * P16;          // L_j, 1 / diag, padded work vector
    }
    const size_t N16 = (size_t)((N + 15) & ~15), xl = (size_t)((xlen + 15) & ~15);
    tot += (N16 + 2) * N16 + N16;                  // L_Q, 1 / diag
    tot += (size_t)xlen * (size_t)N;               // LinvB
    tot += xl + N16 + 8 * N16 + 4 * xl;            // t, dy work, partial sums of the two products
    return tot + 16;
}

// The cluster descriptors travel BY VALUE in the kernel arguments (J <= 8): a descriptor fetched from memory is a dependent
// round trip (~1 us) in front of every load that needs its pointers.
struct CSolve8 {
    CSolve d[8];
};

// The jobs of k_solve_small2, in the LDS layout the kernel carves (returns false when they do not fit the table: the caller keeps
// k_solve_small).  rhs_x / rhs_y are bound per call: their jobs are listed in rx_job[j] / ry_job for patching at launch.
// phase 0: the whole stage; 1: the part before the exchange of u (forward solves, partial u); 2: the part after it (Q solve, backward
// solves), which takes t and the summed u back from the context's buffers d_t / d_u (the cluster-sharded path)
static inline bool solve_small2_jobs(StageJobs &jb, const CSolve *cs, int J, const double *LQ, const double *dinvQ, int N, long long xlen,
                                     const double *LBall, int *rx_job, int *ry_job, int phase = 0, const double *d_t = nullptr, const double *d_u = nullptr) {
    std::vector<StageJob> v;
    int o = 0;
    for (int j = 0; j < J; j++) {
        const int P = cs[j].P, P16 = (P + 15) & ~15, lda = P16 + 2, nt = P16 / 16;
        for (int tj = 0; tj < nt; tj++)
            for (int ti = tj; ti < nt; ti++)       // lower tiles only: the solves never read above the diagonal tiles
                v.push_back(stage_rect(cs[j].L + ti * 16 + (long long)tj * 16 * P, P, std::min(16, P - ti * 16), std::min(16, P - tj * 16), o + ti * 16 + tj * 16 * lda, lda));
        stage_flat(v, cs[j].dinv, P, P16, o + lda * P16);
        rx_job[j] = (int)v.size();                   // the pieces of rhs_x[j] follow: sources patched per call (src = base + what stage_flat added)
        if (phase != 2) stage_flat(v, nullptr, P, P16, o + lda * P16 + P16);
        rx_job[8 + j] = (int)v.size();
        if (P16 > 256) return false;
        o += lda * P16 + 2 * P16;
    }
    const int N16 = (N + 15) & ~15, ldq = N16 + 2, ntq = N16 / 16, xl = (int)((xlen + 15) & ~15);
    if (N16 > 256) return false;
    if (phase != 1)
        for (int tj = 0; tj < ntq; tj++)
            for (int ti = tj; ti < ntq; ti++)
                v.push_back(stage_rect(LQ + ti * 16 + (long long)tj * 16 * N, N, std::min(16, N - ti * 16), std::min(16, N - tj * 16), o + ti * 16 + tj * 16 * ldq, ldq));
    ry_job[0] = ry_job[1] = -1;
    const int oLB = o + ldq * N16 + N16, ott = oLB + (int)(xlen * N), oyy = ott + xl, opu = oyy + N16;
    if (N > 0) {
        if (phase != 1) {
            stage_flat(v, dinvQ, N, N16, o + ldq * N16);
            ry_job[0] = (int)v.size();
            stage_flat(v, nullptr, N, N16, oyy);
            ry_job[1] = (int)v.size();
        }
        if (phase == 2) stage_flat(v, d_u, N, N16, opu);             // the summed u, parked where the partial sums live in the other phases
        // LinvB: whole 16-entry columns are written, the last piece may spill (zeros) into t, which is filled later
        stage_flat(v, LBall, (int)(xlen * N), (int)((xlen * N + 15) & ~15ll), oLB);
    }
    // t of the first part: AFTER the LinvB pieces, whose last one may spill zeros over the first entries of t (same threads of the
    // same wave write both, in job order)
    if (phase == 2) stage_flat(v, d_t, (int)xlen, xl, ott);
    if (v.size() > (size_t)STAGE_MAX_JOBS) return false;
    std::memset(&jb, 0, sizeof(jb));
    for (size_t i = 0; i < v.size(); i++) jb.j[i] = v[i];
    jb.n = (int)v.size();
    for (int i = jb.n; i < STAGE_MAX_JOBS; i++) jb.j[i] = stage_rect(cs[0].L, 16, 0, 0, 0, 16, 0);     // empty: reads one valid word, writes nothing
    return true;
}

// NJ: number of staging jobs processed (the table is padded with empty jobs up to it).  PH: 0 = the whole stage; 1 / 2 = the parts
// before / after the exchange of u between the ranks of the cluster-sharded path (t_out, u_out: the context's d_t, d_u buffers).
template <int NJ, int PH>
__global__ __launch_bounds__(256) void k_solve_small2(const CSolve8 descs8, const StageJobs jobs, int J, int N, int xlen, double *__restrict__ dx, double *__restrict__ dy,
                                                      double *__restrict__ t_out, double *__restrict__ u_out) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ int offA[8], offZ[8];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int N16 = (N + 15) & ~15, ldq = N16 + 2, xl = (xlen + 15) & ~15;
    SS2_STAMP(0);
    // ---- LDS layout (same order as solve_small2_lds_doubles): the cluster part needs the descriptors, see below ----
    // ---- every load of the stage in one trip to memory (job table built by the host: solve_small2_jobs) ----
    __shared__ CSolve descs_lds[8];
    if (tid >= 64 && tid < 64 + J) descs_lds[tid - 64] = descs8.d[tid - 64];
    stage_all<NJ>(jobs, lds, tid);
    __syncthreads();
    int o = 0;
    for (int j = 0; j < J; j++) {
        const int P16 = (descs_lds[j].P + 15) & ~15;
        if (tid == 0) { offA[j] = o; offZ[j] = o + (P16 + 2) * P16 + P16; }
        o += (P16 + 2) * P16 + 2 * P16;
    }
    double *AQ = lds + o, *dvQ = AQ + ldq * N16, *LBs = dvQ + N16, *tt = LBs + (size_t)xlen * N, *yy = tt + xl, *pu = yy + N16, *pw = pu + 8 * N16;
    __syncthreads();
    SS2_STAMP(1);
    // ---- t_j = L_j^-1 rhs_x[j]: one wave per cluster, concurrently (src/solver.jl:1537-1540) ----
    if (PH != 2) {
        for (int j = wave; j < J; j += 4) {
            const CSolve d = descs_lds[j];
            const int P16 = (d.P + 15) & ~15, lda = P16 + 2;
            double *A = lds + offA[j], *dv = A + lda * P16, *z = lds + offZ[j];
            wave_trsv_fwd(A, lda, dv, z, P16, lane);
            for (int i = lane; i < d.P; i += 64) {
                tt[d.off + i] = z[i];
                if (PH == 1) t_out[d.off + i] = z[i];
            }
        }
        __syncthreads();
    }
    SS2_STAMP(2);
    if (N > 0) {
        // ---- u = LinvB^T t (src/solver.jl:1546), 8 partial sums per column in a fixed order; dy = Q^-1 (rhs_y - u) (:1550-1558) ----
        if (PH != 2) {
            for (int k0 = 0; k0 < N; k0 += 32) {
                const int k = k0 + (tid >> 3), part = tid & 7;
                double s = 0.0;
                if (k < N)
                    for (int i = part; i < xlen; i += 8) s = __builtin_fma(LBs[i + (size_t)k * xlen], tt[i], s);
                if (k < N) pu[k * 8 + part] = s;
            }
            __syncthreads();
        }
        if (tid < N) {
            const double *p = pu + tid * 8;
            const double usum = (PH == 2) ? pu[tid] : ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
            if (PH == 1) u_out[tid] = usum;                        // this rank's share of u: summed over the ranks by the caller
            else yy[tid] -= usum;
        }
        if (PH == 1) return;
        __syncthreads();
        SS2_STAMP(3);
        if (wave == 0) {
            wave_trsv_fwd(AQ, ldq, dvQ, yy, N16, lane);
            wave_trsv_bwd(AQ, ldq, dvQ, yy, N16, lane);
            for (int k = lane; k < N; k += 64) dy[k] = yy[k];
        }
        __syncthreads();
        SS2_STAMP(4);
        // ---- t_j + LinvB_j dy (src/solver.jl:1567-1569): wave w sums the columns k = w mod 4, then a fixed-order sum of the four ----
        for (int i = lane; i < xlen; i += 64) {
            double s = 0.0;
            for (int k = wave; k < N; k += 4) s = __builtin_fma(LBs[i + (size_t)k * xlen], yy[k], s);
            pw[wave * xl + i] = s;
        }
        __syncthreads();
    }
    SS2_STAMP(5);
    if (PH == 1) return;
    // ---- dx_j = L_j^-T (...) (src/solver.jl:1570-1573) ----
    for (int j = wave; j < J; j += 4) {
        const CSolve d = descs_lds[j];
        const int P16 = (d.P + 15) & ~15, lda = P16 + 2;
        double *A = lds + offA[j], *dv = A + lda * P16, *z = lds + offZ[j];
        for (int i = lane; i < d.P; i += 64) {
            double s = tt[d.off + i];
            if (N > 0) s += (pw[d.off + i] + pw[xl + d.off + i]) + (pw[2 * xl + d.off + i] + pw[3 * xl + d.off + i]);
            z[i] = s;
        }
        wave_sync();
        wave_trsv_bwd(A, lda, dv, z, P16, lane);
        for (int i = lane; i < d.P; i += 64) dx[d.off + i] = z[i];
    }
    SS2_STAMP(6);
}

}  // namespace clrs
