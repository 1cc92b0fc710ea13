// clrs_factor_small.hip.h -- k_factor_small: steps 3-4 of compute_T_decomposition! (src/solver.jl:1244-1279) in ONE launch of ONE
// workgroup, for problems with a handful of small clusters (the named configurations).
//
//     L_j = chol(S_j),  LinvB_j = L_j^-1 B_j,  Q = sum_j LinvB_j^T LinvB_j,  L_Q = chol(Q)
//
// k_cluster_factor (one workgroup per cluster) + k_small_potrf (Q) are two launches with the partial Q_j passing through memory,
// and each stages its operands matrix by matrix.  Here every S_j and B_j is staged in one trip to memory (job table in the kernel
// arguments: clrs_solve_small.hip.h), with two or more clusters each cluster is factored by ONE wave on its own (the blocked
// routines of clrs_wave.hip.h with wave-level ordering instead of workgroup barriers: the clusters run concurrently, nothing
// waits for a slower wave until Q), a single cluster is worked on by all four waves; Q never leaves LDS before it is factored.
#pragma once
#include <type_traits>
#include "clrs_solve_small.hip.h"

namespace clrs {

struct FSmallCluster {
    double *S;         // P x P (ld P): receives L_j, zero above the diagonal (as approx_cholesky!, src/tools.jl:100-105)
    double *LB;        // rows of the cluster in the stacked LinvB (ld ldb)
    double *dinv;      // P: 1 / diag(L_j)
    int P, code;       // failure code of the cluster (j + 1)
};
struct FSmallArgs {
    FSmallCluster c[4];
    double *Q, *dinvQ;     // N x N (ld N): receives L_Q; N: 1 / diag(L_Q)
    int J, N, ldb, codeQ;
};

// LDS doubles of k_factor_small (the host uses the same formula)
static inline size_t factor_small_lds_doubles(const int *P, int J, int N) {
    const size_t N16 = (size_t)((N + 15) & ~15), ldq = N16 + 2;
    size_t tot = 0;
    for (int j = 0; j < J; j++) {
        const size_t P16 = (size_t)((P[j] + 15) & ~15), lda = P16 + 2;
        tot += lda * P16 + P16 + lda * N16 + ldq * N16;      // S_j / L_j, 1 / diag, B_j / LinvB_j, partial Q_j
    }
    return tot + ldq * N16 + N16 + 16;                          // Q, 1 / diag
}

// staging jobs: lower tiles of S_j, all tiles of B_j (zero padded to P16 x N16)
static inline bool factor_small_jobs(StageJobs &jb, const FSmallArgs &a, const double *const *S_src, const double *const *B_src) {
    std::vector<StageJob> v;
    const int N = a.N, N16 = (N + 15) & ~15, ldq = N16 + 2;
    int o = 0;
    for (int j = 0; j < a.J; j++) {
        const int P = a.c[j].P, P16 = (P + 15) & ~15, lda = P16 + 2, nt = P16 / 16;
        if (P16 > 256 || N16 > 256) return false;
        for (int tj = 0; tj < nt; tj++)
            for (int ti = tj; ti < nt; ti++)
                v.push_back(stage_rect(S_src[j] + ti * 16 + (long long)tj * 16 * P, P, std::min(16, P - ti * 16), std::min(16, P - tj * 16), o + ti * 16 + tj * 16 * lda, lda));
        const int oZ = o + lda * P16 + P16;
        for (int tc = 0; tc < N16 / 16; tc++)
            for (int ti = 0; ti < nt; ti++)
                v.push_back(stage_rect(B_src[j] + ti * 16 + (long long)tc * 16 * a.ldb, a.ldb, std::min(16, P - ti * 16), std::min(16, N - tc * 16), oZ + ti * 16 + tc * 16 * lda, lda));
        o += lda * P16 + P16 + lda * N16 + ldq * N16;
    }
    if (v.empty() || v.size() > (size_t)STAGE_MAX_JOBS) return false;
    std::memset(&jb, 0, sizeof(jb));
    for (size_t i = 0; i < v.size(); i++) jb.j[i] = v[i];
    jb.n = (int)v.size();
    for (int i = jb.n; i < STAGE_MAX_JOBS; i++) jb.j[i] = stage_rect(S_src[0], 16, 0, 0, 0, 16, 0);
    return true;
}

template <int NJ>
__global__ __launch_bounds__(256) void k_factor_small(const FSmallArgs a, const StageJobs jobs, int *__restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ int offA[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int J = a.J, N = a.N, N16 = (N + 15) & ~15, ldq = N16 + 2;
    stage_all<NJ>(jobs, lds, tid);
    int o = 0;
    for (int j = 0; j < J; j++) {
        const int P16 = (a.c[j].P + 15) & ~15, lda = P16 + 2;
        if (tid == 0) offA[j] = o;
        o += lda * P16 + P16 + lda * N16 + ldq * N16;
    }
    double *AQ = lds + o, *dvQ = AQ + ldq * N16;
    __syncthreads();
    // identity in the padding of the diagonal (what lds_potrf expects of rows / columns n .. n16 - 1)
    for (int j = 0; j < J; j++) {
        const int P = a.c[j].P, P16 = (P + 15) & ~15, lda = P16 + 2;
        if (tid >= P && tid < P16) lds[offA[j] + tid + tid * lda] = 1.0;
    }
    __syncthreads();
    // ---- per cluster: L_j = chol(S_j), LinvB_j = L_j^-1 B_j, partial Q_j (lower tiles) ----
    auto cluster = [&](int j, auto wg, int w, int nw) {
        constexpr bool WG = decltype(wg)::value;
        const int P = a.c[j].P, P16 = (P + 15) & ~15, lda = P16 + 2;
        double *A = lds + offA[j], *dv = A + lda * P16, *Z = dv + P16, *Qs = Z + lda * N16;
        const bool bad = lds_potrf<WG>(A, lda, dv, P, w, nw, lane);
        if (bad && lane == 0) atomicMin(info, a.c[j].code);
        if (N > 0) {
            lds_sync<WG>();
            lds_trsm<false, WG>(A, lda, dv, Z, 1, lda, P, N, w, nw, lane);
            lds_sync<WG>();
            lds_gemm_tn(Z, lda, Z, lda, Qs, ldq, N, N, P, w, nw, lane, true);
        }
    };
    if (J == 1) cluster(0, std::true_type{}, wave, 4);
    else if (wave < J) cluster(wave, std::false_type{}, 0, 1);
    __syncthreads();
    // ---- results of the clusters to memory (all threads), Q = sum_j Q_j into LDS (identity padded) ----
    const int i16 = tid & 15, j16 = tid >> 4;
    for (int j = 0; j < J; j++) {
        const FSmallCluster cj = a.c[j];
        const int P = cj.P, P16 = (P + 15) & ~15, lda = P16 + 2;
        const double *A = lds + offA[j], *dv = A + lda * P16, *Z = dv + P16;
        for (int j0 = 0; j0 < P; j0 += 16)
            for (int i0 = 0; i0 < P; i0 += 16) {
                const int i = i0 + i16, k = j0 + j16;
                if (i < P && k < P) cj.S[i + (long long)k * P] = (i >= k) ? A[i + k * lda] : 0.0;
            }
        if (tid < P) cj.dinv[tid] = dv[tid];
        for (int j0 = 0; j0 < N; j0 += 16)
            for (int i0 = 0; i0 < P; i0 += 16) {
                const int i = i0 + i16, k = j0 + j16;
                if (i < P && k < N) cj.LB[i + (long long)k * a.ldb] = Z[i + k * lda];
            }
    }
    if (N == 0) return;
    for (int j0 = 0; j0 < N16; j0 += 16)
        for (int i0 = 0; i0 < N16; i0 += 16) {
            const int i = i0 + i16, k = j0 + j16;
            double v = (i == k) ? 1.0 : 0.0;
            if (i < N && k < N) {
                v = 0.0;
                if (i >= k)
                    for (int j = 0; j < J; j++) {                         // fixed order: deterministic
                        const int P16 = (a.c[j].P + 15) & ~15, lda = P16 + 2;
                        v += lds[offA[j] + lda * P16 + P16 + lda * N16 + i + k * ldq];
                    }
            }
            AQ[i + k * ldq] = v;
        }
    __syncthreads();
    // ---- L_Q = chol(Q) (src/solver.jl:1274) ----
    const bool badq = lds_potrf<true>(AQ, ldq, dvQ, N, wave, 4, lane);
    if (badq && lane == 0) atomicMin(info, a.codeQ);
    __syncthreads();
    for (int j0 = 0; j0 < N; j0 += 16)
        for (int i0 = 0; i0 < N; i0 += 16) {
            const int i = i0 + i16, k = j0 + j16;
            if (i < N && k < N) a.Q[i + (long long)k * N] = (i >= k) ? AQ[i + k * ldq] : 0.0;
        }
    if (tid < N) a.dinvQ[tid] = dvQ[tid];
}

}  // namespace clrs
