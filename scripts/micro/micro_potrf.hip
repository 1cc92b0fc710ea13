// Where the time of one workgroup-level multi-word factorisation (wg_potrf, clrs_mw_kernels.hip.h) goes: wall-clock stamps of
// thread 0 at the top of every elimination step and around the post-processing, for an n x n SPD matrix in LDS.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -DMW_STAMPS -o _build/micro_potrf micro_potrf.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define MW_STAMPS 1
__device__ unsigned long long *g_stamps;
#include "../../clusteredlowranksolver.jl_amd/csrc/clrs_mw_kernels.hip.h"

template <int K, bool INV, int NT>
__global__ __launch_bounds__(NT) void k(const double *A, double *out, int n, unsigned long long *stamps) {
    using namespace mwk;
    const int tid = threadIdx.x;
    lds_d *scr = MW_LDS, *M = scr + MW_POTRF_SCR(K, n), *W = M + (long)K * n * n, *rd = W + (INV ? (long)K * n * n : 0);
    if (tid == 0) g_stamps = stamps;
    for (int e = tid; e < n * n; e += NT)
        for (int l = 0; l < K; l++) M[(long)l * n * n + e] = l == 0 ? A[e] : 0.0;
    __syncthreads();
    if (tid == 0) stamps[0] = wall_clock64();
    bool ok = wg_potrf<K, INV, NT>(M, (long)n * n, n, n, rd, n, W, (long)n * n, n, scr, tid);
    if (tid == 0) stamps[127] = wall_clock64();
    for (int e = tid; e < n * n; e += NT) out[e] = ok ? (double)M[e] : -1.0;
}
template <int K, bool INV, int NT>
void run(int n) {
    std::vector<double> A((size_t)n * n);
    srand(1);
    std::vector<double> G((size_t)n * n);
    for (auto &g : G) g = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double s = i == j ? 1.0 : 0.0;
            for (int t = 0; t < n; t++) s += G[i + t * n] * G[j + t * n];
            A[i + j * n] = s;
        }
    double *dA, *dO;
    unsigned long long *dS;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dO, A.size() * 8); hipMalloc(&dS, 128 * 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    size_t lds = ((size_t)MW_POTRF_SCR(K, n) + (size_t)K * n * n * (INV ? 2 : 1) + (size_t)K * n) * 8;
    hipFuncSetAttribute((const void *)k<K, INV, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; rep++) {
        hipMemset(dS, 0, 128 * 8);
        hipLaunchKernelGGL((k<K, INV, NT>), dim3(1), dim3(NT), lds, 0, dA, dO, n, dS);
        hipDeviceSynchronize();
    }
    unsigned long long st[128];
    hipMemcpy(st, dS, sizeof(st), hipMemcpyDeviceToHost);
    std::vector<double> O(A.size());
    hipMemcpy(O.data(), dO, A.size() * 8, hipMemcpyDeviceToHost);
    const double tick = 10.0;   // wall_clock64: 100 MHz
    printf("K=%d INV=%d NT=%d n=%d: total %.1f us, L[0,0]=%.6f (sqrt A00 = %.6f)\n  per step (ns):", K, (int)INV, NT, n, (st[127] - st[0]) * tick / 1e3, O[0], __builtin_sqrt(A[0]));
    for (int kk = 1; kk <= n; kk++) printf(" %.0f", (st[kk] - st[kk - 1]) * tick);
    printf("\n  post: pivots %.0f ns, scaling %.0f ns\n", (st[100] - st[n]) * tick, (st[101] - st[100]) * tick);
}
int main() {
    run<5, true, 512>(31); run<5, true, 1024>(31); run<5, true, 256>(31); run<5, false, 512>(16); run<5, false, 256>(16); run<4, true, 512>(31);
    return 0;
}
