// Where the time of one "column" workgroup of k_chol_level (clrs_kernels.hip.h) goes: wall-clock stamps (100 MHz) of thread 0 of workgroup 1
// after the loads, the updates, the factorisation of the diagonal block with the panel solve, and the store.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DCL_STAMPS -o _build/micro_chol_level micro_chol_level.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
__device__ unsigned long long g_cl_stamps[8];
#include "../../clusteredlowranksolver.jl_amd/csrc/clrs_wave.hip.h"
#include "../../clusteredlowranksolver.jl_amd/csrc/clrs_kernels.hip.h"
using namespace clrs;
int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1025, np = (n + 63) / 64;
    std::vector<double> A((size_t)n * n);
    srand(1);
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) A[i + (size_t)j * n] = (i == j ? n : 0.0) + 0.5 * (rand() / (double)RAND_MAX - 0.5) * (i >= j ? 1 : 0);
    for (int j = 0; j < n; j++)
        for (int i = 0; i < j; i++) A[i + (size_t)j * n] = A[j + (size_t)i * n];
    double *dA, *dD;
    int *dinfo;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dD, (size_t)np * 4096 * 8); hipMalloc(&dinfo, 4);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k_chol_level, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chol_level_lds_bytes());
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int lvl = -1; lvl <= np - 2; lvl++) {
        std::vector<CholLevelWork> w;
        CholLevelJob J{dA, dD, n, n, lvl, 1, lvl + 2, 0};
        for (int i = lvl + 1; i < np; i++) w.push_back(CholLevelWork{0, 0, i, 0});
        const int ncol = (int)w.size(), rb = (lvl + 2) * 64;
        if (lvl >= 0 && rb < n) {
            const int nt = (n - rb + 127) / 128;
            for (int tj = 0; tj < nt; tj++)
                for (int ti = tj; ti < nt; ti++) w.push_back(CholLevelWork{0, 1, ti, tj});
        }
        CholLevelJob *dj; CholLevelWork *dw;
        hipMalloc(&dj, sizeof(J)); hipMalloc(&dw, w.size() * sizeof(CholLevelWork));
        hipMemcpy(dj, &J, sizeof(J), hipMemcpyHostToDevice);
        hipMemcpy(dw, w.data(), w.size() * sizeof(CholLevelWork), hipMemcpyHostToDevice);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_chol_level, dim3((unsigned)w.size()), dim3(CL_NT), chol_level_lds_bytes(), 0, J, ncol, (const CholLevelJob *)nullptr, (const CholLevelWork *)nullptr, dinfo);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long st[8];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_cl_stamps), sizeof(st));
        if (lvl < 4 || lvl == np / 2)
            printf("level %2d: %3d column + %4d bulk workgroups, %.1f us;  workgroup 1: loads %.2f  updates %.2f  potrf + panel %.2f  store %.2f us\n", lvl, ncol,
                   (int)w.size() - ncol, 1e3 * ms, (st[1] - st[0]) / 100.0, (st[2] - st[1]) / 100.0, (st[3] - st[2]) / 100.0, (st[4] - st[3]) / 100.0);
    }
    return 0;
}
