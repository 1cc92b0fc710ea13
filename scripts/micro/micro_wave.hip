// Micro-benchmark of the wave-level primitives of clrs_wave.hip.h: cycles (s_memtime) per call, one wave.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/micro/micro_wave scripts/micro/micro_wave.hip   (build on the CPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../clusteredlowranksolver.jl_amd/csrc/clrs_wave.hip.h"
using namespace clrs;

__global__ void k_micro(double *A, unsigned long long *out, int reps) {
    __shared__ double Ls[18 * 16];
    __shared__ double Zs[18 * 64];
    __shared__ double dinv[16];
    const int lane = threadIdx.x & 63, row16 = lane & 15;
    for (int e = threadIdx.x; e < 18 * 16; e += blockDim.x) Ls[e] = 0.0;
    for (int e = threadIdx.x; e < 18 * 64; e += blockDim.x) Zs[e] = 1.0 + (e % 7);
    __syncthreads();
    double a[16];
    for (int c = 0; c < 16; c++) a[c] = (c <= row16) ? A[row16 + c * 16] : 0.0;
    unsigned long long t0, t1;
    // ---- potrf16 ----
    bool bad = false;
    t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
        double b[16];
        for (int c = 0; c < 16; c++) b[c] = a[c] + 1e-9 * r;
        potrf16(b, row16, 16, bad);
        if (r == reps - 1) for (int c = 0; c < 16; c++) Ls[row16 + c * 18] = b[c];
        asm volatile("" ::"v"(b[15]));
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = (t1 - t0) / reps;
    if (lane < 16) dinv[lane] = 1.0 / Ls[lane * 19];
    __syncthreads();
    // ---- trsm16: one pass (two column groups interleaved) ----
    double Lr[16];
    for (int k = 0; k < 16; k++) Lr[k] = Ls[row16 + k * 18];
    const double di = dinv[row16];
    double x0 = Zs[row16 + (lane >> 4) * 18], x1 = Zs[row16 + (4 + (lane >> 4)) * 18];
    t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
        Trsm16<0>::run(x0, x1, Lr, di);
        x0 += 1.0; x1 += 1.0;
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[1] = (t1 - t0) / reps;
    Zs[row16 + (lane >> 4) * 18] = x0 + x1;
    // ---- 4 dependent MFMAs with LDS operand reads (one 16x16x16 tile) ----
    v4d_f acc = {0, 0, 0, 0};
    t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
        const double *ap = Zs + (lane >> 4) + (lane & 15) * 18;
        for (int k = 0; k < 16; k += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[k], ap[k + 18 * 16], acc, 0, 0, 0);
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[2] = (t1 - t0) / reps;
    // ---- 16 independent MFMAs back to back (issue rate) ----
    v4d_f ac2[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const double u = Zs[lane], w = Zs[lane + 64];
    t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int m = 0; m < 4; m++) ac2[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(u, w, ac2[m], 0, 0, 0);
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[3] = (t1 - t0) / reps / 16;
    // ---- dependent v_fma_f64 chain ----
    double f = u;
    t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int q = 0; q < 16; q++) f = __builtin_fma(f, w, u);
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[4] = (t1 - t0) * 100 / reps / 16;   // x100
    // ---- barrier cost (4 waves) ----
    t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) __syncthreads();
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[5] = (t1 - t0) / reps;
    A[threadIdx.x] = acc[0] + ac2[0][0] + ac2[1][1] + ac2[2][2] + ac2[3][3] + f + (bad ? 1.0 : 0.0);
}

int main() {
    std::vector<double> h(256 + 1024, 0.0);
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) h[i + j * 16] = (i == j ? 20.0 : 0.0) + 1.0 / (1 + i + j);
    double *d;
    unsigned long long *o, ho[8];
    hipMalloc(&d, h.size() * 8);
    hipMalloc(&o, 64);
    hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    for (int it = 0; it < 2; it++) {
        hipLaunchKernelGGL(k_micro, dim3(1), dim3(256), 0, 0, d, o, 200);
        hipDeviceSynchronize();
    }
    hipMemcpy(ho, o, 48, hipMemcpyDeviceToHost);
    printf("cycles: potrf16 %llu | trsm16 pass (2 groups) %llu | 4 dependent MFMA + LDS reads %llu | MFMA issue %llu | dependent fma x100 %llu | barrier %llu\n",
           ho[0], ho[1], ho[2], ho[3], ho[4], ho[5]);
    return 0;
}
