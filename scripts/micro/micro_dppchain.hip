// Latency of the dependent v_fmac_f64_dpp elimination chain (clrs_solve_small.hip.h) against the three-instruction form
// (v_mul, v_mov_b64_dpp, v_fma: clrs_wave.hip.h Trsm16), one wave, s_memtime cycles.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -o scripts/micro/micro_dppchain scripts/micro/micro_dppchain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../clusteredlowranksolver.jl_amd/csrc/clrs_solve_small.hip.h"
using namespace clrs;

__global__ void k(const double *in, double *out, unsigned long long *cyc, int reps) {
    const int lane = threadIdx.x & 63, l15 = lane & 15;
    double m[16], Lr[16];
    for (int k2 = 0; k2 < 16; k2++) { m[k2] = (k2 < l15) ? in[lane + 64 * k2] * 1e-3 : 0.0; Lr[k2] = -m[k2]; }
    double x = in[lane], y0 = x, y1 = x + 1.0;
    const double di = 0.5;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) { trsv16_chain_fwd(x, m); x += 1.0; }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) { Trsm16<0>::run(y0, y1, Lr, di); y0 += 1.0; y1 += 1.0; }
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    // LDS read burst + multiply (the m[] set-up of wave_trsv_fwd)
    __shared__ double A[18 * 16];
    for (int e = threadIdx.x; e < 18 * 16; e += 64) A[e] = in[e];
    __syncthreads();
    double acc = 0.0;
    unsigned long long t3 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
        double mm[16];
#pragma unroll
        for (int k2 = 0; k2 < 15; k2++) mm[k2] = A[l15 + k2 * 18];
#pragma unroll
        for (int k2 = 0; k2 < 15; k2++) mm[k2] = (k2 < l15) ? -(mm[k2] * di) : 0.0;
#pragma unroll
        for (int k2 = 0; k2 < 15; k2++) acc += mm[k2];
        A[lane] = acc;
        wave_sync();
    }
    unsigned long long t4 = __builtin_amdgcn_s_memtime();
    out[lane] = x + y0 + y1 + acc;
    if (lane == 0) { cyc[0] = (t1 - t0) / reps; cyc[1] = (t2 - t1) / reps; cyc[2] = (t4 - t3) / reps; }
}
int main() {
    double *din, *dout; unsigned long long *dc, hc[3];
    hipMalloc(&din, 8 * 2048); hipMalloc(&dout, 8 * 64); hipMalloc(&dc, 64);
    double h[2048]; for (int i = 0; i < 2048; i++) h[i] = 0.25 + 1e-3 * (i % 97);
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    for (int it = 0; it < 2; it++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, dc, 200); hipDeviceSynchronize(); }
    hipMemcpy(hc, dc, 24, hipMemcpyDeviceToHost);
    printf("cycles: 15-step v_fmac_f64_dpp chain %llu | Trsm16 (two chains, mul + dpp mov + fma per step) %llu | 15 LDS reads + scale + sync %llu\n", hc[0], hc[1], hc[2]);
    return 0;
}
