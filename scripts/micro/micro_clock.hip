// Core clock under a light load: one wave runs a dependent v_fma_f64 chain; s_memtime (core clock cycles) against s_memrealtime (100 MHz) gives the
// frequency the chain ran at, (a) in a lone small launch after an idle gap, (b) in back-to-back small launches (the interior-point iteration's
// regime: a few workgroups busy all the time), (c) with a chip-filling kernel running beside it on a second stream.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/micro/micro_clock scripts/micro/micro_clock.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
__global__ void k_chain(double *out, unsigned long long *st, int n, int waves_wanted) {
    double x = out[threadIdx.x & 63] + 1.0, y = 0.999999;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) x = __builtin_fma(x, y, 1e-9);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    out[threadIdx.x & 63] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) { st[0] = c1 - c0; st[1] = w1 - w0; }
}
__global__ void k_fill(double *out, int n) {
    double x = out[threadIdx.x & 63] + 1.0, y = 0.999999, z = x + 1.0;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) { x = __builtin_fma(x, y, 1e-9); z = __builtin_fma(z, y, 1e-9); }
    }
    if (x + z == 123.456) out[0] = x;
}
static void report(const char *what, unsigned long long *dst, int n) {
    unsigned long long h[2];
    hipMemcpy(h, dst, 16, hipMemcpyDeviceToHost);
    printf("%-64s %8.0f cycles per 16-fma chain step x %d: %6.2f cycles per dependent v_fma_f64, %7.1f us, core clock %.0f MHz\n", what, (double)h[0] / n, n, (double)h[0] / n / 16.0,
           h[1] / 100.0, 100.0 * (double)h[0] / (double)h[1]);
}
int main() {
    double *dout, *dout2; unsigned long long *dst;
    hipMalloc(&dout, 8 * 64); hipMalloc(&dout2, 8 * 64); hipMalloc(&dst, 64);
    hipMemset(dout, 0, 8 * 64); hipMemset(dout2, 0, 8 * 64);
    hipStream_t s1, s2;
    hipStreamCreate(&s1); hipStreamCreate(&s2);
    const int n = 2000;          // 32000 dependent FMAs: ~100 us
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, s1, dout, dst, n, 0); hipDeviceSynchronize();
    usleep(200000);
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, s1, dout, dst, n, 0); hipDeviceSynchronize();
    report("lone launch of one wave after 200 ms idle", dst, n);
    for (int waves : {1, 8, 64}) {
        for (int i = 0; i < 300; i++) hipLaunchKernelGGL(k_chain, dim3(waves), dim3(64), 0, s1, dout, dst, n, 0);
        hipDeviceSynchronize();
        char b[128]; snprintf(b, sizeof b, "300 back-to-back launches of %d workgroup(s) of one wave (~30 ms), last", waves);
        report(b, dst, n);
    }
    for (int i = 0; i < 300; i++) hipLaunchKernelGGL(k_chain, dim3(8), dim3(512), 0, s1, dout, dst, n, 0);
    hipDeviceSynchronize();
    report("300 back-to-back launches of 8 workgroups of 512 threads, last", dst, n);
    // a chip-filling kernel beside it
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, s2, dout2, 40000);
    usleep(20000);
    for (int i = 0; i < 100; i++) hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, s1, dout, dst, n, 0);
    hipStreamSynchronize(s1);
    report("beside a chip-filling fp64 kernel on a second stream, last of 100", dst, n);
    hipDeviceSynchronize();
    return 0;
}
