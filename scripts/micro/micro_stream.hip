// What a pure streaming kernel reaches on MI355X with the read : write mix of the Schur assembly (2 bytes read per byte written),
// against the footprint (beside / beyond the 256 MiB Infinity Cache): the practical roof the assembly kernel is compared with.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/micro/micro_stream scripts/micro/micro_stream.hip   (build on the CPU box)
//   ./scripts/micro/micro_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));

// persistent grid; every wave moves 1 KB per instruction (16 bytes per lane); reads 2 streams, writes 1
template <bool NT>
__global__ __launch_bounds__(256) void k_stream(const v2d *__restrict__ a, const v2d *__restrict__ b, v2d *__restrict__ c, long long n) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += 4 * stride) {
        v2d x[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const long long k = i + u * stride;
            if (k < n) { x[u] = a[k]; y[u] = b[k]; }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const long long k = i + u * stride;
            if (k < n) {
                const v2d r = x[u] + y[u];
                if (NT) __builtin_nontemporal_store(r, c + k); else c[k] = r;
            }
        }
    }
}

int main() {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mb : {32, 64, 128, 180, 256, 512, 768, 1024, 2048}) {          // MB per stream; footprint = 3 x
        const long long n = (long long)mb * 1024 * 1024 / 16;
        v2d *a, *b, *c;
        if (hipMalloc(&a, n * 16) != hipSuccess || hipMalloc(&b, n * 16) != hipSuccess || hipMalloc(&c, n * 16) != hipSuccess) return 1;
        hipMemset(a, 0, n * 16); hipMemset(b, 0, n * 16); hipMemset(c, 0, n * 16);
        for (int nt = 0; nt < 2; nt++)
            for (int wgs : {512, 2048}) {
                const int reps = 20;
                for (int r = 0; r < 3; r++) {
                    if (nt) hipLaunchKernelGGL(k_stream<true>, dim3(wgs), dim3(256), 0, 0, a, b, c, n);
                    else hipLaunchKernelGGL(k_stream<false>, dim3(wgs), dim3(256), 0, 0, a, b, c, n);
                }
                hipEventRecord(e0, 0);
                for (int r = 0; r < reps; r++) {
                    if (nt) hipLaunchKernelGGL(k_stream<true>, dim3(wgs), dim3(256), 0, 0, a, b, c, n);
                    else hipLaunchKernelGGL(k_stream<false>, dim3(wgs), dim3(256), 0, 0, a, b, c, n);
                }
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                const double us = 1e3 * ms / reps;
                std::printf("footprint %5d MB (read 2 x %d, write %d) wgs %4d %s: %8.1f us  %6.0f GB/s\n", 3 * mb, mb, mb, wgs, nt ? "nt-store" : "store   ", us,
                            3.0 * mb * 1.048576e6 / (us * 1e-6) / 1e9);
            }
        hipFree(a); hipFree(b); hipFree(c);
    }
    return 0;
}
