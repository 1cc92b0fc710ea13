// Chip-wide rate of v_mfma_f64_16x16x4_f64 on MI355X, alone and beside fp64 VALU work.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/micro/micro_mfma64 scripts/micro/micro_mfma64.hip   (build on the CPU box)
//   ./scripts/micro/micro_mfma64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

// MODE 0: NCH independent accumulator chains of MFMAs; MODE 1: the same with VALU FMAs interleaved (VPER per MFMA); MODE 2: VALU FMAs only
template <int NCH, int VPER, int MODE>
__global__ __launch_bounds__(256) void k_rate(const double *in, double *out, int iters, unsigned long long *cyc) {
    const int lane = threadIdx.x & 63;
    const double a = in[lane], b = in[64 + lane];
    v4d acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) acc[c] = (v4d){0.0, 0.0, 0.0, 0.0};
    double f[8];
#pragma unroll
    for (int i = 0; i < 8; i++) f[i] = in[128 + lane + i];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                if (MODE != 2) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
                if (MODE != 0) {
#pragma unroll
                    for (int v = 0; v < VPER; v++) f[(c * VPER + v) & 7] = __builtin_fma(f[(c * VPER + v) & 7], a, b);
                }
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < NCH; c++) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
#pragma unroll
    for (int i = 0; i < 8; i++) s += f[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NCH, int VPER, int MODE>
static void run(const char *name, int wgs, double *din, double *dout, unsigned long long *dcyc) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k_rate<NCH, VPER, MODE>), dim3(wgs), dim3(256), 0, 0, din, dout, iters, dcyc);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_rate<NCH, VPER, MODE>), dim3(wgs), dim3(256), 0, 0, din, dout, iters, dcyc);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long cyc = 0;
    hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
    const double nm = (MODE == 2) ? 0.0 : (double)wgs * 4 * iters * 4 * NCH, nv = (MODE == 0) ? 0.0 : (double)wgs * 4 * iters * 4 * NCH * VPER;
    printf("%-44s wgs %4d: %8.1f us  MFMA %6.2f TF  VALU-fma %6.2f TF  | wave 0: %.1f cycles per (MFMA + %d fma) [s_memtime]\n", name, wgs, ms * 1e3,
           nm * 2048 / (ms * 1e-3) / 1e12, nv * 128 / (ms * 1e-3) / 1e12, (double)cyc / (iters * 4.0 * NCH), MODE == 0 ? 0 : VPER);
}

int main() {
    std::vector<double> h(1024);
    for (size_t i = 0; i < h.size(); i++) h[i] = 0.5 + 1e-3 * (double)((i * 2654435761u) % 1000);
    double *din, *dout;
    unsigned long long *dcyc;
    hipMalloc(&din, h.size() * 8); hipMalloc(&dout, 8 * 256 * 2048); hipMalloc(&dcyc, 64);
    hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    for (int wgs : {256, 512, 1024}) {
        run<4, 0, 0>("MFMA only, 4 chains", wgs, din, dout, dcyc);
        run<1, 0, 0>("MFMA only, 1 chain (dependent)", wgs, din, dout, dcyc);
        run<4, 4, 1>("MFMA + 4 fma each", wgs, din, dout, dcyc);
        run<4, 8, 1>("MFMA + 8 fma each", wgs, din, dout, dcyc);
        run<4, 16, 1>("MFMA + 16 fma each", wgs, din, dout, dcyc);
        run<4, 8, 2>("fma only (8 per slot)", wgs, din, dout, dcyc);
    }
    return 0;
}
