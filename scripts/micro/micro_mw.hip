// Latency of the multi-word primitives on one wave: N dependent operations per thread, one workgroup of 64 threads.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o micro_mw micro_mw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../clusteredlowranksolver.jl_amd/csrc/clrs_mw_arith.h"
using namespace mwa;
template <int K, int OP>
__global__ void k(double *out, int n, double seed) {
    mw<K> a = from_double<K>(1.0 + seed * (threadIdx.x + 1) * 1e-3), b = from_double<K>(1.0 + 1e-7 * threadIdx.x), c = from_double<K>(0.5);
    a.l[1] = 1e-17 * a.l[0]; b.l[1] = -3e-18;
    for (int i = 0; i < n; i++) {
        if (OP == 0) a = mul<K>(a, b);
        if (OP == 1) a = fnma<K>(a, b, c);
        if (OP == 2) a = add<K>(a, b);
        if (OP == 3) { a = rsqrt<K>(a); a.l[0] += 1.0; }
        if (OP == 4) { a = recip<K>(a); a.l[0] += 1.0; }
        if (OP == 5) { acc<K> s; acc_zero<K>(s); for (int j = 0; j < 16; j++) { acc_fma<K, K, K>(s, a, b); b.l[0] += 1e-9; } a = acc_result<K>(s); a = mul_pow2<K>(a, 1.0 / 16); }
        if (OP == 6) a = mul_d<K>(a, 1.0000001);
        if (OP == 7) { a = div_hr<K>(a, b, recip<(K + 1) / 2>(cvt<(K + 1) / 2, K>(b))); b.l[0] += 1e-9; }     // shared-divisor division with its half-precision reciprocal
        if (OP == 8) { mw<(K + 1) / 2> x = recip<(K + 1) / 2>(cvt<(K + 1) / 2, K>(a)); a.l[0] = x.l[0] + 1.0; a.l[1] = x.l[1] * 1e-3; }
        if (OP == 9) { mw<(K + 1) / 2> x; x.l[0] = 1.0 / b.l[0]; for (int l = 1; l < (K + 1) / 2; l++) x.l[l] = 1e-17 * x.l[l - 1]; a = div_hr<K>(a, b, x); }   // div_hr alone
    }
    double s = 0;
    for (int l = 0; l < K; l++) s += a.l[l];
    out[threadIdx.x + blockIdx.x * blockDim.x] = s;
}
template <int K, int OP>
void run(const char *name, int n, int per) {
    double *d; hipMalloc(&d, 8 * 1024 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves = 1; waves <= 8; waves *= 2) {
        hipLaunchKernelGGL((k<K, OP>), dim3(1), dim3(64 * waves), 0, 0, d, 10, 1.0);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<K, OP>), dim3(1), dim3(64 * waves), 0, 0, d, n, 1.0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("K=%d %-10s waves/CU=%d: %8.1f ns per op (%6.0f cycles at 2.4 GHz)\n", K, name, waves, 1e6 * ms / n / per, 2.4e3 * ms / n / per * 1e3 / 1e3);
    }
    hipFree(d);
}
int main() {
    run<5, 0>("mul", 2000, 1); run<5, 1>("fnma", 2000, 1); run<5, 2>("add", 2000, 1); run<5, 6>("mul_d", 2000, 1); run<5, 3>("rsqrt", 500, 1); run<5, 4>("recip", 500, 1); run<5, 5>("dot16/term", 200, 16);
    run<5, 7>("div_fast", 500, 1); run<5, 8>("recip<KH>", 500, 1); run<5, 9>("div_hr", 500, 1);
    run<4, 0>("mul", 2000, 1); run<4, 1>("fnma", 2000, 1); run<4, 3>("rsqrt", 500, 1); run<4, 5>("dot16/term", 200, 16); run<4, 7>("div_fast", 500, 1); run<4, 9>("div_hr", 500, 1);
    return 0;
}
