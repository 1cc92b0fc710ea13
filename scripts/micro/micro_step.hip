// Latency of ONE elimination step of the fraction-free Cholesky (clrs_mw_kernels.hip.h wg_potrf / clrs_mw_pipe.hip.h mwp_step) on one wave:
//     v <- renorm(dh v - ci cj)
// as the chain kernels run it (one entry per thread, the next step needs this one's result), in variants of the same arithmetic.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value -o scripts/micro/micro_step scripts/micro/micro_step.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../clusteredlowranksolver.jl_amd/csrc/clrs_mw_arith.h"
using namespace mwa;
template <int K, int V>
__device__ __forceinline__ mw<K> step(const mw<K> &dh, const mw<K> &v, const mw<K> &ci, const mw<K> &cj) {
    if (V == 0) {                      // as shipped: one accumulator, the two products one after the other
        acc<K> s; acc_zero<K>(s);
        acc_fma<K, K, K>(s, dh, v);
        acc_fma<K, K, K>(s, ci, cj, -1.0);
        return acc_result<K>(s);
    }
    if (V == 1) {                      // two accumulators, merged bin by bin
        acc<K> s, t; acc_zero<K>(s); acc_zero<K>(t);
        acc_fma<K, K, K>(s, dh, v);
        acc_fma<K, K, K>(t, ci, cj, -1.0);
        PushLimbs<K, K, 0>::run(s, t.s, 1.0);
        return acc_result<K>(s);
    }
    if (V == 2) {                      // the second product first (its operands are there before v is)
        acc<K> s; acc_zero<K>(s);
        acc_fma<K, K, K>(s, ci, cj, -1.0);
        acc_fma<K, K, K>(s, dh, v);
        return acc_result<K>(s);
    }
    return v;
}
template <int K, int V>
__global__ void k(double *out, unsigned long long *cyc, int n, double seed) {
    __shared__ double col[3 * 8 * 64];             // d, c_i, c_j of the step: read from LDS behind a barrier, as in the kernels (nothing can be hoisted out of the loop)
    mw<K> v = from_double<K>(1.0 + seed * (threadIdx.x + 1) * 1e-3);
    for (int l = 1; l < K; l++) v.l[l] = 1e-17 * v.l[l - 1];
    const int lane = threadIdx.x & 63;
    for (int e = threadIdx.x; e < 3 * 8 * 64; e += blockDim.x) {
        const int what = e / (8 * 64), l = (e / 64) % 8;
        double x = what == 0 ? 0.75 + 1e-7 * (e % 64) : what == 1 ? 0.5 : 0.5 + 1e-9 * (e % 64);
        for (int j = 0; j < l; j++) x *= (what == 0 ? -3e-18 : what == 1 ? 2e-17 : -1e-17);
        col[e] = x;
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
        mw<K> dh, ci, cj;
#pragma unroll
        for (int l = 0; l < K; l++) { dh.l[l] = col[(0 * 8 + l) * 64 + lane]; ci.l[l] = col[(1 * 8 + l) * 64 + lane]; cj.l[l] = col[(2 * 8 + l) * 64 + lane]; }
        v = step<K, V>(mul_pow2<K>(dh, 1.0), v, mul_pow2<K>(ci, 1.0), mul_pow2<K>(cj, 1.0));
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int l = 0; l < K; l++) s += v.l[l];
    out[threadIdx.x + blockIdx.x * blockDim.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int K, int V>
void run(const char *name) {
    double *d; unsigned long long *c, h;
    (void)hipMalloc(&d, 8 * 4096); (void)hipMalloc(&c, 64);
    for (int waves : {1, 2}) {
        hipLaunchKernelGGL((k<K, V>), dim3(1), dim3(256 * waves), 0, 0, d, c, 2000, 1.0);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        double out[4];
        (void)hipMemcpy(out, d, 32, hipMemcpyDeviceToHost);
        printf("K=%d %-44s %d wave(s) per SIMD: %6.0f cycles per step (%.2f us at 2.4 GHz)   [value %.17g]\n", K, name, waves, (double)h / 2000, (double)h / 2000 / 2400.0, out[1]);
    }
    (void)hipFree(d); (void)hipFree(c);
}
int main() {
    run<4, 0>("one accumulator (shipped)");
    run<4, 1>("two accumulators, merged");
    run<4, 2>("one accumulator, c_i c_j first");
    run<5, 0>("one accumulator (shipped)");
    run<5, 1>("two accumulators, merged");
    run<5, 2>("one accumulator, c_i c_j first");
    return 0;
}
