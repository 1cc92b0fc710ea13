// One-way latency of a flag hand-off between two workgroups, by the cache-control bits of the store and of the polling load (gfx950: sc0 / sc1), for a
// pair of workgroups on the SAME XCD (workgroup ids 0 and 8 of a 16-workgroup launch) and on DIFFERENT XCDs (ids 0 and 1).  The pipelined factorisations
// (clrs_mw_pipe.hip.h) hand pivot columns over with agent-scope (sc1) stores and loads: 2.5-4 us per hop.  Question: is there a cheaper hand-off
// through the L2 of one XCD?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -o scripts/micro/micro_hop scripts/micro/micro_hop.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int F> __device__ __forceinline__ void st(unsigned long long *p, unsigned long long v) {
    if (F == 0) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (F == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    if (F == 2) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if (F == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
template <int F> __device__ __forceinline__ unsigned long long ld(const unsigned long long *p) {
    unsigned long long v;
    if (F == 0) asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (F == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (F == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (F == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// flags[0]: A -> B, flags[32]: B -> A (different cache lines); out: {ticks of 100 MHz, failures, xcc of A, xcc of B}
template <int FS, int FL>
__global__ void k_pingpong(unsigned long long *flags, unsigned long long *out, int a_id, int b_id, int trips, unsigned long long base) {
    if (threadIdx.x != 0) return;
    const int me = blockIdx.x;
    if (me != a_id && me != b_id) return;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const bool isA = me == a_id;
    out[isA ? 2 : 3] = xcc & 0xf;
    unsigned long long fails = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 1; i <= trips; i++) {
        const unsigned long long seq = base + i;
        if (isA) {
            st<FS>(flags, seq);
            int spins = 0;
            while (ld<FL>(flags + 32) != seq && ++spins < 200000) { }
            if (spins >= 200000) { fails++; break; }
        } else {
            int spins = 0;
            while (ld<FL>(flags) != seq && ++spins < 200000) { }
            if (spins >= 200000) { fails++; st<FS>(flags + 32, seq); break; }
            st<FS>(flags + 32, seq);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (isA) { out[0] = t1 - t0; out[1] = fails; } else out[4] = fails;
}
static unsigned long long g_base = 0;
template <int FS, int FL>
static void run(unsigned long long *flags, unsigned long long *out, int a, int b, const char *what) {
    const int trips = 2000;
    unsigned long long h[8] = {0};
    (void)hipMemset(out, 0, 64);
    hipLaunchKernelGGL((k_pingpong<FS, FL>), dim3(16), dim3(64), 0, 0, flags, out, a, b, trips, g_base);
    g_base += 1000000;
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    static const char *nm[4] = {"plain", "sc0", "sc1", "sc0 sc1"};
    printf("%-16s store %-8s load %-8s XCC %llu -> %llu : %s%7.0f ns per one-way hop\n", what, nm[FS], nm[FL], h[2], h[3], (h[1] || h[4]) ? "NEVER SEEN (gave up) " : "",
           (h[1] || h[4]) ? 0.0 : 10.0 * (double)h[0] / (2.0 * trips));
}
template <int FS>
static void runs(unsigned long long *flags, unsigned long long *out, int a, int b, const char *what) {
    run<FS, 0>(flags, out, a, b, what); run<FS, 1>(flags, out, a, b, what); run<FS, 2>(flags, out, a, b, what); run<FS, 3>(flags, out, a, b, what);
}
int main() {
    unsigned long long *flags, *out;
    (void)hipMalloc(&flags, 8 * 64); (void)hipMalloc(&out, 64);
    (void)hipMemset(flags, 0, 8 * 64);
    for (int pair = 0; pair < 2; pair++) {
        const int a = 0, b = pair == 0 ? 8 : 1;
        const char *what = pair == 0 ? "same XCD (0, 8)" : "other XCD (0, 1)";
        runs<0>(flags, out, a, b, what); runs<1>(flags, out, a, b, what); runs<2>(flags, out, a, b, what); runs<3>(flags, out, a, b, what);
    }
    return 0;
}
