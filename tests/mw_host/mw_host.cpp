// Host build of the product's multi-word arithmetic (csrc/clrs_mw_arith.h) for CPU-side unit tests:
// elementwise operations on K-limb planar arrays.  Test infrastructure; compiled by tests/test_mw_arith_cpu.py.
#include "../../clusteredlowranksolver.jl_amd/csrc/clrs_mw_arith.h"
#include "../../clusteredlowranksolver.jl_amd/csrc/clrs_mw_slices.h"
using namespace mwa;

template <int K>
static void run(int op, long n, const double *a, const double *b, double *c) {
    for (long i = 0; i < n; i++) {
        mw<K> x = ld<K>(a, n, i), y = ld<K>(b, n, i), r;
        switch (op) {
        case 0: r = add<K>(x, y); break;
        case 1: r = sub<K>(x, y); break;
        case 2: r = mul<K>(x, y); break;
        case 3: r = div<K>(x, y); break;
        case 4: r = mwa::sqrt<K>(x); break;
        case 5: r = recip<K>(x); break;
        case 6: r = rsqrt<K>(x); break;
        case 7: r = fnma<K>(x, y, y); break;            // x - y*y
        case 8: r = mul_d<K>(x, y.l[0]); break;
        case 9: r = div_fast<K>(x, y); break;
        default: r = zero<K>();
        }
        st<K>(c, n, i, r);
    }
}
// dot product of two planar vectors through the unnormalised accumulator
template <int K>
static void dot(long n, const double *a, const double *b, double *c) {
    acc<K> s;
    acc_zero<K>(s);
    for (long i = 0; i < n; i++) acc_fma<K, K, K>(s, ld<K>(a, n, i), ld<K>(b, n, i));
    st<K>(c, 1, 0, acc_result<K>(s));
}
extern "C" int mw_host_op(int K, int op, long n, const double *a, const double *b, double *c) {
    switch (K) {
    case 2: run<2>(op, n, a, b, c); return 0;
    case 3: run<3>(op, n, a, b, c); return 0;
    case 4: run<4>(op, n, a, b, c); return 0;
    case 5: run<5>(op, n, a, b, c); return 0;
    case 6: run<6>(op, n, a, b, c); return 0;
    case 8: run<8>(op, n, a, b, c); return 0;
    case 10: run<10>(op, n, a, b, c); return 0;
    }
    return -1;
}
extern "C" int mw_host_dot(int K, long n, const double *a, const double *b, double *c) {
    switch (K) {
    case 2: dot<2>(n, a, b, c); return 0;
    case 3: dot<3>(n, a, b, c); return 0;
    case 4: dot<4>(n, a, b, c); return 0;
    case 5: dot<5>(n, a, b, c); return 0;
    case 6: dot<6>(n, a, b, c); return 0;
    case 8: dot<8>(n, a, b, c); return 0;
    case 10: dot<10>(n, a, b, c); return 0;
    }
    return -1;
}

// the two conversions of the exact-product scheme (csrc/clrs_mw_slices.h): digits[s * n + i] of x_i relative to e[i]; limbs of sum_o a[o * n + i] 2^-(o+2)B
template <int K>
static void slice(long n, const double *x, const int *e, float *digits) {
    constexpr int S = mws_slices(K);
    for (long i = 0; i < n; i++) mwk::mws_slice<K, S>(ld<K>(x, n, i), e[i], [&](int s, float d) { digits[(long)s * n + i] = d; });
}
template <int K>
static void recombine(long n, const double *a, int escale, double *out) {
    constexpr int S = mws_slices(K);
    for (long i = 0; i < n; i++) {
        double o[S];
        for (int s = 0; s < S; s++) o[s] = a[(long)s * n + i];
        st<K>(out, n, i, mwk::mws_recombine_orders<K, S>(o, escale));
    }
}
extern "C" int mw_host_slices(int K) { return mws_slices(K); }
extern "C" int mw_host_exponent(double h) { return mwk::mws_exponent(h); }
extern "C" int mw_host_slice(int K, long n, const double *x, const int *e, float *digits) {
    switch (K) {
    case 2: slice<2>(n, x, e, digits); return 0;
    case 3: slice<3>(n, x, e, digits); return 0;
    case 4: slice<4>(n, x, e, digits); return 0;
    case 5: slice<5>(n, x, e, digits); return 0;
    case 6: slice<6>(n, x, e, digits); return 0;
    case 8: slice<8>(n, x, e, digits); return 0;
    case 10: slice<10>(n, x, e, digits); return 0;
    }
    return -1;
}
extern "C" int mw_host_recombine(int K, long n, const double *a, int escale, double *out) {
    switch (K) {
    case 2: recombine<2>(n, a, escale, out); return 0;
    case 3: recombine<3>(n, a, escale, out); return 0;
    case 4: recombine<4>(n, a, escale, out); return 0;
    case 5: recombine<5>(n, a, escale, out); return 0;
    case 6: recombine<6>(n, a, escale, out); return 0;
    case 8: recombine<8>(n, a, escale, out); return 0;
    case 10: recombine<10>(n, a, escale, out); return 0;
    }
    return -1;
}
