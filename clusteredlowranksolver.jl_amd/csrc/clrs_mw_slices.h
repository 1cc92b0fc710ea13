// clrs_mw_slices.h -- the two conversions of the exact-product scheme (clrs_mw_exact.hip.h), host + device:
//   mws_slice      a K-limb number -> S digits of MWS_BETA bits relative to a window exponent,
//   mws_recombine  the S order sums of a slice product (exact integers in fp64) -> a K-limb number.
// Both are on the critical path of k_mws_pair between its MFMA phases (7.8 M MFMA among 79.6 M wave instructions in round 3), so they are written for
// instruction count: every step is a rounding to a fixed grid by add-and-subtract of a constant and plain, provably exact adds -- no two_sum cascades.
// Compiled for the host by tests/mw_host (against mpmath) and for the device by clrs_mw_exact.hip.h.
#ifndef CLRS_MW_SLICES_H
#define CLRS_MW_SLICES_H

#include "clrs_mw_arith.h"

#define MWS_BETA 23
constexpr int mws_slices(int K) { return (52 * K + 16 + MWS_BETA - 1) / MWS_BETA; }

namespace mwk {

// 2^n for a compile-time (after unrolling) n, as a double built from its exponent field: no call to ldexp in unrolled code
MWF double mws_pow2(int n) { return __builtin_ldexp(1.0, n); }

// exponent e with |x| < 2^(e-2) for a renormalised expansion with head h (0 for h = 0: the slices of a zero are zeros)
MWF int mws_exponent(double h) {
    if (h == 0.0) return 0;
    int ex;
    (void)__builtin_frexp(h, &ex);          // |h| < 2^ex
    return ex + 2;
}

// Digits of x relative to the window exponent e (|x| < 2^(e-2)): x = 2^e sum_s d[s] 2^-(s+1)B + O(K 2^(e-SB-1)), d[s] integers, |d[s]| <= 2^(B-1) (+1 for d[0]).
//
// Limb by limb: the part of a limb that lies on the grid 2^-(s+1)B of slice s is its rounding to that grid by add-and-subtract of 1.5 * 2^52 * grid (exact
// while |r| < 2^51 grid), the remainder goes on to the next slice; both steps are exact, whatever the limb holds.  The contributions of the limbs to a
// slice add up as small integers (exact), and one pass from the last slice to the first moves what exceeds half a grid step of the slice above into
// that slice (a digit holds the tail of one limb and the head of the next: up to 1.5 * 2^(B-1) before the pass, and the accumulation bound of the slice
// products is stated for 2^(B-1) + 1).  A limb l of a renormalised x is below 2^(e - 2 - 51 l) (renorm leaves |x_l+1| <= 2^-51 |x_l|): it cannot reach the
// slices above s0(l) = ceil((51 l + 1) / B) - 1, which are skipped at compile time -- 40 of the 60 (limb, slice) pairs remain at K = 5.  Limbs up to
// 2^28 above that bound (input that was never renormalised) are still sliced exactly: the first step of a limb only needs |x_l| < 2^51 grid, which the
// choice of s0 leaves 29 bits of room for, and the pass over the digits takes carries of any size.  (The first version rounded the head of the whole remainder and swept it with K - 1 two_sums per slice:
// 353 fp64 instructions per number at K = 5, against 225 here.)
template <int K, int S, class OUT>
MWF void mws_slice(const mwa::mw<K> &x, int e, OUT &&put) {
    double d[S];
#pragma unroll
    for (int s = 0; s < S; s++) d[s] = -0.0;                            // (-0 + t = t for every t: the first accumulation folds to a multiply)
#pragma unroll
    for (int l = 0; l < K; l++) {
        double r = __builtin_ldexp(x.l[l], -e);
        const int s0 = l == 0 ? 0 : (51 * l + 1 + MWS_BETA - 1) / MWS_BETA - 1;
#pragma unroll
        for (int s = 0; s < S; s++) {
            if (s < s0) continue;
            const double C = 0x1.8p52 * mws_pow2(-(s + 1) * MWS_BETA);
            const double t = (r + C) - C;
            r -= t;
            d[s] = mwa::fma_(t, mws_pow2((s + 1) * MWS_BETA), d[s]);     // in units of the slice's grid: an integer
        }
    }
    const double C2 = 0x1.8p52 * mws_pow2(MWS_BETA);
#pragma unroll
    for (int s = S - 1; s >= 1; s--) {
        const double c = (d[s] + C2) - C2;                               // the multiple of 2^B nearest to d[s]
        d[s] -= c;
        d[s - 1] = mwa::fma_(c, mws_pow2(-MWS_BETA), d[s - 1]);
        put(s, (float)d[s]);
    }
    put(0, (float)d[0]);
}

// sum_o a[o] 2^-(o+2)B as K limbs, times 2^escale; a[o] exact integers, |a[o]| < 2^53.6 (the order sums of a slice product).
//
// Bins of 2B = 46 bits: bin j collects what lies between 2^-2Bj and 2^-2B(j+1).  An order of even index 2j has its last bit on the last bit of bin j and
// reaches at most 8 bits into bin j-1; an order of odd index 2j+1 ends B bits below bin j: one rounding to the grid of the bin boundary (add-and-subtract,
// exact) cuts either into its two parts.  A bin receives four parts, below 2^45, 2^45, 2^30 and 2^8 of its last bit: their plain fp64 sum is exact, so the
// S orders become ceil(S / 2) numbers that overlap by two bits at most, in 58 instructions at S = 12 -- against 180 for S pushes into an accumulator of
// two_sum cascades.  One robust renormalisation (clrs_mw_arith.h: a triangular sweep of two_sums, repeated only if the leading bins cancel) turns them
// into limbs.  The last odd order has no bin for its lowest 23 bits: it is added to the last bin as it is (rounded at 2^-52 of that bin: 2^-(SB+5) of the window, where
// the orders >= S were dropped at 2^-SB).
constexpr int mws_bins(int S) { return (S + 1) / 2; }
template <int S>
MWF void mws_bins_zero(double (&bin)[mws_bins(S)]) {
#pragma unroll
    for (int j = 0; j < mws_bins(S); j++) bin[j] = -0.0;                // (-0 + t = t for every t: the first add folds away)
}
// bin += the S order sums a[] -- exact while the bins stay below 2^53 of their last bit: up to 64 sets of order sums (the 32-row chunks of a long
// contraction index, k_mwx_gram) may be added into the same bins before mws_bins_result
template <int S>
MWF void mws_bins_add(double (&bin)[mws_bins(S)], const double (&a)[S]) {
    constexpr int NB = mws_bins(S), W = 2 * MWS_BETA;
    bin[0] += a[0] * mws_pow2(-W);
#pragma unroll
    for (int o = 1; o < S; o++) {
        const double v = a[o] * mws_pow2(-(o + 2) * MWS_BETA);
        const int j = o / 2;                                            // even o = 2j: parts to bins j-1 | j; odd o = 2j+1: parts to bins j | j+1
        const int hi_bin = (o & 1) ? j : j - 1;
        if (hi_bin + 1 >= NB) { bin[hi_bin] += v; continue; }           // (the last odd order: no bin below)
        const double C = 0x1.8p52 * mws_pow2(-W * (hi_bin + 1));        // grid = last bit of bin hi_bin
        const double hi = (v + C) - C;
        bin[hi_bin] += hi;
        bin[hi_bin + 1] += v - hi;
    }
}
template <int K, int S>
MWF mwa::mw<K> mws_bins_result(double (&bin)[mws_bins(S)], int escale) {
    constexpr int NB = mws_bins(S);
    static_assert(NB >= K, "bins of 46 bits must cover the limbs");
    mwa::renorm<NB>(bin);
    mwa::mw<K> r;
#pragma unroll
    for (int l = 0; l < K; l++) r.l[l] = __builtin_ldexp(bin[l], escale);
    return r;
}
template <int K, int S>
MWF mwa::mw<K> mws_recombine_orders(const double (&a)[S], int escale) {
    double bin[mws_bins(S)];
    mws_bins_zero<S>(bin);
    mws_bins_add<S>(bin, a);
    return mws_bins_result<K, S>(bin, escale);
}

}  // namespace mwk

#endif
