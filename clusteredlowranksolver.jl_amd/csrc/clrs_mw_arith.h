// clrs_mw_arith.h -- multi-word fp64 arithmetic of the extended-precision hot path (host + device).
//
// The reference computes in Arb midpoints at `prec` = 256 bits by default (src/solver.jl:73,103; every result is
// collapsed with get_mid!, products are Arblib.approx_mul!: src/tools.jl:59-107, src/solver.jl:1125-1143), because
// the sampled Schur complements of its headline problems are not positive definite to fp64 working accuracy
// (cohnelkies(8,15): lambda_min(S)/lambda_max(S) < 2^-53 at the first iterate; 113 bits fail too, 160 bits reach the
// pinned objective, 212 bits a gap of 1e-14, 256 bits the reference's default thresholds: DESIGN.md section 2).
// On gfx950 the fp64 FMA is a full-rate instruction and integer multiplies are not, so the multi-precision number
// here is an unevaluated sum of K doubles ("limbs", value = l[0] + l[1] + ... + l[K-1], |l[i+1]| <~ ulp(l[i])),
// K = 2..6 (106..318 bits), built from the error-free transformations two_sum / two_prod(fma).
//
// Rounding model: like Arb's approx_* functions these operations are not correctly rounded; each result carries a
// relative error of a few units of 2^(-53K+K) with respect to the magnitudes of the operands (additions: Cray-style,
// i.e. relative to |a| + |b|), which is what the backward-error analyses of GEMM / Cholesky / substitution need.
//
// Everything is a template over K and fully unrolled; no function here may be compiled with floating-point
// contraction (the translation unit is built with -ffp-contract=off): a*b + c fused behind the back of two_prod /
// two_sum breaks their exactness.
#ifndef CLRS_MW_ARITH_H
#define CLRS_MW_ARITH_H

#include <cmath>

#if defined(__HIPCC__) || defined(__HIP__)
#define MWF __host__ __device__ __forceinline__
#else
#define MWF inline __attribute__((always_inline))
#endif

namespace mwa {

template <int K>
struct mw {
    double l[K];
};

MWF double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// s + e = a + b exactly (Knuth), no assumption on magnitudes
MWF void two_sum(double a, double b, double &s, double &e) {
    s = a + b;
    double bb = s - a;
    e = (a - (s - bb)) + (b - bb);
}
// s + e = a + b exactly if |a| >= |b| (or a == 0)
MWF void fast_two_sum(double a, double b, double &s, double &e) {
    s = a + b;
    e = b - (s - a);
}
// p + e = a * b exactly
MWF void two_prod(double a, double b, double &p, double &e) {
    p = a * b;
    e = fma_(a, b, -p);
}

// ---------------------------------------------------------------------------------------------------------------------
// accumulator: K bins by order, NOT normalised.  push<O>(v) adds v into bin O with error-free carries into the higher
// orders; only the last bin rounds.  The exact value is the plain sum of the bins.
// ---------------------------------------------------------------------------------------------------------------------
template <int K>
struct acc {
    double s[K];
};

template <int K>
MWF void acc_zero(acc<K> &a) {
#pragma unroll
    for (int i = 0; i < K; i++) a.s[i] = 0.0;
}

template <int K, int O>
MWF void acc_push(acc<K> &a, double v) {
    if (O >= K) return;
#pragma unroll
    for (int n = O; n < K - 1; n++) {
        double s, e;
        two_sum(a.s[n], v, s, e);
        a.s[n] = s;
        v = e;
    }
    a.s[K - 1] += v;
}

// One renormalisation sweep over K terms: pass p leaves the (nearly) rounded sum of s[p..K-1] in s[p] and the exact
// remainders of the two_sums behind it.  The sum of the terms is preserved exactly.
template <int K>
MWF void renorm_sweep(double (&s)[K]) {
#pragma unroll
    for (int p = 0; p < K - 1; p++) {
        double t = s[K - 1];
#pragma unroll
        for (int i = K - 2; i >= p; i--) {
            double sum, e;
            two_sum(s[i], t, sum, e);
            t = sum;
            s[i + 1] = e;
        }
        s[p] = t;
    }
}
// true if some limb is not small against its predecessor (overlap, or a cancelled head with the value further down)
template <int K>
MWF bool renorm_bad(const double (&s)[K]) {
    bool bad = false;
#pragma unroll
    for (int i = 0; i < K - 1; i++) bad = bad || (__builtin_fabs(s[i + 1]) > 0x1p-51 * __builtin_fabs(s[i]));
    return bad;
}
// Robust renormalisation.  One sweep is enough unless leading terms cancel (1 - a x in a Newton step, a pivot
// s_kk - sum l^2 of an ill-conditioned matrix): a two_sum chain that cancels at the top leaves the value in the remainder
// positions, (0, x, y, ..).  The sweep is repeated while that is so (each repetition moves the head to the front and
// removes overlaps; rarely taken, at most K times), so that the head limb is always the value rounded to fp64 and the
// relative accuracy of a result never depends on how it was produced.
template <int K>
MWF void renorm(double (&s)[K]) {
    renorm_sweep<K>(s);
    for (int it = 0; it < K && renorm_bad<K>(s); it++) renorm_sweep<K>(s);
}

template <int K>
MWF mw<K> acc_result(const acc<K> &a) {
    double s[K];
#pragma unroll
    for (int i = 0; i < K; i++) s[i] = a.s[i];
    renorm<K>(s);
    mw<K> r;
#pragma unroll
    for (int i = 0; i < K; i++) r.l[i] = s[i];
    return r;
}

// --- helpers to unroll "for i, for j with i + j == n" at compile time ----------------------------------------------
template <int K, int KA, int KB, int I, int J>
struct PushProd {
    // adds a.l[I] * b.l[J] into the accumulator: exact (two_prod) while I + J < K - 1, plain in the last order
    static MWF void run(acc<K> &c, const double *a, const double *b, double sgn) {
        if (I + J < K - 1) {
            double p, e;
            two_prod(sgn * a[I], b[J], p, e);
            acc_push<K, I + J>(c, p);
            acc_push<K, I + J + 1>(c, e);
        } else if (I + J == K - 1) {
            c.s[K - 1] = fma_(sgn * a[I], b[J], c.s[K - 1]);
        }
    }
};
template <int K, int KA, int KB, int I, int J>
struct PushRow {
    static MWF void run(acc<K> &c, const double *a, const double *b, double sgn) {
        PushProd<K, KA, KB, I, J>::run(c, a, b, sgn);
        if constexpr (J + 1 < KB && I + J + 1 < K) PushRow<K, KA, KB, I, J + 1>::run(c, a, b, sgn);
    }
};
template <int K, int KA, int KB, int I>
struct PushAll {
    static MWF void run(acc<K> &c, const double *a, const double *b, double sgn) {
        PushRow<K, KA, KB, I, 0>::run(c, a, b, sgn);
        if constexpr (I + 1 < KA && I + 1 < K) PushAll<K, KA, KB, I + 1>::run(c, a, b, sgn);
    }
};

// c += sgn * a * b  (a: KA limbs, b: KB limbs), no renormalisation: the inner operation of every dot product
template <int K, int KA, int KB>
MWF void acc_fma(acc<K> &c, const mw<KA> &a, const mw<KB> &b, double sgn = 1.0) {
    PushAll<K, KA, KB, 0>::run(c, a.l, b.l, sgn);
}
template <int K, int KA>
MWF void acc_fma_d(acc<K> &c, const mw<KA> &a, double b, double sgn = 1.0) {
    PushAll<K, KA, 1, 0>::run(c, a.l, &b, sgn);
}
template <int K, int KA, int I>
struct PushLimbs {
    static MWF void run(acc<K> &c, const double *a, double sgn) {
        acc_push<K, I>(c, sgn * a[I]);
        if constexpr (I + 1 < KA && I + 1 < K) PushLimbs<K, KA, I + 1>::run(c, a, sgn);
    }
};
// c += sgn * a
template <int K, int KA>
MWF void acc_add(acc<K> &c, const mw<KA> &a, double sgn = 1.0) {
    PushLimbs<K, KA, 0>::run(c, a.l, sgn);
}
template <int K>
MWF void acc_add_d(acc<K> &c, double a) {
    acc_push<K, 0>(c, a);
}

// ---------------------------------------------------------------------------------------------------------------------
// the arithmetic proper
// ---------------------------------------------------------------------------------------------------------------------
template <int K>
MWF mw<K> from_double(double a) {
    mw<K> r;
    r.l[0] = a;
#pragma unroll
    for (int i = 1; i < K; i++) r.l[i] = 0.0;
    return r;
}
template <int K>
MWF mw<K> zero() { return from_double<K>(0.0); }

template <int K>
MWF mw<K> neg(const mw<K> &a) {
    mw<K> r;
#pragma unroll
    for (int i = 0; i < K; i++) r.l[i] = -a.l[i];
    return r;
}
// precision change: truncation of an expansion is a faithful rounding
template <int KO, int KI>
MWF mw<KO> cvt(const mw<KI> &a) {
    mw<KO> r;
#pragma unroll
    for (int i = 0; i < KO; i++) r.l[i] = i < KI ? a.l[i] : 0.0;
    return r;
}

template <int K>
MWF mw<K> add(const mw<K> &a, const mw<K> &b) {
    acc<K> c;
#pragma unroll
    for (int i = 0; i < K; i++) c.s[i] = a.l[i];
    acc_add<K, K>(c, b);
    return acc_result<K>(c);
}
template <int K>
MWF mw<K> sub(const mw<K> &a, const mw<K> &b) {
    acc<K> c;
#pragma unroll
    for (int i = 0; i < K; i++) c.s[i] = a.l[i];
    acc_add<K, K>(c, b, -1.0);
    return acc_result<K>(c);
}
template <int K>
MWF mw<K> add_d(const mw<K> &a, double b) {
    acc<K> c;
#pragma unroll
    for (int i = 0; i < K; i++) c.s[i] = a.l[i];
    acc_push<K, 0>(c, b);
    return acc_result<K>(c);
}
// KC-limb product of a KA-limb and a KB-limb number
template <int KC, int KA, int KB>
MWF mw<KC> mulx(const mw<KA> &a, const mw<KB> &b) {
    acc<KC> c;
    acc_zero<KC>(c);
    acc_fma<KC, KA, KB>(c, a, b);
    return acc_result<KC>(c);
}
template <int K>
MWF mw<K> mul(const mw<K> &a, const mw<K> &b) { return mulx<K, K, K>(a, b); }
template <int K>
MWF mw<K> mul_d(const mw<K> &a, double b) {
    acc<K> c;
    acc_zero<K>(c);
    acc_fma_d<K, K>(c, a, b);
    return acc_result<K>(c);
}
// exact scaling by a power of two
template <int K>
MWF mw<K> mul_pow2(const mw<K> &a, double p) {
    mw<K> r;
#pragma unroll
    for (int i = 0; i < K; i++) r.l[i] = a.l[i] * p;
    return r;
}
// a - b * c and a + b * c with one renormalisation
template <int K>
MWF mw<K> fnma(const mw<K> &a, const mw<K> &b, const mw<K> &c) {
    acc<K> r;
#pragma unroll
    for (int i = 0; i < K; i++) r.s[i] = a.l[i];
    acc_fma<K, K, K>(r, b, c, -1.0);
    return acc_result<K>(r);
}
template <int K>
MWF mw<K> fma(const mw<K> &a, const mw<K> &b, const mw<K> &c) {
    acc<K> r;
#pragma unroll
    for (int i = 0; i < K; i++) r.s[i] = a.l[i];
    acc_fma<K, K, K>(r, b, c, 1.0);
    return acc_result<K>(r);
}

template <int K>
MWF bool is_positive(const mw<K> &a) { return a.l[0] > 0.0; }   // the head carries the sign of a renormalised number
template <int K>
MWF double to_double(const mw<K> &a) { return a.l[0]; }
template <int K>
MWF mw<K> abs(const mw<K> &a) { return a.l[0] < 0.0 ? neg<K>(a) : a; }
template <int K>
MWF bool less(const mw<K> &a, const mw<K> &b) {           // a < b
    mw<K> d = sub<K>(a, b);
    return d.l[0] < 0.0;
}

// ---------------------------------------------------------------------------------------------------------------------
// reciprocal, reciprocal square root: Newton's iteration with the working precision doubled per step
// (53 -> 106 -> 212 -> 424 bits), each step evaluating its residual with just the limbs it needs.
// ---------------------------------------------------------------------------------------------------------------------
// precision schedule of the Newton iterations: the last step should start from ceil(K/2) limbs (not from the largest power of
// two below K), the one before from ceil(K/4), ...: for K = 5 the chain is 1 -> 2 -> 3 -> 5 instead of 1 -> 2 -> 4 -> 5, and
// the steps before the last run in cheaper arithmetic
constexpr int newton_next(int K, int KX) {
    int t = K;
    while ((t + 1) / 2 > KX) t = (t + 1) / 2;
    return t;
}

template <int K, int KX>
struct NewtonRecip {
    // x (KX limbs, accurate to ~53 KX bits) -> K limbs
    static MWF mw<K> run(const mw<K> &a, const mw<KX> &x) {
        if constexpr (KX >= K) {
            return cvt<K, KX>(x);
        } else {
            constexpr int KN = newton_next(K, KX);
            // r = 1 - a x  to KN limbs (its leading KX limbs cancel)
            acc<KN> c;
            acc_zero<KN>(c);
            c.s[0] = 1.0;
            acc_fma<KN, (KN < K ? KN : K), KX>(c, cvt<(KN < K ? KN : K), K>(a), x, -1.0);
            mw<KN> r = acc_result<KN>(c);
            // x' = x + x r    (x r only matters to KN - KX limbs)
            constexpr int KR = KN - KX;
            mw<KR> xr = mulx<KR, KX, KR>(x, cvt<KR, KN>(r));
            acc<KN> d;
            acc_zero<KN>(d);
#pragma unroll
            for (int i = 0; i < KX; i++) d.s[i] = x.l[i];
            acc_add<KN, KR>(d, xr);
            mw<KN> xn = acc_result<KN>(d);
            return NewtonRecip<K, KN>::run(a, xn);
        }
    }
};
template <int K>
MWF mw<K> recip(const mw<K> &a) {
    mw<1> x;
    x.l[0] = 1.0 / a.l[0];
    return NewtonRecip<K, 1>::run(a, x);
}
template <int K>
MWF mw<K> div(const mw<K> &a, const mw<K> &b) {
    mw<K> r = recip<K>(b);
    mw<K> q = mul<K>(a, r);
    // one residual correction: q += r (a - q b)
    mw<K> rem = fnma<K>(a, q, b);
    return fma<K>(q, rem, r);
}

// a / b on a dependent chain: x ~ 1/b to KH = ceil(K/2) limbs (a reciprocal shared by every quotient with this divisor),
// q0 = a x to KH limbs, q = q0 + x (a - b q0): the last Newton step of the reciprocal merged with the multiplication by a,
// so that the full-precision work is one K x KH product for the remainder (its leading KH limbs cancel) and one short
// product for the correction.  Error ~ eps_KH^2 + eps_K, as for recip<K> followed by mul<K> at 0.75 of the latency.
template <int K>
MWF mw<K> div_hr(const mw<K> &a, const mw<K> &b, const mw<(K + 1) / 2> &x) {
    constexpr int KH = (K + 1) / 2;
    constexpr int KR = (K - KH + 1 < K) ? K - KH + 1 : K;
    mw<KH> q0 = mulx<KH, KH, KH>(cvt<KH, K>(a), x);
    acc<K> c;
#pragma unroll
    for (int i = 0; i < K; i++) c.s[i] = a.l[i];
    acc_fma<K, K, KH>(c, b, q0, -1.0);
    mw<K> rem = acc_result<K>(c);
    mw<KR> corr = mulx<KR, KR, KH>(cvt<KR, K>(rem), x);
    acc<K> d;
    acc_zero<K>(d);
#pragma unroll
    for (int i = 0; i < KH; i++) d.s[i] = q0.l[i];
    acc_add<K, KR>(d, corr);
    return acc_result<K>(d);
}
template <int K>
MWF mw<K> div_fast(const mw<K> &a, const mw<K> &b) { return div_hr<K>(a, b, recip<(K + 1) / 2>(cvt<(K + 1) / 2, K>(b))); }

template <int K, int KX>
struct NewtonRsqrt {
    static MWF mw<K> run(const mw<K> &a, const mw<KX> &y) {
        if constexpr (KX >= K) {
            return cvt<K, KX>(y);
        } else {
            constexpr int KN = newton_next(K, KX);
            constexpr int KA = (KN < K ? KN : K);
            // r = 1 - a y^2 to KN limbs
            mw<KN> y2 = mulx<KN, KX, KX>(y, y);
            acc<KN> c;
            acc_zero<KN>(c);
            c.s[0] = 1.0;
            acc_fma<KN, KA, KN>(c, cvt<KA, K>(a), y2, -1.0);
            mw<KN> r = acc_result<KN>(c);
            // y' = y + y r / 2
            constexpr int KR = KN - KX;
            mw<KR> yr = mulx<KR, KX, KR>(y, cvt<KR, KN>(r));
            acc<KN> d;
            acc_zero<KN>(d);
#pragma unroll
            for (int i = 0; i < KX; i++) d.s[i] = y.l[i];
            acc_add<KN, KR>(d, yr, 0.5);
            mw<KN> yn = acc_result<KN>(d);
            return NewtonRsqrt<K, KN>::run(a, yn);
        }
    }
};
// 1 / sqrt(a), a > 0
template <int K>
MWF mw<K> rsqrt(const mw<K> &a) {
    mw<1> y;
    y.l[0] = 1.0 / __builtin_sqrt(a.l[0]);
    return NewtonRsqrt<K, 1>::run(a, y);
}
// sqrt(a) = a * rsqrt(a) with one residual correction: s += (a - s^2) * y / 2
template <int K>
MWF mw<K> sqrt_with_rsqrt(const mw<K> &a, const mw<K> &y) {
    mw<K> s = mul<K>(a, y);
    mw<K> rem = fnma<K>(a, s, s);
    return fma<K>(s, rem, mul_pow2<K>(y, 0.5));
}
template <int K>
MWF mw<K> sqrt(const mw<K> &a) {
    if (!(a.l[0] > 0.0)) return zero<K>();
    return sqrt_with_rsqrt<K>(a, rsqrt<K>(a));
}

// ---------------------------------------------------------------------------------------------------------------------
// planar storage: element i of an array of logical length `plane` has limb l at p[l * plane + i]
// ---------------------------------------------------------------------------------------------------------------------
template <int K>
MWF mw<K> ld(const double *p, long plane, long i) {
    mw<K> r;
#pragma unroll
    for (int l = 0; l < K; l++) r.l[l] = p[(long)l * plane + i];
    return r;
}
template <int K>
MWF mw<K> ld_(const double *p, long plane, long i) { return ld<K>(p, plane, i); }   // for scopes where `ld` is a leading dimension
template <int K>
MWF void st(double *p, long plane, long i, const mw<K> &v) {
#pragma unroll
    for (int l = 0; l < K; l++) p[(long)l * plane + i] = v.l[l];
}

}  // namespace mwa
#endif
