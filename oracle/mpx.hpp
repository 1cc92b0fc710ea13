/*
 * mpx.hpp -- CPU ORACLE arithmetic (test infrastructure, NOT the product).
 *
 * A small binary multi-precision floating-point type used only by the CPU oracle
 * (oracle/clrs_oracle_mp.cpp): sign, 64-bit exponent and a mantissa of L 64-bit limbs, every
 * operation truncated to `mpx_prec_bits` bits (default 64*L).  It plays the role Arb midpoints
 * play in the reference -- "ordinary rounded multi-precision", src/tools.jl:59-107 uses
 * Arblib.approx_* / get_mid! throughout, i.e. no error balls -- without FLINT, which this image
 * does not have.  Integer limbs on purpose: the HIP product computes in fp64 expansions
 * (csrc/clrs_mw.hip.h); an oracle built from a different number representation cannot share a
 * rounding bug with it.
 *
 * value = s * (m / 2^(64 L)) * 2^e   with the top bit of m[L-1] set (or s == 0: zero).
 * All-zero bytes are the number zero, so calloc'ed arrays are valid.
 */
#ifndef CLRS_ORACLE_MPX_HPP
#define CLRS_ORACLE_MPX_HPP

#include <cmath>
#include <cstdint>
#include <cstring>
#include <type_traits>

static int mpx_prec_bits = 0; /* 0 = full 64*L */

template <int L>
struct mpx {
    typedef unsigned __int128 u128;
    int64_t e;
    int32_t s;
    int32_t pad;
    uint64_t m[L];

    mpx() = default;
    template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type>
    mpx(T v) { from_double((double)v); }

    void from_double(double v) {
        pad = 0;
        for (int i = 0; i < L; i++) m[i] = 0;
        if (v == 0.0 || v != v) { s = 0; e = 0; return; }
        s = v < 0 ? -1 : 1;
        int ex;
        double f = std::frexp(std::fabs(v), &ex); /* f in [0.5, 1) */
        m[L - 1] = (uint64_t)std::ldexp(f, 64);   /* exact: 53 bits */
        e = ex;
    }
    explicit operator double() const {
        if (!s) return 0.0;
        double f = (double)m[L - 1] * 0x1p-64; /* rounds to nearest */
        if (L > 1 && false) f += 0;
        return (double)s * std::ldexp(f, (int)e);
    }

    /* --- helpers -------------------------------------------------------------------------- */
    void trunc_prec() {
        int p = mpx_prec_bits;
        if (p <= 0 || p >= 64 * L) return;
        int drop = 64 * L - p; /* low bits to clear */
        int w = drop / 64, b = drop % 64;
        for (int i = 0; i < w; i++) m[i] = 0;
        if (b) m[w] &= ~((((uint64_t)1) << b) - 1);
    }
    static int cmp_mant(const mpx &a, const mpx &b) {
        for (int i = L - 1; i >= 0; i--) {
            if (a.m[i] != b.m[i]) return a.m[i] < b.m[i] ? -1 : 1;
        }
        return 0;
    }
    static int cmp_abs(const mpx &a, const mpx &b) {
        if (!a.s || !b.s) return (a.s != 0) - (b.s != 0);
        if (a.e != b.e) return a.e < b.e ? -1 : 1;
        return cmp_mant(a, b);
    }
    static int cmp(const mpx &a, const mpx &b) {
        if (a.s != b.s) return a.s < b.s ? -1 : 1;
        if (!a.s) return 0;
        int c = cmp_abs(a, b);
        return a.s > 0 ? c : -c;
    }

    /* --- addition ------------------------------------------------------------------------- */
    static mpx add(const mpx &a, const mpx &b) {
        if (!a.s) return b;
        if (!b.s) return a;
        const mpx *x = &a, *y = &b;
        if (cmp_abs(a, b) < 0) { x = &b; y = &a; }
        int64_t d = x->e - y->e;
        if (d > 64 * L + 2) return *x;
        /* L+1 limb accumulators: one guard limb below */
        uint64_t X[L + 1], Y[L + 1];
        X[0] = 0;
        for (int i = 0; i < L; i++) X[i + 1] = x->m[i];
        {   /* Y = (y.m << 64) >> d */
            uint64_t T[L + 1];
            T[0] = 0;
            for (int i = 0; i < L; i++) T[i + 1] = y->m[i];
            int w = (int)(d / 64), bsh = (int)(d % 64);
            for (int i = 0; i <= L; i++) {
                int src = i + w;
                uint64_t lo = src <= L ? T[src] : 0, hi = src + 1 <= L ? T[src + 1] : 0;
                Y[i] = bsh ? (lo >> bsh) | (hi << (64 - bsh)) : lo;
            }
        }
        mpx r;
        r.pad = 0;
        r.s = x->s;
        r.e = x->e;
        if (x->s == y->s) {
            unsigned carry = 0;
            for (int i = 0; i <= L; i++) {
                u128 t = (u128)X[i] + Y[i] + carry;
                X[i] = (uint64_t)t;
                carry = (unsigned)(t >> 64);
            }
            if (carry) {
                for (int i = 0; i < L; i++) X[i] = (X[i] >> 1) | (X[i + 1] << 63);
                X[L] = (X[L] >> 1) | ((uint64_t)1 << 63);
                r.e += 1;
            }
        } else {
            unsigned borrow = 0;
            for (int i = 0; i <= L; i++) {
                u128 t = (u128)X[i] - Y[i] - borrow;
                X[i] = (uint64_t)t;
                borrow = (unsigned)((t >> 64) & 1);
            }
            /* normalise */
            int top = L;
            while (top >= 0 && X[top] == 0) top--;
            if (top < 0) { r.s = 0; r.e = 0; for (int i = 0; i < L; i++) r.m[i] = 0; return r; }
            int lz = __builtin_clzll(X[top]);
            int wsh = L - top;
            if (wsh) {
                for (int i = L; i >= 0; i--) X[i] = i - wsh >= 0 ? X[i - wsh] : 0;
            }
            if (lz) {
                for (int i = L; i > 0; i--) X[i] = (X[i] << lz) | (X[i - 1] >> (64 - lz));
                X[0] <<= lz;
            }
            r.e -= (int64_t)64 * wsh + lz;
        }
        for (int i = 0; i < L; i++) r.m[i] = X[i + 1];
        r.trunc_prec();
        return r;
    }
    static mpx neg(const mpx &a) { mpx r = a; r.s = -r.s; return r; }

    /* --- multiplication ------------------------------------------------------------------- */
    static mpx mul(const mpx &a, const mpx &b) {
        mpx r;
        r.pad = 0;
        if (!a.s || !b.s) { r.s = 0; r.e = 0; for (int i = 0; i < L; i++) r.m[i] = 0; return r; }
        uint64_t P[2 * L];
        for (int i = 0; i < 2 * L; i++) P[i] = 0;
        for (int i = 0; i < L; i++) {
            uint64_t carry = 0;
            uint64_t ai = a.m[i];
            if (!ai) continue;
            for (int j = 0; j < L; j++) {
                u128 t = (u128)ai * b.m[j] + P[i + j] + carry;
                P[i + j] = (uint64_t)t;
                carry = (uint64_t)(t >> 64);
            }
            P[i + L] = carry;
        }
        r.s = a.s * b.s;
        r.e = a.e + b.e;
        if (!(P[2 * L - 1] >> 63)) {
            for (int i = 2 * L - 1; i > L - 1; i--) P[i] = (P[i] << 1) | (P[i - 1] >> 63);
            r.e -= 1;
        }
        for (int i = 0; i < L; i++) r.m[i] = P[i + L];
        r.trunc_prec();
        return r;
    }

    /* --- division and square root: Newton from a double seed, then one residual correction - */
    static mpx recip(const mpx &b) {
        /* b = s * f * 2^e, f in [0.5,1): 1/b = s * (1/f) * 2^-e */
        double f = (double)b.m[L - 1] * 0x1p-64;
        mpx r(1.0 / f);
        mpx bf = b; bf.e = 0; bf.s = 1;
        mpx one(1.0);
        for (int it = 0; it < 5 && (53 << it) < 64 * L + 64; it++) {
            mpx t = add(one, neg(mul(bf, r)));
            r = add(r, mul(r, t));
        }
        r.e -= b.e;
        r.s = b.s;
        return r;
    }
    static mpx div(const mpx &a, const mpx &b) {
        if (!a.s) return a;
        mpx r = recip(b);
        mpx q = mul(a, r);
        mpx rem = add(a, neg(mul(q, b)));
        return add(q, mul(rem, r));
    }
    static mpx sqrt(const mpx &a) {
        if (a.s <= 0) { mpx z(0.0); return z; }
        /* a = f * 2^e; make e even: f in [0.25, 1) */
        mpx af = a;
        int64_t ex = a.e;
        if (ex & 1) { af.e = -1; ex += 1; } else af.e = 0;
        double f = (double)af;
        mpx y(1.0 / std::sqrt(f));
        mpx one(1.0), half(0.5);
        for (int it = 0; it < 5 && (53 << it) < 64 * L + 64; it++) {
            mpx t = add(one, neg(mul(af, mul(y, y))));
            y = add(y, mul(mul(y, t), half));
        }
        mpx s = mul(af, y);
        mpx rem = add(af, neg(mul(s, s)));
        s = add(s, mul(mul(rem, y), half));
        s.e += ex / 2;
        return s;
    }

    /* --- operators ------------------------------------------------------------------------ */
    friend mpx operator+(const mpx &a, const mpx &b) { return add(a, b); }
    friend mpx operator-(const mpx &a, const mpx &b) { return add(a, neg(b)); }
    friend mpx operator*(const mpx &a, const mpx &b) { return mul(a, b); }
    friend mpx operator/(const mpx &a, const mpx &b) { return div(a, b); }
    mpx operator-() const { return neg(*this); }
    mpx &operator+=(const mpx &b) { *this = add(*this, b); return *this; }
    mpx &operator-=(const mpx &b) { *this = add(*this, neg(b)); return *this; }
    mpx &operator*=(const mpx &b) { *this = mul(*this, b); return *this; }
    mpx &operator/=(const mpx &b) { *this = div(*this, b); return *this; }
    friend bool operator<(const mpx &a, const mpx &b) { return cmp(a, b) < 0; }
    friend bool operator>(const mpx &a, const mpx &b) { return cmp(a, b) > 0; }
    friend bool operator<=(const mpx &a, const mpx &b) { return cmp(a, b) <= 0; }
    friend bool operator>=(const mpx &a, const mpx &b) { return cmp(a, b) >= 0; }
    friend bool operator==(const mpx &a, const mpx &b) { return cmp(a, b) == 0; }
    friend bool operator!=(const mpx &a, const mpx &b) { return cmp(a, b) != 0; }
};

template <int L> static inline mpx<L> mpx_sqrt(const mpx<L> &a) { return mpx<L>::sqrt(a); }
template <int L> static inline mpx<L> mpx_abs(const mpx<L> &a) { mpx<L> r = a; if (r.s < 0) r.s = 1; return r; }

#endif
