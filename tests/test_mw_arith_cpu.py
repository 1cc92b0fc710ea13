"""The product's multi-word fp64 arithmetic (csrc/clrs_mw_arith.h) compiled for the host, against mpmath.

The same header is what the HIP kernels of the extended-precision path compile for the device; here every operation is
checked on the CPU: relative error (with respect to the magnitudes of the operands for sums, Cray-style) within a few units
of 2^-(53K - K - 2) for K = 2..6 limbs, including cancellation, interleaved magnitudes and zero operands."""
import ctypes as C
import os
import subprocess

import mpmath as mp
import numpy as np
import pytest

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "mw_host", "mw_host.cpp")
_LIB = os.path.join(_HERE, "mw_host", "libmw_host.so")
_HDR = os.path.join(_HERE, "..", "clusteredlowranksolver.jl_amd", "csrc", "clrs_mw_arith.h")
_HDR2 = os.path.join(_HERE, "..", "clusteredlowranksolver.jl_amd", "csrc", "clrs_mw_slices.h")
OPS = dict(add=0, sub=1, mul=2, div=3, sqrt=4, recip=5, rsqrt=6, fnma=7, mul_d=8, div_fast=9)


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(_SRC), os.path.getmtime(_HDR), os.path.getmtime(_HDR2)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", _LIB, _SRC], check=True)
    L = C.CDLL(_LIB)
    pd = C.POINTER(C.c_double)
    L.mw_host_op.argtypes = [C.c_int, C.c_int, C.c_long, pd, pd, pd]
    L.mw_host_dot.argtypes = [C.c_int, C.c_long, pd, pd, pd]
    L.mw_host_slice.argtypes = [C.c_int, C.c_long, pd, C.POINTER(C.c_int), C.POINTER(C.c_float)]
    L.mw_host_recombine.argtypes = [C.c_int, C.c_long, pd, C.c_int, pd]
    L.mw_host_exponent.argtypes = [C.c_double]
    return L


def to_limbs(vals, K):
    """mpmath numbers -> (K, n) planar limbs (round to nearest each)."""
    out = np.zeros((K, len(vals)))
    for i, v in enumerate(vals):
        r = mp.mpf(v)
        for l in range(K):
            h = float(r)
            out[l, i] = h
            r -= mp.mpf(h)
    return out


def from_limbs(a):
    return [mp.fsum(mp.mpf(float(a[l, i])) for l in range(a.shape[0])) for i in range(a.shape[1])]


def rand_values(rng, n, K, spread=0):
    vals = []
    for _ in range(n):
        v = mp.mpf(0)
        e0 = int(rng.integers(-40, 40))
        for l in range(K + 1):
            v += mp.mpf(float(rng.standard_normal())) * mp.mpf(2) ** (e0 - 53 * l - int(rng.integers(0, spread + 1)))
        vals.append(v)
    return vals


def call(lib, K, op, a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    c = np.zeros_like(a)
    pd = C.POINTER(C.c_double)
    assert lib.mw_host_op(K, OPS[op], a.shape[1], a.ctypes.data_as(pd), b.ctypes.data_as(pd), c.ctypes.data_as(pd)) == 0
    return c


@pytest.mark.parametrize("K", [2, 3, 4, 5, 6, 8, 10])
def test_mw_operations_against_mpmath(lib, K):
    mp.mp.prec = 53 * K + 200
    rng = np.random.default_rng(K)
    n = 300
    av, bv = rand_values(rng, n, K, spread=20), rand_values(rng, n, K, spread=20)
    # cancellation: b = -a (1 + tiny), and exact opposites; zero operands
    for i in range(0, 40):
        bv[i] = -av[i] * (1 + mp.mpf(2) ** (-int(rng.integers(20, 53 * K - 10))))
    bv[40] = -av[40]
    av[41] = mp.mpf(0)
    a, b = to_limbs(av, K), to_limbs(bv, K)
    av, bv = from_limbs(a), from_limbs(b)        # the values the limbs really hold
    unit = mp.mpf(2) ** (-(53 * K - K - 2))
    worst = {}
    for op in OPS:
        if op in ("sqrt", "rsqrt"):
            aa = to_limbs([abs(v) + mp.mpf(2) ** -60 for v in av], K)
            xs = from_limbs(aa)
            got = from_limbs(call(lib, K, op, aa, b))
            ex = [mp.sqrt(x) if op == "sqrt" else 1 / mp.sqrt(x) for x in xs]
            scale = [abs(e) for e in ex]
        elif op == "recip":
            aa = to_limbs([v if v != 0 else mp.mpf(1) for v in av], K)
            xs = from_limbs(aa)
            got = from_limbs(call(lib, K, op, aa, b))
            ex = [1 / x for x in xs]
            scale = [abs(e) for e in ex]
        elif op in ("div", "div_fast"):
            bb = to_limbs([v if v != 0 else mp.mpf(1) for v in bv], K)
            ys = from_limbs(bb)
            got = from_limbs(call(lib, K, op, a, bb))
            ex = [x / y for x, y in zip(av, ys)]
            scale = [abs(e) for e in ex]
        else:
            got = from_limbs(call(lib, K, op, a, b))
            if op == "add":
                ex = [x + y for x, y in zip(av, bv)]; scale = [abs(x) + abs(y) for x, y in zip(av, bv)]
            elif op == "sub":
                ex = [x - y for x, y in zip(av, bv)]; scale = [abs(x) + abs(y) for x, y in zip(av, bv)]
            elif op == "mul":
                ex = [x * y for x, y in zip(av, bv)]; scale = [abs(e) for e in ex]
            elif op == "fnma":
                ex = [x - y * y for x, y in zip(av, bv)]; scale = [abs(x) + y * y for x, y in zip(av, bv)]
            elif op == "mul_d":
                ex = [x * mp.mpf(float(b[0, i])) for i, x in enumerate(av)]; scale = [abs(e) for e in ex]
        w = mp.mpf(0)
        for g, e, s in zip(got, ex, scale):
            if s == 0:
                assert g == 0
                continue
            w = max(w, abs(g - e) / s)
        worst[op] = float(w / unit)
        assert w <= 4 * unit, (op, K, float(w / unit))
    print(K, {k: round(v, 3) for k, v in worst.items()})


@pytest.mark.parametrize("K", [2, 4, 5, 8])
def test_mw_dot_accumulator(lib, K):
    """Sum a_i b_i through the unnormalised accumulator: error relative to sum |a_i b_i|, including heavy cancellation."""
    mp.mp.prec = 53 * K + 200
    rng = np.random.default_rng(100 + K)
    n = 64
    av, bv = rand_values(rng, n, K), rand_values(rng, n, K)
    # make the sum cancel to ~2^-80 of its terms
    a, b = to_limbs(av, K), to_limbs(bv, K)
    av, bv = from_limbs(a), from_limbs(b)
    partial = mp.fsum(x * y for x, y in zip(av[:-1], bv[:-1]))
    av[-1] = -partial / bv[-1] * (1 + mp.mpf(2) ** -80)
    a = to_limbs(av, K)
    av = from_limbs(a)
    c = np.zeros((K, 1))
    pd = C.POINTER(C.c_double)
    assert lib.mw_host_dot(K, n, a.ctypes.data_as(pd), np.ascontiguousarray(b).ctypes.data_as(pd), c.ctypes.data_as(pd)) == 0
    got = from_limbs(c)[0]
    ex = mp.fsum(x * y for x, y in zip(av, bv))
    scale = mp.fsum(abs(x * y) for x, y in zip(av, bv))
    assert abs(got - ex) <= n * mp.mpf(2) ** (-(53 * K - K - 2)) * scale


def test_limbs_are_nonoverlapping_after_operations(lib):
    """Results are renormalised: |l[i+1]| <= 2 ulp(l[i]) (nearly non-overlapping), zeros only at the tail."""
    K = 4
    mp.mp.prec = 600
    rng = np.random.default_rng(7)
    av, bv = rand_values(rng, 200, K, spread=30), rand_values(rng, 200, K, spread=30)
    a, b = to_limbs(av, K), to_limbs(bv, K)
    for op in ("add", "mul", "div", "fnma"):
        c = call(lib, K, op, a, b)
        for i in range(c.shape[1]):
            for l in range(K - 1):
                hi, lo = c[l, i], c[l + 1, i]
                if hi == 0:
                    assert lo == 0
                else:
                    assert abs(lo) <= 2 * np.spacing(abs(hi)), (op, i, l, hi, lo)


# ---- the conversions of the exact-product scheme (csrc/clrs_mw_slices.h; k_mws_pair and the k_mwx_* kernels run the same code on the device) ----------
BETA = 23


@pytest.mark.parametrize("K", [2, 3, 4, 5, 6, 8, 10])
def test_slices_of_a_number_are_exact_balanced_digits(lib, K):
    """mws_slice: x = 2^e sum_s d_s 2^-(s+1)B up to K 2^(e-SB-1), the digits integers with |d_s| <= 2^(B-1) (+1 in the first) -- the bound the exact
    accumulation of the slice products is stated for -- on renormalised numbers anywhere below their window, numbers with zero limbs in the middle,
    limbs of alternating sign, powers of two and their neighbours, zeros, and expansions that were never renormalised (limbs 2^27 above where a renormalised number has them)."""
    mp.mp.prec = 53 * K + 400
    S = lib.mw_host_slices(K)
    rng = np.random.default_rng(40 + K)
    vals, drops = [], []
    for i in range(400):
        kind = i % 8
        e0 = int(rng.integers(-30, 30))
        if kind == 0:
            v = rand_values(rng, 1, K)[0]
        elif kind == 1:                                   # gaps: a few bits, hundreds of bits apart
            v = sum(mp.mpf(int(rng.integers(-7, 8))) * mp.mpf(2) ** (e0 - int(rng.integers(0, 53 * K))) for _ in range(K))
        elif kind == 2:                                   # limbs of alternating sign, each as large as it can be
            v = sum(mp.mpf((-1) ** l) * (mp.mpf(2) ** 53 - 1) * mp.mpf(2) ** (e0 - 53 * (l + 1)) for l in range(K + 1))
        elif kind == 3:
            v = mp.mpf(2) ** e0 * (1 + (mp.mpf(2) ** -int(rng.integers(1, 53 * K)) if i % 16 == 3 else -mp.mpf(2) ** -int(rng.integers(1, 53 * K))))
        elif kind == 4:
            v = mp.mpf(0)
        elif kind == 5:                                   # digits at their extremes: every slice half a grid step from the next
            v = mp.mpf(2) ** e0 * sum(mp.mpf(2) ** (BETA - 1) * mp.mpf(2) ** (-(s + 1) * BETA) for s in range(S)) / 4
        else:
            v = rand_values(rng, 1, K, spread=30)[0]
        vals.append(v)
        drops.append(int(rng.integers(0, 60)) if kind != 5 else 0)     # the window's exponent lies this far above the number's own
    x = to_limbs(vals, K)
    # never renormalised: every limb 2^27 above the bound of a renormalised number (every fourth number)
    for i in range(3, 400, 4):
        for l in range(1, K):
            x[l, i] = float(np.clip(rng.standard_normal(), -2, 2)) * 2.0 ** (np.floor(np.log2(abs(x[0, i]) + 1e-300)) - 53 * l + 26) if x[0, i] != 0 else 0.0
    xs = from_limbs(x)
    e = np.array([lib.mw_host_exponent(float(sum(x[:, i]))) + 1 + drops[i] if x[0, i] != 0 else 0 for i in range(400)], dtype=np.int32)
    # (the product takes the exponent of the head of a renormalised number: + 1 covers the unnormalised ones, whose head is not their value)
    dg = np.zeros((S, 400), dtype=np.float32)
    assert lib.mw_host_slice(K, 400, np.ascontiguousarray(x).ctypes.data_as(C.POINTER(C.c_double)), e.ctypes.data_as(C.POINTER(C.c_int)),
                             dg.ctypes.data_as(C.POINTER(C.c_float))) == 0
    bad = np.argwhere(~(dg == np.round(dg)))
    assert bad.size == 0, (bad[:5], [(int(i) % 8, x[:, i], int(e[i]), dg[:, i]) for _, i in bad[:2]])
    assert np.abs(dg[1:]).max() <= 2 ** (BETA - 1) and np.abs(dg[0]).max() <= 2 ** (BETA - 1) + 1
    assert np.abs(dg).max() >= 2 ** (BETA - 2)                       # (the cases do reach large digits)
    for i in range(400):
        got = mp.fsum(mp.mpf(float(dg[s, i])) * mp.mpf(2) ** (int(e[i]) - (s + 1) * BETA) for s in range(S))
        assert abs(got - xs[i]) <= K * mp.mpf(2) ** (int(e[i]) - S * BETA - 1), (i, i % 8)


@pytest.mark.parametrize("K", [2, 3, 4, 5, 6, 8, 10])
def test_order_sums_recombine_to_limbs(lib, K):
    """mws_recombine_orders: sum_o a_o 2^-(o+2)B for integer order sums up to the bound of the scheme ((o+1) 32 2^44, any signs), for sums whose leading
    orders cancel, for sparse and for zero input: K limbs, renormalised (each at most 2^-51 of its predecessor), the value good to 2^-(53K-2) of itself
    or to 2^-(SB+5) of the window (the last order is rounded into the last bin)."""
    mp.mp.prec = 53 * K + 400
    S = lib.mw_host_slices(K)
    rng = np.random.default_rng(70 + K)
    n = 300
    a = np.zeros((S, n))
    for i in range(n):
        kind = i % 6
        for o in range(S):
            bound = min((o + 1) * 32 * 2 ** 44, 2 ** 53)
            if kind == 0:
                a[o, i] = float(int(rng.integers(-bound, bound + 1)))
            elif kind == 1:
                a[o, i] = float(bound * (1 if rng.random() < 0.5 else -1))
            elif kind == 2:
                a[o, i] = float(int(rng.integers(-bound, bound + 1))) if rng.random() < 0.3 else 0.0
            elif kind == 3:                                # leading orders cancel: a_0 2^B + a_1 = small
                a[o, i] = float(int(rng.integers(-2 ** 20, 2 ** 20)))
            elif kind == 4:
                a[o, i] = 0.0
            else:
                a[o, i] = float(int(rng.integers(-bound, bound + 1))) if o >= S // 2 else 0.0
        if kind == 3:
            a[0, i] = float(int(rng.integers(1, 2 ** 25)))
            a[1, i] = -a[0, i] * 2 ** BETA + float(int(rng.integers(-5, 6)))
    out = np.zeros((K, n))
    esc = 37
    assert lib.mw_host_recombine(K, n, np.ascontiguousarray(a).ctypes.data_as(C.POINTER(C.c_double)), esc, out.ctypes.data_as(C.POINTER(C.c_double))) == 0
    got = from_limbs(out)
    for i in range(n):
        ex = mp.fsum(mp.mpf(float(a[o, i])) * mp.mpf(2) ** (esc - (o + 2) * BETA) for o in range(S))
        tol = abs(ex) * mp.mpf(2) ** (-(53 * K - 2)) + mp.mpf(2) ** (esc - S * BETA - 5)      # the last odd order is added to the last bin as it is: rounded at 2^-52 of that bin, below 2^-(SB-46)
        assert abs(got[i] - ex) <= tol, (i, i % 6, float(abs(got[i] - ex) / tol))
        for l in range(K - 1):
            assert abs(out[l + 1, i]) <= 2.0 ** -51 * abs(out[l, i])
