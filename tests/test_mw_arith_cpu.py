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
OPS = dict(add=0, sub=1, mul=2, div=3, sqrt=4, recip=5, rsqrt=6, fnma=7, mul_d=8, div_fast=9)


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(_SRC), os.path.getmtime(_HDR)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", _LIB, _SRC], check=True)
    L = C.CDLL(_LIB)
    pd = C.POINTER(C.c_double)
    L.mw_host_op.argtypes = [C.c_int, C.c_int, C.c_long, pd, pd, pd]
    L.mw_host_dot.argtypes = [C.c_int, C.c_long, pd, pd, pd]
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
