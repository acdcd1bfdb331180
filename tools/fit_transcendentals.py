#!/usr/bin/env python3
"""Coefficients of the float atan / log2 / exp2 kernels that csrc/pt_math.h (co_atan2, co_log2, co_exp2) and oracle/hlsl.h state identically
(plain float multiplies and adds in a fixed order: the same bits on the GPU and the CPU), and their accuracy when evaluated in float32.
Least-squares fits at Chebyshev nodes in float64, coefficients rounded to float32.   usage: python tools/fit_transcendentals.py"""
import numpy as np
from numpy.polynomial import polynomial as P
f32 = np.float32


def nodes(lo, hi, n=6000):
    x = np.cos(np.pi * (np.arange(n) + 0.5) / n)
    return (x * (hi - lo) + (hi + lo)) / 2


def horner32(c, x):                      # c[0] + x*(c[1] + ...), every product and sum rounded to float32
    acc = np.full_like(x, f32(c[-1]))
    for k in c[-2::-1]: acc = (acc * x).astype(f32) + f32(k)
    return acc.astype(f32)


def ulps(got, want):
    want32 = want.astype(f32)
    return np.abs(got.astype(np.float64) - want) / np.spacing(np.abs(want32)).astype(np.float64)


def show(name, c):
    print("%s = {%s}" % (name, ", ".join(float(f32(v)).hex() + "f" for v in c)))


# atan(a), a in [0, 1]:  a + a * (s * A(s)), s = a^2
s = nodes(1e-12, 1.0); a = np.sqrt(s)
A = P.polyfit(s, (np.arctan(a) / a - 1) / s, 8).astype(f32)
show("atan  A[9]", A)
a = np.linspace(1e-4, 1, 2_000_001).astype(f32); s = (a * a).astype(f32)
got = (a + (a * (s * horner32(A, s)).astype(f32)).astype(f32)).astype(f32)
print("   max error %.2f ulp" % ulps(got, np.arctan(a.astype(np.float64))).max())

# log2(m), m in (sqrt(1/2), sqrt(2)]:  t * L(t^2), t = (m - 1) / (m + 1)
tm = (np.sqrt(2) - 1) / (np.sqrt(2) + 1)
s = nodes(1e-14, tm * tm); t = np.sqrt(s)
L = P.polyfit(s, np.log2((1 + t) / (1 - t)) / t, 3).astype(f32)
show("log2  L[4]", L)
m = np.linspace(0.7072, 1.4142, 2_000_001).astype(f32)
t = ((m - f32(1)) / (m + f32(1))).astype(f32); s = (t * t).astype(f32)
got = (t * horner32(L, s)).astype(f32)
want = np.log2(m.astype(np.float64)); ok = np.abs(want) > 1e-3
print("   max error %.2f ulp (|log2| > 1e-3), max absolute error %.2e" % (ulps(got[ok], want[ok]).max(), np.abs(got - want).max()))

# exp2(r), r in [-1/2, 1/2]:  1 + r * E(r)
r = nodes(-0.5, 0.5); r = r[np.abs(r) > 1e-9]
E = P.polyfit(r, (2.0 ** r - 1) / r, 5).astype(f32)
show("exp2  E[6]", E)
r = np.linspace(-0.5, 0.5, 2_000_001).astype(f32)
got = (f32(1) + (r * horner32(E, r)).astype(f32)).astype(f32)
print("   max error %.2f ulp" % ulps(got, 2.0 ** r.astype(np.float64)).max())
