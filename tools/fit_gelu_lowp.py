#!/usr/bin/env python3
"""GELU for epilogues whose output is rounded to bf16 (8 significand bits) or e4m3 (4 bits): the form of
tools/fit_gelu.py -- erf(a) = sign(a) (1 - 2^(-t S(t))), t = min(|a|, c) -- with S of a degree matched to the output
format instead of to fp32, and the final algebra folded:  gelu(x) = max(x, 0) - 0.5 |x| 2^(-t S(t)).
Prints the coefficients (highest power first, as the device code's Horner chain takes them) and, evaluated in fp32
like the device does, the largest absolute error and the largest error relative to the rounding step of the output
format (bf16: 2^-9 of the value, e4m3: 2^-4) against the exact GELU."""
import numpy as np
from numpy.polynomial import chebyshev as C
from scipy import special


def fit(deg, clamp):
    n = 6000
    u = np.cos(np.pi * (np.arange(n) + 0.5) / n)
    t = 0.5 * clamp * (u + 1)
    f = -np.log2(special.erfc(t)) / t
    p = C.cheb2poly(C.chebfit(u, f, deg))
    a, b = 2.0 / clamp, -1.0
    out, base = np.zeros(len(p)), np.array([1.0])
    for ci in p:
        out[:len(base)] += ci * base
        base = np.convolve(base, [b, a])
    return out.astype(np.float32)


def fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def gelu_dev(x, q, clamp):
    a = (x * np.float32(0.70710678118654752)).astype(np.float32)
    t = np.minimum(np.abs(a), np.float32(clamp)).astype(np.float32)
    s = np.full_like(x, q[-1])
    for c in q[-2::-1]:
        s = fma32(s, t, np.full_like(x, c))
    e = np.exp2(-(t * s).astype(np.float32).astype(np.float64)).astype(np.float32)
    h = ((np.float32(0.5) * x).astype(np.float32) * e).astype(np.float32)
    return (np.maximum(x, np.float32(0)) - np.abs(h)).astype(np.float32)


def report(name, deg, clamp, step):
    q = fit(deg, clamp)
    x = np.linspace(-10, 10, 2000001).astype(np.float32)
    want = 0.5 * x.astype(np.float64) * (1 + special.erf(x.astype(np.float64) / np.sqrt(2)))
    got = gelu_dev(x, q, clamp).astype(np.float64)
    err = np.abs(got - want)
    rel = err / np.maximum(np.abs(want), 1e-300)
    body = (x > -3) & (np.abs(x) > 1e-3)
    print(f"{name}: degree {deg}, clamp {clamp}")
    print("  coefficients, highest power first:", ", ".join(f"{c:.9e}f" for c in q[::-1]))
    print(f"  max |gelu - exact| = {err.max():.2e} (x = {x[err.argmax()]:.2f}); for x > -3 the largest relative error is "
          f"{rel[body].max():.2e} = {rel[body].max() / step:.3f} of the output format's rounding step")


if __name__ == "__main__":
    report("bf16 output (one-part planes)", 4, 3.5, 2.0 ** -9)
    report("e4m3 output (MX)", 3, 3.0, 2.0 ** -4)
