#!/usr/bin/env python3
"""Single-branch erf for the GELU epilogue: erf(x) = sign(x) * (1 - 2^(-S(t))), t = min(|x|, c),
S(t) = -log2(erfc(t)) fitted as t*poly(t) on [0, c].  GELU only needs ABSOLUTE erf accuracy
(it computes 1 + erf), so the cancellation of 1 - 2^-S near 0 is harmless.
Prints coefficients and the max |gelu_device - gelu_reference| where the reference is the
scalar formula of ViT_seq.c:285 evaluated with glibc erff in fp32."""
import ctypes
import numpy as np
from scipy import special
from numpy.polynomial import chebyshev as C

libm = ctypes.CDLL("libm.so.6")
libm.erff.restype = ctypes.c_float
libm.erff.argtypes = [ctypes.c_float]
CLAMP = 4.0


def cheb_to_mono(c, lo, hi):
    p = C.cheb2poly(c)
    a, b = 2.0 / (hi - lo), -(hi + lo) / (hi - lo)
    out = np.zeros(len(p))
    base = np.array([1.0])
    for ci in p:
        out[:len(base)] += ci * base
        base = np.convolve(base, [b, a])
    return out


def fit(deg):
    n = 6000
    u = np.cos(np.pi * (np.arange(n) + 0.5) / n)
    t = 0.5 * CLAMP * (u + 1)
    f = -np.log2(special.erfc(t)) / t
    return cheb_to_mono(C.chebfit(u, f, deg), 0.0, CLAMP).astype(np.float32)


def fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def erf_dev(x, q):
    t = np.minimum(np.abs(x), np.float32(CLAMP)).astype(np.float32)
    r = np.full_like(x, q[-1])
    for c in q[-2::-1]:
        r = fma32(r, t, np.full_like(x, c))
    s = (-(t * r)).astype(np.float32)
    e = np.exp2(s.astype(np.float64)).astype(np.float32)          # v_exp_f32, ~1 ulp
    return np.copysign((np.float32(1) - e).astype(np.float32), x)


def gelu_dev(x, q):
    a = (x * np.float32(0.70710678118654752)).astype(np.float32)
    h = (np.float32(0.5) * x).astype(np.float32)
    return (h * (np.float32(1) + erf_dev(a, q)).astype(np.float32)).astype(np.float32)


def gelu_ref(x):
    a = (x / np.sqrt(np.float32(2.0))).astype(np.float32)
    e = np.array([libm.erff(float(v)) for v in a], dtype=np.float32)
    return ((np.float32(0.5) * x).astype(np.float32) * (np.float32(1) + e).astype(np.float32)).astype(np.float32)


xs = np.linspace(-8, 8, 400001).astype(np.float32)
ref = gelu_ref(xs)
exact = 0.5 * xs.astype(np.float64) * (1 + special.erf(xs.astype(np.float64) / np.sqrt(2)))
for deg in (7, 8, 9, 10):
    q = fit(deg)
    g = gelu_dev(xs, q)
    ea = np.abs(erf_dev(xs, q).astype(np.float64) - special.erf(xs.astype(np.float64))).max()
    print(f"deg {deg}: max|erf-erf64|={ea:.3e}  max|gelu-ref(glibc fp32)|={np.abs(g - ref).max():.3e}  "
          f"max|gelu-exact|={np.abs(g - exact).max():.3e}  (ref vs exact {np.abs(ref - exact).max():.3e})")
    if deg in (9, 10):
        print("   coeffs:", ", ".join("%.9ef" % v for v in q))
