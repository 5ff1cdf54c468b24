#!/usr/bin/env python3
"""Derive and verify the branch-free fp32 erf used in the fc1 GELU epilogue.

Two minimax-style (Chebyshev-node least-squares, then fp32-rounded) polynomials:
  |x| <  T : erf(x)  = x + x*P(x^2)
  |x| >= T : erf(x)  = sign(x) * (1 - exp(-|x| * Q(|x|)))      (Q(t) = -log(1-erf(t))/t)
evaluated with fp32 fma chains exactly as the device code does, and compared with
(a) float64 erf and (b) glibc erff (what ViT_seq.c calls) over a dense grid.
Prints the coefficients as C float literals."""
import ctypes
import math
import numpy as np
from scipy import special
from numpy.polynomial import chebyshev as C

T = 0.921875  # crossover (exactly representable)
libm = ctypes.CDLL("libm.so.6")
libm.erff.restype = ctypes.c_float
libm.erff.argtypes = [ctypes.c_float]


def cheb_fit(f, lo, hi, deg, n=4000):
    k = np.arange(n)
    u = np.cos(np.pi * (k + 0.5) / n)
    x = 0.5 * (hi - lo) * u + 0.5 * (hi + lo)
    c = C.chebfit(u, f(x), deg)
    # convert to monomials in x
    p = C.cheb2poly(c)                      # polynomial in u
    # u = (2x - (hi+lo))/(hi-lo)
    a, b = 2.0 / (hi - lo), -(hi + lo) / (hi - lo)
    out = np.zeros(deg + 1)
    base = np.array([1.0])
    for i, ci in enumerate(p):
        out[:len(base)] += ci * base
        base = np.convolve(base, [b, a])
    return out                               # ascending powers of x


def fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def erf_small(x, p):   # x + x*P(s), s = x*x
    s = (x * x).astype(np.float32)
    r = np.full_like(x, np.float32(p[-1]))
    for c in p[-2::-1]:
        r = fma32(r, s, np.full_like(x, np.float32(c)))
    return fma32(r, x, x)


def erf_large(x, q):   # sign(x) * (1 - exp(-t*Q(t)))
    t = np.abs(x)
    r = np.full_like(x, np.float32(q[-1]))
    for c in q[-2::-1]:
        r = fma32(r, t, np.full_like(x, np.float32(c)))
    e = np.exp((-(t.astype(np.float64)) * r.astype(np.float64)).astype(np.float32).astype(np.float64)).astype(np.float32)
    return np.copysign((np.float32(1.0) - e).astype(np.float32), x)


def main():
    # small: (erf(x)/x - 1) as a polynomial in s = x^2 on [0, T^2]
    def fs(s):
        x = np.sqrt(np.maximum(s, 1e-300))
        return np.where(s > 1e-12, special.erf(x) / x - 1.0, 2 / math.sqrt(math.pi) - 1.0)
    p = cheb_fit(fs, 0.0, T * T, 6)
    # large: Q(t) = -log(erfc(t))/t on [T, 4.0]; beyond ~3.92 fp32 erf == 1 (exp underflows to < 2^-25)
    def fq(t):
        return -np.log(special.erfc(t)) / t
    q = cheb_fit(fq, T, 4.0, 7)
    p32, q32 = p.astype(np.float32), q.astype(np.float32)

    xs = np.concatenate([np.linspace(-6, 6, 2_000_001), np.linspace(-T - 0.01, -T + 0.01, 200001),
                         np.linspace(T - 0.01, T + 0.01, 200001)]).astype(np.float32)
    tcl = np.minimum(np.abs(xs), np.float32(4.0)).astype(np.float32)
    approx = np.where(np.abs(xs) < np.float32(T), erf_small(xs, p32), erf_large(np.copysign(tcl, xs), q32))
    exact = special.erf(xs.astype(np.float64))
    err = np.abs(approx.astype(np.float64) - exact)
    glibc = np.array([libm.erff(float(v)) for v in xs[::50]], dtype=np.float32)
    dg = np.abs(approx[::50].astype(np.float64) - glibc.astype(np.float64))
    print("max |approx - erf64| = %.3e at x=%.6f" % (err.max(), xs[err.argmax()]))
    print("max |approx - glibc erff| = %.3e ; glibc vs erf64 max = %.3e" %
          (dg.max(), np.abs(glibc.astype(np.float64) - exact[::50]).max()))
    print("P (small, ascending in s):", ", ".join("%.9ef" % v for v in p32))
    print("Q (large, ascending in t):", ", ".join("%.9ef" % v for v in q32))


if __name__ == "__main__":
    main()
