#!/usr/bin/env python3
"""Numerics of the 3-way bf16 split used by the SPLIT3 GEMM path (csrc/gemm_mfma.hip).
x = x0 + x1 + x2 with x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1); a dot product is
formed from the six partial products of weight >= 2^-16.  Prints the truncation error of that
(partial sums in float64, i.e. the split alone) next to the error of a sequential fp32 loop like
the reference's (ViT_seq.c:301-306), both relative to the largest exact result."""
import numpy as np

rng = np.random.default_rng(0)


def bf16_rne(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32) << 16).view(np.float32)


def split3(x):
    a0 = bf16_rne(x)
    r1 = (x - a0).astype(np.float32)
    a1 = bf16_rne(r1)
    r2 = (r1 - a1).astype(np.float32)
    return a0, a1, bf16_rne(r2)


for K in (768, 3072):
    a = rng.uniform(-1, 1, (64, K)).astype(np.float32)
    b = (0.04 * rng.uniform(-1, 1, (64, K))).astype(np.float32)
    exact = a.astype(np.float64) @ b.astype(np.float64).T
    a0, a1, a2 = split3(a)
    b0, b1, b2 = split3(b)
    d = lambda x, y: x.astype(np.float64) @ y.astype(np.float64).T  # noqa: E731
    s6 = d(a0, b0) + d(a0, b1) + d(a1, b0) + d(a0, b2) + d(a2, b0) + d(a1, b1)
    seq = np.zeros((64, 64), np.float32)
    for k in range(K):
        seq += (a[:, k:k + 1] * b[:, k:k + 1].T).astype(np.float32)
    scale = np.abs(exact).max()
    print(f"K={K}: reconstruction max|a-(a0+a1+a2)| = {np.abs(a0.astype(np.float64) + a1 + a2 - a).max():.1e}; "
          f"six-product truncation {np.abs(s6 - exact).max() / scale:.2e}; "
          f"sequential fp32 loop {np.abs(seq - exact).max() / scale:.2e}")


# ---- alternatives measured for DESIGN.md's experiment log (not used by the product unless stated) ----
def split_f16x2(x, scale):
    """x*scale = h0 + h1 + eps with fp16 parts (11-bit significands: 22 bits together)."""
    xs = (x * np.float32(scale)).astype(np.float32)
    h0 = xs.astype(np.float16)
    h1 = (xs - h0.astype(np.float32)).astype(np.float16)
    return h0.astype(np.float64), h1.astype(np.float64)


print("\nalternative decompositions, truncation error relative to the largest exact result:")
for K in (768, 3072):
    a = rng.uniform(-1, 1, (64, K)).astype(np.float32)
    b = (0.04 * rng.uniform(-1, 1, (64, K))).astype(np.float32)
    exact = a.astype(np.float64) @ b.astype(np.float64).T
    scale = np.abs(exact).max()
    a0, a1, a2 = split3(a)
    b0, b1, b2 = split3(b)
    d = lambda x, y: x.astype(np.float64) @ y.astype(np.float64).T  # noqa: E731
    s5 = d(a0, b0) + d(a0, b1) + d(a1, b0) + d(a0, b2) + d(a2, b0)
    ha0, ha1 = split_f16x2(a, 2.0 ** 7)
    hb0, hb1 = split_f16x2(b, 2.0 ** 10)
    s3h = (ha0 @ hb0.T + ha0 @ hb1.T + ha1 @ hb0.T) * 2.0 ** -17
    s4h = s3h + (ha1 @ hb1.T) * 2.0 ** -17
    print(f"K={K}: five bf16 products (a1b1 dropped) {np.abs(s5 - exact).max() / scale:.2e}; "
          f"2 x fp16 parts, three products {np.abs(s3h - exact).max() / scale:.2e}; four products "
          f"{np.abs(s4h - exact).max() / scale:.2e}")
