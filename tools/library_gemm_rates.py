#!/usr/bin/env python3
"""Yardstick, not product: what the vendor GEMM libraries behind torch.matmul (hipBLASLt / rocBLAS) reach on the four
projection shapes of ViT-B/16 at batch 512, fp32 and bf16 operands, on this GPU -- to set beside
tools/gemm_rates.py (this library's kernels on the same shapes).  torch computes x @ w.T only (no bias, GELU,
residual, no pre-split output); a second fp32 line adds them the way eager torch would (separate kernels).
Usage (GPU box): python tools/library_gemm_rates.py [rounds]"""
import sys

import torch
import torch.nn.functional as F


def timed(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    torch.backends.cuda.matmul.allow_tf32 = False
    dev = torch.device("cuda:0")
    M = 197 * 512
    g = torch.Generator(device=dev).manual_seed(0)
    print(f"torch {torch.__version__}, {torch.cuda.get_device_name(0)}; M = {M}; best of {rounds} rounds x 10 launches")
    for name, K, N, gelu, resid in [("qkv", 768, 2304, False, False), ("out_proj", 768, 768, False, True),
                                    ("fc1", 768, 3072, True, False), ("fc2", 3072, 768, False, True)]:
        x = torch.randn(M, K, device=dev, generator=g)
        w = torch.randn(N, K, device=dev, generator=g) * 0.03
        b = torch.zeros(N, device=dev)
        r = torch.randn(M, N, device=dev, generator=g) if resid else None
        xb, wb, bb = x.bfloat16(), w.bfloat16(), b.bfloat16()
        out32, outb = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev, dtype=torch.bfloat16)

        def full32():
            y = F.linear(x, w, b)
            if gelu:
                y = F.gelu(y)
            if resid:
                y = y + r
            return y

        def fullb():
            y = F.linear(xb, wb, bb)
            return F.gelu(y) if gelu else y

        legs = {"fp32 matmul only          ": lambda: torch.matmul(x, w.t(), out=out32),
                "fp32 linear+bias(+gelu|+r)": full32,
                "bf16 matmul only          ": lambda: torch.matmul(xb, wb.t(), out=outb),
                "bf16 linear+bias(+gelu)   ": fullb}
        best = {k: 1e9 for k in legs}
        for f in legs.values():
            f()
        torch.cuda.synchronize()
        for _ in range(rounds):
            for k, f in legs.items():
                best[k] = min(best[k], timed(f))
        flop = 2.0 * M * N * K
        for k, ms in best.items():
            print(f"{name:9s} M={M} N={N} K={K}  {k}  {ms:7.3f} ms  {flop / ms / 1e9:8.1f} TFLOP/s")
        del x, w, r, xb, wb, out32, outb
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
