#!/usr/bin/env python3
"""Quantisation-error report of the reduced-precision GEMM modes against the fp32 twin
(SURVEY section 8f item 4): per weight family the error of the stored format, and end to end
the class-logit error of the bf16-operand and fp8-operand modes over a batch of images.
Usage: quant_report.py [preset] [n_images]   (needs the GPU; numpy restates the formats)."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import __graft_entry__ as graft  # noqa: E402
import fp8_ref  # noqa: E402
import mx_ref  # noqa: E402


def bf16_round(a):
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32) << 16).view(np.float32)


def fp8_rows(w):
    """e4m3 with one scale per row (round 1's format; kept for comparison)"""
    s = (np.abs(w).max(axis=1) / np.float32(448.0)).astype(np.float32)
    s[s == 0] = 1.0
    return fp8_ref.dequantize(fp8_ref.quantize(w * (np.float32(1.0) / s)[:, None])) * s[:, None]


def fp8_mx(w):
    """block-scaled e4m3 (MX): what the fp8 mode stores"""
    return mx_ref.dequantize(*mx_ref.quantize(w))


def snr_db(ref, q):
    return 10.0 * np.log10(float((ref.astype(np.float64) ** 2).sum()) / max(float(((ref - q).astype(np.float64) ** 2).sum()), 1e-300))


def main():
    preset = sys.argv[1] if len(sys.argv) > 1 else "vit_b_16"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    pkg = graft.load_package()
    cfg = pkg.preset(preset)
    weights = pkg.synth_weights(cfg, 0)
    print(f"# {preset}: {len(weights)} tensors, synthetic weights (seed base 0), {n} synthetic images")
    print("## stored weight formats, layer 0 (signal-to-quantisation-noise, dB)")
    for k, name in ((2, "in_proj"), (4, "out_proj"), (8, "fc1"), (10, "fc2")):
        w = weights[4 + k]
        out_f = weights[4 + k + 1].size
        w2 = w.reshape(out_f, -1)
        print(f"  {name:9s} [{out_f}][{w2.shape[1]}]  bf16 {snr_db(w2, bf16_round(w2)):6.1f}   "
              f"e4m3 + per-row scale {snr_db(w2, fp8_rows(w2)):6.1f}   e4m3 + 32-element block scales (MX) {snr_db(w2, fp8_mx(w2)):6.1f}")
    imgs = pkg.synth_images(cfg, 0, n)
    out = {}
    for prec in ("f32", "bf16", "fp8"):
        m = pkg.ViTHip(cfg, weights, device=0, max_batch=min(n, 64), precision=prec)
        out[prec], _ = m.forward(imgs)
        m.close()
    ref = out["f32"]
    spread = np.linalg.norm(ref - ref.mean(axis=1, keepdims=True), axis=1)
    print("## class logits against the fp32 path (itself within 1e-4 of ViT_seq.c)")
    for prec in ("bf16", "fp8"):
        d = out[prec] - ref
        print(f"  {prec:5s} max |dlogit| {np.abs(d).max():.4f}   rms {np.sqrt((d ** 2).mean()):.5f}   "
              f"relative L2 per image: mean {np.mean(np.linalg.norm(d, axis=1) / spread):.4f} "
              f"max {np.max(np.linalg.norm(d, axis=1) / spread):.4f}   "
              f"top-1 agreement {np.mean(out[prec].argmax(1) == ref.argmax(1)):.3f}   "
              f"top-5 overlap {np.mean([len(set(np.argsort(a)[-5:]) & set(np.argsort(b)[-5:])) / 5 for a, b in zip(out[prec], ref)]):.3f}")


if __name__ == "__main__":
    main()
