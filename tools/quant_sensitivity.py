#!/usr/bin/env python3
"""Where the fp8 mode's logit error comes from: block-scaled e4m3 (MX) applied to ONE operand class at a time.

The library's fp8 mode quantises eight operand classes per layer -- the inputs of the four big projections (LayerNorm-1
output, attention output, LayerNorm-2 output, GELU output) and their weights (in_proj, out_proj, fc1, fc2).  This tool
restates the forward pass in plain PyTorch fp32 (a calculator here, not the product: ViT_seq.c:402-515 in tensor form)
with a fake-quantiser (quantise -> dequantise, the numpy statement tests/mx_ref.py ported to torch) on exactly one class,
in every layer, and reports the relative L2 error of the class logits against the unquantised pass.  Independent errors
add in squares: the root of the sum of squares against `all eight` says how much of the whole the table explains.

With the LayerNorms folded (csrc/norm_fold.h, the library's default since round 4) the activation classes in front of QKV
and fc1 are the UN-normalised residual rows and gamma sits in the weights.  The last block prices the remedy the round-3
review proposed: one projection's two operands on bf16 instead of MX, everything else as the fp8 mode.

Usage: quant_sensitivity.py [preset ...]   (a GPU only makes it quick; it runs on the CPU too)"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as graft  # noqa: E402

DEV = "cuda" if torch.cuda.is_available() else "cpu"
CLASSES = ["ln1_out", "attn_out", "ln2_out", "gelu_out", "in_proj_w", "out_proj_w", "fc1_w", "fc2_w"]
OPERANDS = {"in_proj": ("ln1_out", "in_proj_w"), "out_proj": ("attn_out", "out_proj_w"), "fc1": ("ln2_out", "fc1_w"),
            "fc2": ("gelu_out", "fc2_w")}


def mx(x: torch.Tensor) -> torch.Tensor:
    """quantise -> dequantise along the last axis in blocks of 32: scale 2^E, E = floor(log2 amax) - 8 (+1 where the
    maximum's significand exceeds 1.75: nothing saturates), elements rounded to nearest-even e4m3 (tests/mx_ref.py)"""
    b = x.reshape(-1, 32).float()
    amax = b.abs().amax(dim=1, keepdim=True)
    bits = amax.view(torch.int32)
    expo = (((bits >> 23) & 0xff) - 127 - 8 + ((bits & 0x7fffff) > 0x600000).int()).clamp(min=-126)
    one = torch.ones_like(amax)
    q = (b * torch.ldexp(one, -expo)).to(torch.float8_e4m3fn).float()
    return (q * torch.ldexp(one, expo)).reshape(x.shape)


def bf16(x):
    return x.to(torch.bfloat16).float()


def same(x):
    return x


def forward(cfg, w, imgs, table, fold):
    """The forward pass with operand class c passed through table[c] (missing: untouched) in every layer."""
    t_ = {c: table.get(c, same) for c in CLASSES}
    E, H, L, P = cfg.embed_dim, cfg.num_heads, cfg.depth, cfg.patch_size
    D = E // H
    x = torch.nn.functional.conv2d(imgs, w[1].reshape(E, cfg.in_chans, P, P), w[2], stride=P).flatten(2).transpose(1, 2)
    x = torch.cat([w[0].reshape(1, 1, E).expand(x.shape[0], 1, E), x], dim=1) + w[3].reshape(1, -1, E)

    def normed_linear(x, g, b, W, bias, act_cls, w_cls):
        W = W.reshape(bias.numel(), -1)
        if not fold:
            return t_[act_cls](torch.nn.functional.layer_norm(x, (E,), g, b, 1e-6)) @ t_[w_cls](W).T + bias
        mean = x.mean(-1, keepdim=True)
        rstd = torch.rsqrt((x * x).mean(-1, keepdim=True) - mean * mean + 1e-6)
        Wq = t_[w_cls](W * g[None, :])
        return rstd * (t_[act_cls](x) @ Wq.T - mean * Wq.sum(1)) + (bias + W @ b)
    for l in range(L):
        lw = w[4 + 12 * l: 16 + 12 * l]
        qkv = normed_linear(x, lw[0], lw[1], lw[2], lw[3], "ln1_out", "in_proj_w")
        B, T, _ = qkv.shape
        qh, kh, vh = (t.reshape(B, T, H, D).transpose(1, 2) for t in qkv.split(E, dim=2))
        a = (torch.softmax(qh @ kh.transpose(2, 3) / D ** 0.5, dim=-1) @ vh).transpose(1, 2).reshape(B, T, E)
        x = x + t_["attn_out"](a) @ t_["out_proj_w"](lw[4].reshape(E, E)).T + lw[5]
        h = torch.nn.functional.gelu(normed_linear(x, lw[6], lw[7], lw[8], lw[9], "ln2_out", "fc1_w"))
        x = x + t_["gelu_out"](h) @ t_["fc2_w"](lw[10].reshape(E, -1)).T + lw[11]
    t = w[4 + 12 * L:]
    return torch.nn.functional.layer_norm(x[:, 0], (E,), t[0], t[1], 1e-6) @ t[2].reshape(cfg.num_classes, E).T + t[3]


def main():
    presets = sys.argv[1:] or ["vit_b_16", "vit_l_16", "vit_h_14"]
    pkg = graft.load_package()
    torch.backends.cuda.matmul.allow_tf32 = False
    for preset in presets:
        cfg = pkg.preset(preset)
        n = 32 if preset == "vit_b_16" else 16
        w = [torch.from_numpy(np.ascontiguousarray(a)).to(DEV) for a in pkg.synth_weights(cfg, 0)]
        imgs = torch.from_numpy(pkg.synth_images(cfg, 0, n)).to(DEV)
        with torch.no_grad():
            ref = forward(cfg, w, imgs, {}, False)
            spread = (ref - ref.mean(1, keepdim=True)).norm(dim=1)

            def rel(out):
                return float(((out - ref).norm(dim=1) / spread).mean())
            print(f"# {preset}: {n} synthetic images, synthetic weights; relative L2 of the class logits against the unquantised "
                  f"fp32 pass (mean over images)")
            for fold in (False, True):
                tag = "LayerNorms folded: x quantised, gamma in W" if fold else "separate LayerNorms: LN(x) quantised"
                print(f"## block-scaled e4m3 on ONE operand class, every layer ({tag}; the algebra alone: "
                      f"{rel(forward(cfg, w, imgs, {}, fold)):.1e})")
                errs = {c: rel(forward(cfg, w, imgs, {c: mx}, fold)) for c in CLASSES}
                rss = float(np.sqrt(sum(v * v for v in errs.values())))
                for c in CLASSES:
                    print(f"  {c:11s} {errs[c]:.4f}   {errs[c] ** 2 / rss ** 2:5.1%} of the squared error")
                print(f"  all eight   {rel(forward(cfg, w, imgs, {c: mx for c in CLASSES}, fold)):.4f}   (root of the sum of squares of "
                      f"the rows: {rss:.4f})")
            print("## the fp8 mode (folded) with ONE projection's two operands on bf16 instead")
            for keep, pair in OPERANDS.items():
                table = {c: (bf16 if c in pair else mx) for c in CLASSES}
                print(f"  {keep:8s} on bf16: {rel(forward(cfg, w, imgs, table, True)):.4f}")
            print(f"  bf16 everywhere: {rel(forward(cfg, w, imgs, {c: bf16 for c in CLASSES}, True)):.4f}")


if __name__ == "__main__":
    main()
