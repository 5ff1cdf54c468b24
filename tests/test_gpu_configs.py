"""BASELINE configs 4 and 5 -- ViT-L/16 with bf16 GEMM operands, ViT-H/14 with block-scaled fp8
operands -- checked IN THEIR OWN PRECISION against the CPU port, and at their per-GPU batch sizes
(256 and 512 images) for batch-position independence, bit for bit.

"Parity unpinned" throughout: the reference hard-codes ViT-B/16 (ViT_seq.c:10-21) and has neither
of these shapes nor reduced-precision arithmetic, so the checker is the port with other loop
bounds (bit-identical to the reference's own ViT_seq.c on ViT-B/16, tests/test_oracle.py):
live for the first three layers' residual stream, and through the committed full-depth vectors
tests/golden/{h14,l16}_port_logits.npz (oracle/make_golden_port.py; 1-2 min of CPU per image).
Tolerances are the measured errors plus margin, stated where they are asserted.
"""
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = Path(__file__).resolve().parent / "golden"
H14_SEED, H14_FIRST = 3, 5      # weights seed_base / first synthetic image of the ViT-H/14 vectors
L16_SEED, L16_FIRST = 7, 3


@pytest.fixture(scope="module")
def h14(pkg):
    cfg = pkg.preset("vit_h_14")
    return cfg, pkg.synth_weights(cfg, H14_SEED)


@pytest.fixture(scope="module")
def l16(pkg):
    cfg = pkg.preset("vit_l_16")
    return cfg, pkg.synth_weights(cfg, L16_SEED)


def _rel_l2(got, want):
    return float(np.linalg.norm(got.astype(np.float64) - want) / np.linalg.norm(want.astype(np.float64)))


def _logit_rel_l2(got, want):
    """error of a logit vector relative to the spread of the reference logits (the mean carries no class information)"""
    return float(np.linalg.norm(got - want) / np.linalg.norm(want - want.mean()))


def _clear_top1(got, want, factor=4.0):
    """arg-max must agree where the port's top-2 margin exceeds `factor` x the logit error"""
    top2 = np.sort(want)[-2:]
    return (top2[1] - top2[0]) <= factor * np.abs(got - want).max() or int(got.argmax()) == int(want.argmax())


def test_vit_h14_three_layer_residual_stream_of_every_precision_vs_port(pkg, device, h14):
    """ViT-H/14 (patch 14, T = 257, E = 1280, 16 heads of 80, F = 5120): the residual stream after three
    encoder layers of two images -- gathered-rows patch embedding, streaming / resident-fp16 attention, the
    planes and MX GEMMs -- against the port's (port_forward_image, stop_after_layers = 3; ~10 s of CPU per image),
    for the fp32 path AND, directly against the same port tokens, the bf16-operand and block-scaled-fp8 modes."""
    from oracle.oracle import Oracle
    cfg, weights = h14
    imgs = pkg.synth_images(cfg, H14_FIRST, 2)
    orc = Oracle("vit_h_14")
    with ThreadPoolExecutor(2) as ex:      # the port releases the GIL (ctypes): both images at once
        want = list(ex.map(lambda i: orc.forward(imgs[i], weights, stop_after_layers=3)[2], range(2)))
    short = pkg.preset("vit_h_14")
    short.depth = 3
    w3 = weights[:4 + 12 * 3] + weights[-4:]
    got = {}
    for precision in ("f32", "bf16", "fp8"):
        m = pkg.ViTHip(short, w3, device=0, max_batch=2, precision=precision)
        m.forward(imgs)
        got[precision] = m.read_tokens(2).reshape(2, 257, cfg.embed_dim)
        m.close()
    for i in range(2):
        scale = max(float(np.abs(want[i]).max()), 1.0)
        err = {p: (float(np.abs(got[p][i] - want[i]).max()) / scale, _rel_l2(got[p][i], want[i])) for p in got}
        print(f"ViT-H/14 image {H14_FIRST + i}, 3 layers, (max|d|/max|x|, relative L2) vs port:", err)
        assert err["f32"][0] <= 2e-5, f"fp32 path, image {i}"
        # bf16 operands: 8 significand bits per GEMM operand, fp32 accumulation and residual stream.
        # Measured 3.3e-3 (max) / 3.4e-3 (L2) after three layers; stated bound 6e-3 for both.
        assert err["bf16"][0] <= 6e-3 and err["bf16"][1] <= 6e-3, f"bf16 mode, image {i}"
        # block-scaled e4m3 operands: 4 significand bits, one power-of-two scale per 32 K elements.
        # Measured 5.4e-2 (max) / 5.5e-2 (L2); stated bound 9e-2 for both.
        assert err["fp8"][0] <= 9e-2 and err["fp8"][1] <= 9e-2, f"fp8 mode, image {i}"


def test_vit_h14_full_depth_logits_of_every_precision_vs_port_golden(pkg, device, h14):
    """BASELINE config 5's model at FULL depth (32 layers) against the port's committed logits
    (tests/golden/h14_port_logits.npz, "port, unpinned"): fp32 path within the north star's 1e-4, the
    bf16 mode and config 5's own block-scaled fp8 mode at their stated bounds -- against the port, not
    against this library's fp32 path."""
    cfg, weights = h14
    gold = np.load(GOLDEN / "h14_port_logits.npz")
    assert list(gold["images"]) == [H14_FIRST, H14_FIRST + 1] and int(gold["seed_base"]) == H14_SEED
    imgs = pkg.synth_images(cfg, H14_FIRST, 2)
    out = {}
    for precision in ("f32", "bf16", "fp8"):
        m = pkg.ViTHip(cfg, weights, device=0, max_batch=2, precision=precision)
        out[precision] = m.forward(imgs)
        m.close()
    for i in range(2):
        want_l, want_p = gold["logits"][i], gold["probs"][i]
        l32, p32 = out["f32"][0][i], out["f32"][1][i]
        assert np.abs(l32 - want_l).max() <= 1e-4 and int(l32.argmax()) == int(want_l.argmax())
        assert np.abs(p32 - want_p).max() <= 1e-6
        l16, l8 = out["bf16"][0][i], out["fp8"][0][i]
        e16, e8 = float(np.abs(l16 - want_l).max()), float(np.abs(l8 - want_l).max())
        r16, r8 = _logit_rel_l2(l16, want_l), _logit_rel_l2(l8, want_l)
        print(f"ViT-H/14 image {H14_FIRST + i}, 32 layers vs port: bf16 max|dlogit| {e16:.3e} relL2 {r16:.3e}; "
              f"fp8 max|dlogit| {e8:.3e} relL2 {r8:.3e}")
        assert np.isfinite(l16).all() and np.isfinite(l8).all()
        # bf16 mode, 32 layers: measured max |dlogit| 2.1e-2, relative L2 0.62 %; stated bound 5e-2 / 1.5 %
        assert e16 <= 5e-2 and r16 <= 1.5e-2 and _clear_top1(l16, want_l)
        # fp8 mode (config 5's precision), 32 layers: measured relative L2 0.093-0.097 (max |dlogit| 0.31).  The bound is
        # 1.3 x what the independent error model predicts for this model (tools/quant_sensitivity.py, a PyTorch restatement
        # with fake-quantised operands: 0.0924 with the LayerNorms folded, profiles/r04_quant_sensitivity.txt) = 0.12
        # (round 3 asserted 0.15, set from the measurement itself; round 2 0.25 against this library's own fp32 path)
        assert r8 <= 0.12 and _clear_top1(l8, want_l)
        assert abs(float(out["fp8"][1][i].sum()) - 1.0) < 1e-5


def test_vit_l16_full_depth_logits_of_both_precisions_vs_port_golden(pkg, device, l16):
    """BASELINE config 4's model (ViT-L/16, 24 layers) on two images against the port's committed logits: fp32 path
    within 1e-4; config 4's own precision (bf16 GEMM operands) within 5e-2 (measured 2.0e-2), probabilities within 3e-4."""
    cfg, weights = l16
    gold = np.load(GOLDEN / "l16_port_logits.npz")
    assert list(gold["images"]) == [L16_FIRST, L16_FIRST + 1] and int(gold["seed_base"]) == L16_SEED
    imgs = pkg.synth_images(cfg, L16_FIRST, 2)
    m = pkg.ViTHip(cfg, weights, device=0, max_batch=2)
    l32, p32 = m.forward(imgs)
    m.close()
    m = pkg.ViTHip(cfg, weights, device=0, max_batch=2, precision="bf16")
    l16_, p16 = m.forward(imgs)
    m.close()
    for i in range(2):
        want_l, want_p = gold["logits"][i], gold["probs"][i]
        assert np.abs(l32[i] - want_l).max() <= 1e-4 and int(l32[i].argmax()) == int(want_l.argmax())
        assert np.abs(p32[i] - want_p).max() <= 1e-6
        e16 = float(np.abs(l16_[i] - want_l).max())
        print(f"ViT-L/16 image {L16_FIRST + i}, 24 layers, bf16 mode vs port: max|dlogit| {e16:.3e}")
        assert e16 <= 5e-2 and np.abs(p16[i] - want_p).max() <= 3e-4 and _clear_top1(l16_[i], want_l, 2.0)


def test_vit_h14_fp8_at_512_images_is_batch_position_independent(pkg, device, h14):
    """BASELINE config 5's per-GPU batch in its own precision: 512 x ViT-H/14, block-scaled fp8 (M = 131 584 rows).
    Images at the start, the middle and on the hand-over from the big-tile launches to the tail launches (row
    131 072 = image 510 for every projection's 256-row tiling) get, bit for bit, the logits they get eight at a time."""
    cfg, weights = h14
    imgs = pkg.synth_images(cfg, 0, 512)
    big = pkg.ViTHip(cfg, weights, device=0, max_batch=512, precision="fp8")
    lb, pb = big.forward(imgs)
    big.close()
    pick = [0, 1, 255, 256, 509, 510, 511, 300]
    small = pkg.ViTHip(cfg, weights, device=0, max_batch=8, precision="fp8")
    ls, ps = small.forward(imgs[pick])
    small.close()
    assert np.isfinite(lb).all() and np.abs(pb.sum(axis=1) - 1.0).max() < 1e-5
    assert np.array_equal(lb[pick], ls) and np.array_equal(pb[pick], ps)


def test_vit_l16_bf16_at_256_images_is_batch_position_independent(pkg, device, l16):
    """BASELINE config 4's per-GPU batch in its own precision: 256 x ViT-L/16, bf16 operands (M = 50 432 rows; the
    256x256-tile launches of QKV, fc1 and fc2 hand over to 128x128 tiles at row 49 152 = image 249)."""
    cfg, weights = l16
    imgs = pkg.synth_images(cfg, 0, 256)
    big = pkg.ViTHip(cfg, weights, device=0, max_batch=256, precision="bf16")
    lb, pb = big.forward(imgs)
    big.close()
    pick = [0, 1, 127, 128, 248, 249, 250, 255]
    small = pkg.ViTHip(cfg, weights, device=0, max_batch=8, precision="bf16")
    ls, ps = small.forward(imgs[pick])
    small.close()
    assert np.isfinite(lb).all() and np.array_equal(lb[pick], ls) and np.array_equal(pb[pick], ps)


@pytest.mark.parametrize("precision", ["bf16", "fp8"])
def test_repacked_weights_file_round_trip_for_the_vit_h14_shapes(pkg, device, h14, tmp_path, precision):
    """vit_hip_export_planes / vit_hip_create_from_planes on ViT-H/14's shapes (two layers of it: E = 1280, F = 5120, patch 14 --
    the conv_proj planes are padded from K = 588 to 640 -- and the MX / one-part slabs of config 5's and config 4's
    precisions): the context rebuilt from the one file gives bit-identical logits, and reports the shape it was written with."""
    cfg, weights = h14
    short = pkg.preset("vit_h_14")
    short.depth = 2
    w2 = weights[:4 + 12 * 2] + weights[-4:]
    imgs = pkg.synth_images(cfg, 11, 2)
    a = pkg.ViTHip(short, w2, device=0, max_batch=2, precision=precision)
    la, pa = a.forward(imgs)
    path = tmp_path / f"h14_two_layers_{precision}.planes"
    a.export_planes(path)
    a.close()
    b = pkg.ViTHip.from_planes(path, device=0, max_batch=2)
    assert (b.precision, b.cfg.depth, b.cfg.embed_dim, b.cfg.patch_size, b.cfg.num_heads) == (precision, 2, 1280, 14, 16)
    lb, pb = b.forward(imgs)
    b.close()
    assert np.isfinite(la).all() and np.array_equal(la, lb) and np.array_equal(pa, pb)


TINY = {   # name: (img, patch, classes, embed, depth, heads, mlp, precisions)
    "17 tokens, 4 heads of 64": (64, 16, 10, 256, 2, 4, 512, ("f32", "f32_fp16x2", "bf16", "fp8")),
    # 226 tokens > 208: the streaming attention kernel (Q|K|V as fp32 rows); embed 128 = ONE partial sum per row; fp8 needs embed % 256
    "226 tokens, 2 heads of 64": (240, 16, 3, 128, 1, 2, 256, ("f32", "f32_fp16x2", "bf16")),
    "17 tokens, 2 heads of 128": (64, 16, 1000, 256, 1, 2, 768, ("f32", "bf16", "fp8")),   # head_dim 128: streaming kernel
}


@pytest.mark.parametrize("shape", sorted(TINY))
@pytest.mark.parametrize("fold", ["1", "0"])
def test_a_tiny_custom_config_in_every_precision_vs_port(pkg, device, monkeypatch, tmp_path, fold, shape):
    """BASELINE config 0 speaks of "the repo's own tiny ViT config": the reference has none (its shape is #define'd,
    ViT_seq.c:10-21), so here are three through the extended API (TINY above: few tokens, more tokens than the resident
    attention kernels take, head_dim 128) against the port with the same loop bounds, live ("parity unpinned").  Small enough
    that every launch is a single ragged tile, the class-token rows are a sixth of all rows, the classifier is 10 columns
    wide, and the folded LayerNorm adds two partial sums per row (two of the four lanes of a row contribute nothing).
    fp32 path and its fp16-pair emulation within 1e-4, bf16 within 4e-2, fp8 within 0.15 relative L2; with the LayerNorms
    folded and with separate launches ($VIT_HIP_LN_FOLD)."""
    from oracle.oracle import Oracle
    img, patch, classes, embed, depth, heads, mlp, precisions = TINY[shape]
    orc = Oracle("vit_b_16")
    cfg = pkg.preset("vit_b_16")
    for c in (orc.cfg, cfg):
        c.img_size, c.patch_size, c.in_chans, c.num_classes = img, patch, 3, classes
        c.embed_dim, c.depth, c.num_heads, c.mlp_hidden = embed, depth, heads, mlp
    weights = orc.synth_weights(21)
    assert len(weights) == 4 + 12 * depth + 4 and pkg.binding.tokens(cfg) == (img // patch) ** 2 + 1
    imgs = np.stack([orc.synth_image(i) for i in range(5)])
    want = np.stack([orc.forward(imgs[i], weights)[0] for i in range(5)])
    monkeypatch.setenv("VIT_HIP_LN_FOLD", fold)
    if "fp8" not in precisions:      # a shape the block-scaled mode does not take is refused at creation, loudly -- never a fallback
        with pytest.raises(pkg.VitHipError):
            pkg.ViTHip(cfg, weights, device=0, max_batch=5, precision="fp8")
    for precision in precisions:
        m = pkg.ViTHip(cfg, weights, device=0, max_batch=5, precision=precision)
        got, probs = m.forward(imgs)
        again, _ = m.forward(imgs[[3, 0]])
        m.export_planes(tmp_path / f"tiny_{precision}.planes")
        m.close()
        small = pkg.ViTHip.from_planes(tmp_path / f"tiny_{precision}.planes", device=0, max_batch=2)   # chunks of 2, 2 and 1 images
        chunked, _ = small.forward(imgs)
        folded = bool(pkg.lib().vit_hip_ln_fold(small.ctx))
        small.close()
        # the context rebuilt from the repacked-weights file (fold terms included), run in ragged chunks: the same bits
        assert np.array_equal(chunked, got) and folded == (fold == "1" and precision != "f32_fp16x2")     # the fp16-pair emulation never folds
        err = float(np.abs(got - want).max())
        rel = max(_logit_rel_l2(got[i], want[i]) for i in range(5))
        print(f"tiny config ({shape}), {precision}, fold {fold}: max |dlogit| {err:.3e}, relative L2 {rel:.4f}")
        assert np.isfinite(got).all() and np.abs(probs.sum(axis=1) - 1.0).max() < 1e-5
        assert np.array_equal(again, got[[3, 0]])                      # batch-position independence, bit for bit
        if precision in ("f32", "f32_fp16x2"):
            assert err <= 1e-4 and np.array_equal(got.argmax(1), want.argmax(1))
        elif precision == "bf16":
            assert err <= 4e-2
        else:
            assert rel <= 0.15
