"""GPU tests of the block-scaled fp8 (MX) path -- csrc/gemm_mx.hip, BASELINE config 5 at the fp8 matrix rate.
The format and the instruction's operand map are pinned by a numpy statement (tests/mx_ref.py): the quantiser byte
for byte, the GEMM against float64 products of the dequantised operands."""
import numpy as np
import pytest

import mx_ref

pytestmark = pytest.mark.gpu


def _dev(pkg, a):
    return pkg.DeviceBuffer.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def _raw(pkg, arr):
    arr = np.ascontiguousarray(arr)
    d = pkg.DeviceBuffer((arr.nbytes + 3) // 4 + 4)
    L = pkg.lib()
    assert L.vh_h2d(d.ptr, arr.ctypes.data, arr.nbytes, None) == 0 and L.vh_device_sync() == 0
    return d


def _bytes(buf, n):
    return buf.to_numpy().view(np.uint8)[:n].copy()


def _launch(pkg, name, *args):
    L = pkg.lib()
    rc = getattr(L, name)(*args)
    assert rc == 0, f"{name}: {L.vh_last_error().decode()}"
    assert L.vh_device_sync() == 0, L.vh_last_error().decode()


def _act_buf(pkg, rows, cols):
    """device buffer for the scale bytes of an ACTIVATION MX tensor (mx_ref.act_scale_bytes)"""
    assert pkg.lib().vh_mx_act_scale_bytes(rows, cols) == mx_ref.act_scale_bytes(rows, cols)
    return pkg.DeviceBuffer(mx_ref.act_scale_bytes(rows, cols) // 4 + 4)


def _act_scales(buf, rows, cols):
    """scale bytes of an activation MX tensor, back in the order of mx_ref.quantize ([K/128][4][rows])"""
    return mx_ref.from_act_layout(buf.to_numpy().view(np.uint8), rows, cols)


def _quantize_gpu(pkg, x, act=False):
    """act: an activation tensor (a GEMM's A operand; scale bytes in the activation order), else a weight matrix"""
    rows, cols = x.shape
    d_x = _dev(pkg, x)
    d_v, d_s = pkg.DeviceBuffer(rows * cols // 4 + 4), _act_buf(pkg, rows, cols) if act else pkg.DeviceBuffer(rows * cols // 128 + 4)
    _launch(pkg, "vh_launch_quantize_mx_act" if act else "vh_launch_quantize_mx_rows", None, d_x.ptr, d_v.ptr, d_s.ptr, rows, cols)
    return d_v, d_s


@pytest.mark.parametrize("rows,cols", [(1, 128), (197, 768), (300, 3072), (33, 256)])
def test_quantize_mx_rows_is_the_numpy_statement_byte_for_byte(pkg, device, oracle, rows, cols):
    x = oracle.synth_fill(rows * cols, 40 + rows, 3.0, 0.2).reshape(rows, cols)
    x[0, :32] = 0.0                                   # an all-zero block
    x[0, 32:40] = [448.0, -448.0, 1.0, -1.0, 2.0 ** -9, 3.0e-30, 500.0, 0.0]
    x[-1, -32:] *= 1.0e-4                             # a block of small values: its scale, not its elements, carries that
    d_v, d_s = _quantize_gpu(pkg, x)
    want_v, want_s = mx_ref.quantize(x)
    got_v = _bytes(d_v, rows * cols).reshape(cols // 128, rows, 128)
    got_s = _bytes(d_s, rows * cols // 32).reshape(cols // 128, 4, rows)
    assert np.array_equal(got_s, want_s)
    zero = (want_v & 0x7f) == 0                       # signed zeros: compare magnitudes
    assert np.array_equal(got_v[~zero], want_v[~zero]) and np.array_equal(got_v[zero] & 0x7f, want_v[zero] & 0x7f)
    d_va, d_sa = _quantize_gpu(pkg, x, act=True)          # the activation form: same values, the scale bytes regrouped
    assert np.array_equal(_bytes(d_va, rows * cols).reshape(cols // 128, rows, 128), got_v)
    assert np.array_equal(_act_scales(d_sa, rows, cols), want_s)
    back = mx_ref.dequantize(got_v, got_s)
    blocks = np.abs(x.reshape(rows, cols // 32, 32))
    assert (np.abs(back - x).reshape(rows, cols // 32, 32) <= 2.0 ** -3 * blocks.max(axis=2, keepdims=True) + 1e-37).all()


@pytest.mark.parametrize("M,K,N,gelu,resid,mx_out", [
    (197, 768, 2304, 0, False, False),      # QKV: one image, 128x128 tiles, ragged last tile
    (197, 768, 768, 0, True, False),        # out-projection + residual (aliasing the output)
    (300, 768, 3072, 1, False, True),       # fc1 + GELU, quantised output
    (300, 3072, 768, 0, True, False),       # fc2 + residual, long K
    (5, 256, 128, 0, False, True),          # smallest legal shape
    (19700, 768, 2304, 0, False, False),    # 100 images: 256x256 tiles + a tail launch of 128x128 tiles
    (56000, 768, 3072, 1, False, True),     # fc1 at scale: big tiles + tail, quantised output
])
def test_linear_mx_vs_float64_products_of_the_dequantised_operands(pkg, device, oracle, M, K, N, gelu, resid, mx_out):
    """Every product of two e4m3 values times two powers of two is exact in fp32, so against float64 sums of the
    SAME dequantised operands only the fp32 accumulation order differs: the fp32 operator tolerance applies
    (relative to the row's magnitude).  A quantised output is compared after dequantisation, within one e4m3
    step of its block maximum."""
    x = oracle.synth_fill(M * K, 800 + M, 1.0, 0.1).reshape(M, K)
    w = oracle.synth_fill(N * K, 801 + N, 0.04, 0.0).reshape(N, K)
    b = oracle.synth_fill(N, 802, 0.1, 0.0)
    r = oracle.synth_fill(M * N, 803, 1.0, 0.0).reshape(M, N)
    d_xv, d_xs = _quantize_gpu(pkg, x, act=True)
    d_wv, d_ws = _quantize_gpu(pkg, w)
    xq = mx_ref.dequantize(*mx_ref.quantize(x))
    wq = mx_ref.dequantize(*mx_ref.quantize(w))
    d_b = _dev(pkg, b)
    rows = np.arange(M) if M <= 300 else np.unique(np.r_[0:24, 244:268, 4084:4108, M // 2:M // 2 + 24, M - 24:M])
    want = xq[rows].astype(np.float64) @ wq.astype(np.float64).T + b
    if gelu:
        from math import erf
        want = 0.5 * want * (1.0 + np.vectorize(erf)(want / np.sqrt(2.0)))
    if resid:
        want = r[rows] + want
    if mx_out:
        d_ov, d_os = pkg.DeviceBuffer(M * N // 4 + 4), _act_buf(pkg, M, N)
        _launch(pkg, "vh_launch_linear_mx", None, d_ov.ptr, d_os.ptr, d_wv.ptr, d_ws.ptr, d_xv.ptr, d_xs.ptr, d_b.ptr,
                M, K, N, gelu, None)
        got = mx_ref.dequantize(_bytes(d_ov, M * N).reshape(N // 128, M, 128), _act_scales(d_os, M, N))[rows]
        bmax = np.abs(want).reshape(len(rows), N // 32, 32).max(axis=2, keepdims=True)
        assert (np.abs(got - want).reshape(len(rows), N // 32, 32) <= 2.0 ** -3 * bmax + 1e-6).all()
    else:
        d_o = _dev(pkg, r) if resid else pkg.DeviceBuffer(M * N)
        _launch(pkg, "vh_launch_linear_mx", None, d_o.ptr, None, d_wv.ptr, d_ws.ptr, d_xv.ptr, d_xs.ptr, d_b.ptr,
                M, K, N, gelu, d_o.ptr if resid else None)
        got = d_o.to_numpy((M, N))[rows]
        assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max())


def test_linear_mx_rejects_bad_arguments(pkg, device):
    L = pkg.lib()
    d = pkg.DeviceBuffer(65536)
    assert L.vh_launch_linear_mx(None, d.ptr, None, d.ptr, d.ptr, d.ptr, d.ptr, d.ptr, 4, 128, 128, 0, None) != 0     # K % 256
    assert L.vh_launch_linear_mx(None, d.ptr, None, d.ptr, d.ptr, d.ptr, d.ptr, d.ptr, 4, 256, 96, 0, None) != 0      # N % 128
    assert L.vh_launch_linear_mx(None, d.ptr, d.ptr, d.ptr, d.ptr, d.ptr, d.ptr, d.ptr, 4, 256, 128, 0, d.ptr) != 0   # MX out + residual
    assert L.vh_launch_quantize_mx_rows(None, d.ptr, d.ptr, d.ptr, 4, 96) != 0


@pytest.mark.parametrize("rows", [1, 5, 197, 1000])
def test_layer_norm_mx_is_layer_norm_then_the_numpy_quantiser(pkg, device, oracle, weights, rows):
    E = 768
    x = oracle.synth_fill(rows * E, 11 + rows, 3.0, 0.5).reshape(rows, E)
    d_x, d_g, d_b = _dev(pkg, x), _dev(pkg, weights[4]), _dev(pkg, weights[5])
    d_y = pkg.DeviceBuffer(rows * E)
    d_v, d_s = pkg.DeviceBuffer(rows * E // 4 + 4), _act_buf(pkg, rows, E)
    _launch(pkg, "vh_launch_layer_norm", None, d_x.ptr, d_g.ptr, d_b.ptr, d_y.ptr, rows, E, E, E, 1e-6)
    _launch(pkg, "vh_launch_layer_norm_mx", None, d_x.ptr, d_g.ptr, d_b.ptr, d_v.ptr, d_s.ptr, rows, E, E, 1e-6)
    want_v, want_s = mx_ref.quantize(d_y.to_numpy((rows, E)))
    got_v = _bytes(d_v, rows * E).reshape(E // 128, rows, 128)
    got_s = _act_scales(d_s, rows, E)
    assert np.array_equal(got_s, want_s)
    zero = (want_v & 0x7f) == 0
    assert np.array_equal(got_v[~zero], want_v[~zero]) and np.array_equal(got_v[zero] & 0x7f, want_v[zero] & 0x7f)


def test_model_fp8_mode_block_scaled(pkg, device, weights, golden_full):
    """BASELINE config 5's precision on ViT-B/16: block-scaled fp8 GEMM operands (no calibration: every 32-element
    block carries its own scale).  Stated tolerance: relative L2 error of the logit vector against the fp32 path
    <= 0.15 (3-bit significands), top-1 equal where the fp32 margin exceeds four times the logit error;
    batch-position independence holds bit for bit."""
    cfg = pkg.preset("vit_b_16")
    imgs = pkg.synth_images(cfg, 0, 8)
    m = pkg.ViTHip(cfg, weights, device=0, max_batch=8, precision="fp8")
    logits, probs = m.forward(imgs)
    again, _ = m.forward(imgs[::-1].copy())
    m.close()
    m32 = pkg.ViTHip(cfg, weights, device=0, max_batch=8)
    ref, _ = m32.forward(imgs)
    m32.close()
    assert np.abs(ref[:4] - golden_full["logits"][:4]).max() <= 1e-4
    assert np.isfinite(logits).all() and np.abs(probs.sum(axis=1) - 1).max() < 1e-5
    assert np.array_equal(again[::-1], logits)
    rel = np.linalg.norm(logits - ref, axis=1) / np.linalg.norm(ref - ref.mean(axis=1, keepdims=True), axis=1)
    print("mxfp8 relative L2 logit error per image:", rel, "max |dlogit|", np.abs(logits - ref).max())
    assert rel.max() <= 0.15
    top2 = np.sort(ref, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 4 * np.abs(logits - ref).max(axis=1)
    assert (logits.argmax(1) == ref.argmax(1))[clear].all()


@pytest.mark.parametrize("M,K,N", [(197, 768, 2304), (19700, 768, 2304)])
def test_linear_mx_fp16_planes_output_is_the_fp32_result_rounded_to_fp16(pkg, device, oracle, M, K, N):
    """vh_launch_linear_mx_planes_f16 (the fp8 mode's Q|K|V): the fp32 output of vh_launch_linear_mx rounded to
    nearest-even fp16, as one-part planes [N/32][M][32]."""
    x = oracle.synth_fill(M * K, 810 + M, 1.0, 0.1).reshape(M, K)
    w = oracle.synth_fill(N * K, 811 + N, 0.04, 0.0).reshape(N, K)
    d_b = _dev(pkg, oracle.synth_fill(N, 812, 0.1, 0.0))
    d_xv, d_xs = _quantize_gpu(pkg, x, act=True)
    d_wv, d_ws = _quantize_gpu(pkg, w)
    d_o, d_h = pkg.DeviceBuffer(M * N), pkg.DeviceBuffer(M * N // 2 + 1)
    _launch(pkg, "vh_launch_linear_mx", None, d_o.ptr, None, d_wv.ptr, d_ws.ptr, d_xv.ptr, d_xs.ptr, d_b.ptr, M, K, N, 0, None)
    _launch(pkg, "vh_launch_linear_mx_planes_f16", None, d_h.ptr, d_wv.ptr, d_ws.ptr, d_xv.ptr, d_xs.ptr, d_b.ptr, M, K, N)
    got = d_h.to_numpy().view(np.float16)[:M * N].reshape(N // 32, M, 32).astype(np.float32).transpose(1, 0, 2).reshape(M, N)
    assert np.array_equal(got, d_o.to_numpy((M, N)).astype(np.float16).astype(np.float32))


@pytest.mark.parametrize("n_images,tokens", [(1, 197), (3, 197), (2, 5), (40, 33), (2, 1)])
def test_attention_writing_mx_equals_attention_then_the_row_quantiser(pkg, device, oracle, n_images, tokens):
    """vh_launch_attention_planes_f16_mx = vh_launch_attention_planes_f16 (fp32 rows) followed by
    vh_launch_quantize_mx_act, byte for byte (values and scales)."""
    E, H = 768, 12
    rows = n_images * tokens
    qkv = oracle.synth_fill(rows * 3 * E, 278 + tokens, 1.0, 0.0).reshape(rows, 3 * E)
    planes = np.ascontiguousarray(qkv.astype(np.float16).reshape(rows, 3 * E // 32, 32).transpose(1, 0, 2))
    d_qh = pkg.DeviceBuffer.from_numpy(planes.ravel().view(np.float32))
    d_o = pkg.DeviceBuffer(rows * E)
    d_v1, d_s1 = pkg.DeviceBuffer(rows * E // 4 + 4), _act_buf(pkg, rows, E)
    d_v2, d_s2 = pkg.DeviceBuffer(rows * E // 4 + 4), _act_buf(pkg, rows, E)
    _launch(pkg, "vh_launch_attention_planes_f16", None, d_qh.ptr, d_o.ptr, 0, n_images, tokens, E, H)
    _launch(pkg, "vh_launch_quantize_mx_act", None, d_o.ptr, d_v1.ptr, d_s1.ptr, rows, E)
    _launch(pkg, "vh_launch_attention_planes_f16_mx", None, d_qh.ptr, d_v2.ptr, d_s2.ptr, n_images, tokens, E, H)
    assert np.array_equal(_act_scales(d_s1, rows, E), _act_scales(d_s2, rows, E))
    assert np.array_equal(_bytes(d_v1, rows * E), _bytes(d_v2, rows * E))
    assert pkg.lib().vh_launch_attention_planes_f16_mx(None, d_qh.ptr, d_v2.ptr, None, n_images, tokens, E, H) != 0


def test_fc1_gelu_of_the_fp8_mode_holds_its_error_bound_for_every_x(pkg, device):
    """The format-matched GELU of the fp8 mode's fc1 epilogue (gelu_lowp2<1>, csrc/gemm_common.h) through the MX GEMM with an
    identity weight: every row holds one value v = +-(1 + m/8) 2^e repeated (exactly representable as block-scaled e4m3, the
    product with 1.0 exact), e from -4 to 6: the dequantised output against erf-GELU (ViT_seq.c:283-287) within the stated
    absolute 1.1e-4 plus one e4m3 rounding of the result, and the negative tail exactly zero -- the clamp of the polynomial's
    argument used to leave -0.5 |x| erfc(3), -5.5e-4 at x = -50, which an all-negative block quantised as signal."""
    from math import erf
    vals = np.array([sgn * (1.0 + m / 8.0) * 2.0 ** e for sgn in (1.0, -1.0) for e in range(-4, 7) for m in range(8)], np.float32)
    M, K = len(vals), 256
    x = np.repeat(vals[:, None], K, axis=1)
    w = np.eye(K, dtype=np.float32)
    d_xv, d_xs = _quantize_gpu(pkg, x, act=True)
    d_wv, d_ws = _quantize_gpu(pkg, w)
    d_b = _dev(pkg, np.zeros(K, np.float32))
    d_ov, d_os = pkg.DeviceBuffer(M * K // 4 + 4), _act_buf(pkg, M, K)
    _launch(pkg, "vh_launch_linear_mx", None, d_ov.ptr, d_os.ptr, d_wv.ptr, d_ws.ptr, d_xv.ptr, d_xs.ptr, d_b.ptr, M, K, K, 1, None)
    got = mx_ref.dequantize(_bytes(d_ov, M * K).reshape(K // 128, M, 128), _act_scales(d_os, M, K))[:, 0].astype(np.float64)
    v64 = vals.astype(np.float64)
    want = 0.5 * v64 * (1.0 + np.vectorize(erf)(v64 / np.sqrt(2.0)))
    err = np.abs(got - want)
    bound = 1.1e-4 + 2.0 ** -4 * np.abs(want)
    assert (err <= bound).all(), f"v = {vals[np.argmax(err - bound)]}"
    assert np.abs(got[vals <= -8.0]).max() == 0.0


@pytest.mark.parametrize("precision", ["bf16", "fp8"])
def test_reduced_modes_against_the_fp32_path_over_256_images(pkg, device, weights, precision):
    """The reduced modes' error against an error MODEL rather than against last round's measurement: tools/quant_sensitivity.py
    restates the forward pass in PyTorch with the modes' operand formats as fake-quantisers and predicts the relative L2
    error of the class logits (profiles/r04_quant_sensitivity.txt: ViT-B/16, LayerNorms folded: block-scaled e4m3 on all
    eight operand classes 0.0745, bf16 on all eight 0.0050; no single class carries more than a quarter of the squared
    error, weights and activations contribute alike).  Over 256 images the library must stay within 1.3 x that
    prediction, and its top-1 / top-5 decisions must agree with the fp32 path's as far as that much logit noise
    allows on random-weight logits (top-2 margins of a fraction of the noise): measured here and printed."""
    cfg = pkg.preset("vit_b_16")
    n = 256
    imgs = pkg.synth_images(cfg, 0, n)
    m32 = pkg.ViTHip(cfg, weights, device=0, max_batch=64)
    ref, _ = m32.forward(imgs)
    m32.close()
    m = pkg.ViTHip(cfg, weights, device=0, max_batch=64, precision=precision)
    got, probs = m.forward(imgs)
    m.close()
    spread = np.linalg.norm(ref - ref.mean(axis=1, keepdims=True), axis=1)
    rel = np.linalg.norm(got - ref, axis=1) / spread
    top1 = float(np.mean(got.argmax(1) == ref.argmax(1)))
    top5 = float(np.mean([len(set(np.argsort(a)[-5:]) & set(np.argsort(b)[-5:])) / 5 for a, b in zip(got, ref)]))
    srt = np.sort(ref, axis=1)
    clear = (srt[:, -1] - srt[:, -2]) > 4 * np.abs(got - ref).max(axis=1)
    print(f"\n{precision} vs fp32 path, {n} images: relative L2 mean {rel.mean():.4f} max {rel.max():.4f}; top-1 agreement {top1:.3f}, "
          f"top-5 overlap {top5:.3f}; images whose fp32 top-2 margin exceeds 4x their logit error: {int(clear.sum())}")
    model = {"bf16": 0.0050, "fp8": 0.0745}[precision]
    assert np.isfinite(got).all() and np.abs(probs.sum(axis=1) - 1.0).max() < 1e-5
    assert rel.mean() <= 1.3 * model and rel.max() <= 1.6 * model
    assert (got.argmax(1) == ref.argmax(1))[clear].all()
    assert top1 >= {"bf16": 0.95, "fp8": 0.70}[precision] and top5 >= {"bf16": 0.97, "fp8": 0.80}[precision]
