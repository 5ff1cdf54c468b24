"""GPU parity tests of the pre-split ("P3") path -- the default fp32 path of the four big
projections (csrc/gemm_p3.hip): producers write the exact three-part bf16 split of every GEMM
input as planes [K/32][3][rows][32], the GEMM's K loop has no split arithmetic left.

Checked: the plane format itself (exact, layout as documented), every P3 launcher bit for bit
against its fp32-activation twin (same six products per block in the same order), the GEMM
against the CPU oracle (ViT_seq.c:295-309) on rows that straddle tile and launch boundaries,
and the whole model bit for bit against the in-loop-split path.
"""
import os
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OP_TOL = 2e-5


def _dev(pkg, a):
    return pkg.DeviceBuffer.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def _launch(pkg, name, *args):
    L = pkg.lib()
    rc = getattr(L, name)(*args)
    assert rc == 0, f"{name}: {L.vh_last_error().decode()}"
    assert L.vh_device_sync() == 0, L.vh_last_error().decode()


def _planes_buf(pkg, rows, cols):
    """Device buffer for planes [cols/32][3][rows][32] bf16 = 6 bytes per value."""
    return pkg.DeviceBuffer((3 * rows * cols + 1) // 2)


def _planes_to_parts(buf, rows, cols):
    """-> float32 [3][rows][cols]: the three parts, in matrix order."""
    raw = buf.to_numpy().view(np.uint16)[:3 * rows * cols].reshape(cols // 32, 3, rows, 32)
    return (raw.astype(np.uint32) << 16).view(np.float32).transpose(1, 2, 0, 3).reshape(3, rows, cols)


def _bf16_rne(x):
    """float32 -> nearest-even bfloat16, as float32 (finite inputs)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


@pytest.mark.parametrize("rows,cols", [(1, 32), (197, 768), (1000, 3072), (33, 64)])
def test_split3_rows_is_the_exact_three_part_split_in_the_documented_layout(pkg, device, oracle, rows, cols):
    x = oracle.synth_fill(rows * cols, 900 + rows, 2.0, 0.3).reshape(rows, cols)
    x[0, :8] = [0.0, -0.0, 1.0, -1.0, 3.0e-30, 65504.0, 1.0e-30, -7.25]     # zeros, tiny and exact values
    d_x, d_p, d_y = _dev(pkg, x), _planes_buf(pkg, rows, cols), pkg.DeviceBuffer(rows * cols)
    _launch(pkg, "vh_launch_split3_rows", None, d_x.ptr, d_p.ptr, rows, cols)
    parts = _planes_to_parts(d_p, rows, cols)
    p0 = _bf16_rne(x)
    p1 = _bf16_rne(x - p0)
    p2 = _bf16_rne(x - p0 - p1)
    assert np.array_equal(parts[0], p0) and np.array_equal(parts[1], p1) and np.array_equal(parts[2], p2)
    assert np.array_equal(parts[0].astype(np.float64) + parts[1] + parts[2], x.astype(np.float64))
    _launch(pkg, "vh_launch_merge3_rows", None, d_p.ptr, d_y.ptr, rows, cols)
    assert np.array_equal(d_y.to_numpy((rows, cols)), x + 0.0)     # -0.0 + 0.0: the merge adds parts


def _sample_rows(M, extra=()):
    """First rows, rows around every 256-row tile edge near `extra`, last rows."""
    idx = set(range(min(M, 24))) | set(range(max(0, M - 24), M))
    for e in extra:
        idx |= set(range(max(0, e - 12), min(M, e + 12)))
    return np.array(sorted(idx))


def _rows_big(M, K, N, resid, cus=256):
    """The row where launch_p3 hands over from 256x256 tiles to 128x128 tiles (0: single launch)."""
    ntiles, mtiles = N // 256, (M + 255) // 256
    tiles = mtiles * ntiles
    if N % 256 or 2 * tiles < 5 * cus or (resid and K < 2048):
        return 0
    full, rem = tiles // cus, tiles % cus
    rb = (full * cus // ntiles) * 256
    if rem == 0 or 4 * rem > 3 * cus or rb <= 0 or rb >= M:
        return 0
    return rb


@pytest.mark.parametrize("M,K,N,gelu,resid,planes_out", [
    (197, 768, 2304, 0, False, False),      # QKV, one image: 128x128 tiles, ragged last tile
    (197, 768, 768, 0, True, False),        # out-projection + residual (aliasing the output)
    (300, 768, 3072, 1, False, True),       # fc1 + GELU writing planes
    (300, 3072, 768, 0, True, False),       # fc2 + residual, long K
    (1, 64, 128, 0, False, True),           # smallest legal shape
    (4300, 768, 3072, 1, False, True),      # 128x128 tiles only (204 big tiles would not fill one round)
    (12608, 3072, 768, 0, True, False),     # fc2 at batch 64: small tiles, long K
    (19700, 768, 2304, 0, False, False),    # QKV at 100 images: 693 big tiles = 2 full rounds + a tail on small tiles
    (60000, 3072, 768, 0, True, False),     # fc2, 705 big tiles: 2 full rounds of big tiles + a tail launch
    (56000, 768, 3072, 1, False, True),     # fc1, 2628 big tiles: 10 rounds + 68 tiles on the small-tile launch
])
def test_linear_p3_matches_in_loop_split_bitwise_and_the_oracle_on_boundary_rows(
        pkg, device, oracle, M, K, N, gelu, resid, planes_out):
    x = oracle.synth_fill(M * K, 500 + M, 1.0, 0.1).reshape(M, K)
    w = oracle.synth_fill(N * K, 501 + N, 0.04, 0.0)
    b = oracle.synth_fill(N, 502, 0.1, 0.0)
    r = oracle.synth_fill(M * N, 503, 1.0, 0.0).reshape(M, N)
    d_x, d_w, d_b = _dev(pkg, x), _dev(pkg, w), _dev(pkg, b)
    d_w3, d_x3 = _planes_buf(pkg, N, K), _planes_buf(pkg, M, K)
    _launch(pkg, "vh_launch_split3_planes", None, d_w.ptr, d_w3.ptr, N, K)
    _launch(pkg, "vh_launch_split3_rows", None, d_x.ptr, d_x3.ptr, M, K)
    # twin: fp32 activations, split inside the K loop
    d_ref = _dev(pkg, r) if resid else pkg.DeviceBuffer(M * N)
    _launch(pkg, "vh_launch_linear_w3", None, d_ref.ptr, d_w3.ptr, d_x.ptr, d_b.ptr, M, K, N, gelu,
            d_ref.ptr if resid else None)
    ref = d_ref.to_numpy((M, N))
    if planes_out:
        d_o3, d_o = _planes_buf(pkg, M, N), pkg.DeviceBuffer(M * N)
        _launch(pkg, "vh_launch_linear_p3", None, d_o3.ptr, 1, d_w3.ptr, d_x3.ptr, d_b.ptr, M, K, N, gelu, None)
        _launch(pkg, "vh_launch_merge3_rows", None, d_o3.ptr, d_o.ptr, M, N)
        got = d_o.to_numpy((M, N))
    else:
        d_o = _dev(pkg, r) if resid else pkg.DeviceBuffer(M * N)
        _launch(pkg, "vh_launch_linear_p3", None, d_o.ptr, 0, d_w3.ptr, d_x3.ptr, d_b.ptr, M, K, N, gelu,
                d_o.ptr if resid else None)
        got = d_o.to_numpy((M, N))
    assert np.array_equal(got, ref + 0.0)
    # the oracle on the rows where tiles and launches meet
    rb = _rows_big(M, K, N, resid)
    rows = np.arange(M) if M <= 300 else _sample_rows(M, extra=(256, 4096, rb, rb + 128) if rb else (256, 4096))
    want = oracle.linear(x[rows], w, b, N)
    if gelu:
        want = oracle.gelu(want.ravel()).reshape(len(rows), N)
    if resid:
        want = r[rows] + want
    assert np.abs(got[rows] - want).max() <= OP_TOL


def test_linear_p3_small_tiles_only_equal_big_tiles(pkg, device, oracle):
    """The 128x128 tile alone (what the tail launch runs) on a shape the 256x256 tile takes: same bits."""
    M, K, N = 55296, 768, 1024            # 864 big tiles: 3 full rounds + 96 tiles on the small-tile launch
    x = oracle.synth_fill(M * K, 520, 1.0, 0.1)
    w = oracle.synth_fill(N * K, 521, 0.04, 0.0)
    b = oracle.synth_fill(N, 522, 0.1, 0.0)
    d_x, d_w, d_b = _dev(pkg, x), _dev(pkg, w), _dev(pkg, b)
    d_w3, d_x3 = _planes_buf(pkg, N, K), _planes_buf(pkg, M, K)
    _launch(pkg, "vh_launch_split3_planes", None, d_w.ptr, d_w3.ptr, N, K)
    _launch(pkg, "vh_launch_split3_rows", None, d_x.ptr, d_x3.ptr, M, K)
    d_big = pkg.DeviceBuffer(M * N)
    _launch(pkg, "vh_launch_linear_p3", None, d_big.ptr, 0, d_w3.ptr, d_x3.ptr, d_b.ptr, M, K, N, 0, None)
    # the first 2304 rows as a problem of their own (9 x 4 big tiles: far below 2.5 rounds, so the small tile):
    M2 = 2304
    d_x3b, d_o2 = _planes_buf(pkg, M2, K), pkg.DeviceBuffer(M2 * N)
    _launch(pkg, "vh_launch_split3_rows", None, d_x.ptr, d_x3b.ptr, M2, K)
    _launch(pkg, "vh_launch_linear_p3", None, d_o2.ptr, 0, d_w3.ptr, d_x3b.ptr, d_b.ptr, M2, K, N, 0, None)
    assert np.array_equal(d_big.to_numpy((M, N))[:M2], d_o2.to_numpy((M2, N)))


@pytest.mark.parametrize("rows", [1, 5, 197, 1000])
def test_layer_norm_p3_equals_layer_norm_split(pkg, device, oracle, weights, rows):
    x = oracle.synth_fill(rows * 768, 11 + rows, 3.0, 0.5).reshape(rows, 768)
    d_x, d_g, d_b = _dev(pkg, x), _dev(pkg, weights[4]), _dev(pkg, weights[5])
    d_y, d_p, d_m = pkg.DeviceBuffer(rows * 768), _planes_buf(pkg, rows, 768), pkg.DeviceBuffer(rows * 768)
    _launch(pkg, "vh_launch_layer_norm", None, d_x.ptr, d_g.ptr, d_b.ptr, d_y.ptr, rows, 768, 768, 768, 1e-6)
    _launch(pkg, "vh_launch_layer_norm_p3", None, d_x.ptr, d_g.ptr, d_b.ptr, d_p.ptr, rows, 768, 768, 1e-6)
    _launch(pkg, "vh_launch_merge3_rows", None, d_p.ptr, d_m.ptr, rows, 768)
    y = d_y.to_numpy((rows, 768))
    assert np.array_equal(d_m.to_numpy((rows, 768)), y + 0.0)
    assert np.abs(y - oracle.layer_norm(x, weights[4], weights[5])).max() <= OP_TOL


@pytest.mark.parametrize("n_images,tokens", [(1, 197), (3, 197), (2, 5), (1, 208), (40, 33)])
def test_attention_p3_equals_attention_split(pkg, device, oracle, n_images, tokens):
    E, H = 768, 12
    qkv = oracle.synth_fill(n_images * tokens * 3 * E, 77 + tokens, 1.0, 0.0)
    rows = n_images * tokens
    d_q, d_o, d_p, d_m = _dev(pkg, qkv), pkg.DeviceBuffer(rows * E), _planes_buf(pkg, rows, E), pkg.DeviceBuffer(rows * E)
    _launch(pkg, "vh_launch_attention", None, d_q.ptr, d_o.ptr, n_images, tokens, E, H)
    _launch(pkg, "vh_launch_attention_p3", None, d_q.ptr, d_p.ptr, n_images, tokens, E, H)
    _launch(pkg, "vh_launch_merge3_rows", None, d_p.ptr, d_m.ptr, rows, E)
    assert np.array_equal(d_m.to_numpy((rows, E)), d_o.to_numpy((rows, E)) + 0.0)


@pytest.mark.parametrize("n_images,tokens", [(1, 197), (3, 197), (2, 5), (1, 208), (40, 33), (300, 64), (2, 1)])
def test_attention_on_planes_equals_attention_on_rows_bitwise(pkg, device, oracle, n_images, tokens):
    """vh_launch_attention_planes (Q, K, V pre-split by the QKV projection's epilogue; csrc/attention_p3.hip)
    against vh_launch_attention on the same fp32 rows: the same six products per block on the same k
    assignment, the same softmax -- identical bits; and against the CPU oracle (ViT_seq.c:192-262)."""
    E, H = 768, 12
    rows = n_images * tokens
    qkv = oracle.synth_fill(rows * 3 * E, 91 + tokens, 1.0, 0.0).reshape(rows, 3 * E)
    d_q, d_q3 = _dev(pkg, qkv), _planes_buf(pkg, rows, 3 * E)
    _launch(pkg, "vh_launch_split3_rows", None, d_q.ptr, d_q3.ptr, rows, 3 * E)
    d_o, d_p, d_m = pkg.DeviceBuffer(rows * E), _planes_buf(pkg, rows, E), pkg.DeviceBuffer(rows * E)
    _launch(pkg, "vh_launch_attention", None, d_q.ptr, d_o.ptr, n_images, tokens, E, H)
    _launch(pkg, "vh_launch_attention_planes", None, d_q3.ptr, d_p.ptr, n_images, tokens, E, H)
    _launch(pkg, "vh_launch_merge3_rows", None, d_p.ptr, d_m.ptr, rows, E)
    got = d_m.to_numpy((rows, E))
    assert np.array_equal(got, d_o.to_numpy((rows, E)) + 0.0)
    if rows <= 600:
        want = np.concatenate([oracle.attention(qkv[i * tokens:(i + 1) * tokens]) for i in range(n_images)])
        assert np.abs(got - want).max() <= OP_TOL


def test_p3_launchers_reject_bad_arguments(pkg, device):
    L = pkg.lib()
    d = pkg.DeviceBuffer(65536)
    assert L.vh_launch_linear_p3(None, d.ptr, 0, d.ptr, d.ptr, d.ptr, 4, 96, 128, 0, None) != 0      # K % 64
    assert L.vh_launch_linear_p3(None, d.ptr, 0, d.ptr, d.ptr, d.ptr, 4, 64, 96, 0, None) != 0       # N % 128
    assert L.vh_launch_linear_p3(None, d.ptr, 1, d.ptr, d.ptr, d.ptr, 4, 64, 128, 0, d.ptr) != 0     # planes + residual
    assert L.vh_launch_linear_p3(None, None, 0, d.ptr, d.ptr, d.ptr, 4, 64, 128, 0, None) != 0
    assert L.vh_launch_split3_rows(None, d.ptr, d.ptr, 4, 48) != 0                                   # cols % 32
    assert L.vh_launch_layer_norm_p3(None, d.ptr, d.ptr, d.ptr, d.ptr, 4, 48, 48, 1e-6) != 0
    assert L.vh_launch_attention_p3(None, d.ptr, d.ptr, 1, 257, 1280, 16) != 0                       # head_dim 80
    assert b"K" in L.vh_last_error() or L.vh_last_error() != b""


def test_model_p3_path_is_bit_identical_to_the_in_loop_split_path(pkg, device, weights, golden_full):
    """Whole model, 5 images (one chunk of 3 + one of 2): default context (pre-split planes) vs
    VIT_HIP_P3=0 (fp32 activations split inside the GEMM and attention loops) -- identical bits;
    and the goldens of the reference's own ViT_seq.c within 1e-4."""
    cfg = pkg.preset("vit_b_16")
    imgs = pkg.synth_images(cfg, 0, 5)
    assert os.environ.get("VIT_HIP_P3") is None
    m = pkg.ViTHip(cfg, weights, device=0, max_batch=3)
    logits, probs = m.forward(imgs)
    m.close()
    os.environ["VIT_HIP_P3"] = "0"
    try:
        m0 = pkg.ViTHip(cfg, weights, device=0, max_batch=3)
        logits0, probs0 = m0.forward(imgs)
        m0.close()
    finally:
        del os.environ["VIT_HIP_P3"]
    assert np.array_equal(logits, logits0) and np.array_equal(probs, probs0)
    assert np.abs(logits[:4] - golden_full["logits"]).max() <= 1e-4
    assert np.array_equal(logits[:4].argmax(1), golden_full["logits"].argmax(1))


def test_model_p3_full_batch_tile_paths(pkg, device, weights):
    """100 images (M = 19 700 rows: QKV and fc1 run 256x256 tiles plus a tail launch of 128x128 tiles,
    the output projection and fc2 run 128x128 tiles): every image's logits equal, bit for bit, those
    of the same image run in a batch of 2 (small tiles only) -- results do not depend on the tile choice."""
    cfg = pkg.preset("vit_b_16")
    imgs = pkg.synth_images(cfg, 0, 100)
    big = pkg.ViTHip(cfg, weights, device=0, max_batch=100)
    lb, _ = big.forward(imgs)
    big.close()
    small = pkg.ViTHip(cfg, weights, device=0, max_batch=2)
    pick = [0, 1, 49, 50, 72, 73, 98, 99]
    ls, _ = small.forward(imgs[pick])
    small.close()
    assert np.array_equal(lb[pick], ls)


def test_benchmark_batch_of_512_gives_every_image_its_small_batch_logits(pkg, device, weights, golden_full):
    """BASELINE's measured configuration itself (ViT-B/16, 512 images, M = 100 864 rows: every projection on its
    big-tile launch plus tail launch, the persistent attention and LayerNorm grids several items deep): the logits of
    images at the tile and launch boundaries (row 98 304, where fc1's and QKV's 256x256-tile launches hand over to
    128x128 tiles, is in image 499; fc2's hand-over row 87 296 in image 443) equal, bit for bit, those of the same
    images run eight at a time, and the first four are within 1e-4 of ViT_seq.c's."""
    cfg = pkg.preset("vit_b_16")
    imgs = pkg.synth_images(cfg, 0, 512)
    big = pkg.ViTHip(cfg, weights, device=0, max_batch=512)
    lb, pb = big.forward(imgs)
    big.close()
    pick = [0, 1, 255, 442, 443, 498, 499, 511]
    small = pkg.ViTHip(cfg, weights, device=0, max_batch=8)
    ls, ps = small.forward(imgs[pick])
    small.close()
    assert np.array_equal(lb[pick], ls) and np.array_equal(pb[pick], ps)
    assert np.abs(lb[:4] - golden_full["logits"][:4]).max() <= 1e-4
    assert np.isfinite(lb).all() and np.abs(pb.sum(axis=1) - 1.0).max() < 1e-5


def test_linear_p3_with_a_permutation_matrix_moves_columns_exactly_at_full_size(pkg, device, oracle):
    """A size-independent property at the benchmark's M = 100 864: with W a (column-selecting) 0/1 matrix and zero bias
    every product is x_part * 1 or zero and the three parts of a value sum back to it without rounding, so
    out[:, j] = x[:, perm[j]] must hold EXACTLY -- checked on every row of the big-tile launch and the tail launch
    (N = 2304) and of the small-tile-only rule (N = 768), for fp32 rows and for planes out."""
    M, K = 100864, 768
    rng = np.random.default_rng(5)
    x = oracle.synth_fill(M * K, 1234, 2.0, 0.3).reshape(M, K)
    d_x, d_x3 = _dev(pkg, x), _planes_buf(pkg, M, K)
    _launch(pkg, "vh_launch_split3_rows", None, d_x.ptr, d_x3.ptr, M, K)
    for N in (768, 2304):
        perm = rng.integers(0, K, N)
        w = np.zeros((N, K), np.float32)
        w[np.arange(N), perm] = 1.0
        b = np.zeros(N, np.float32)
        d_w, d_w3, d_b = _dev(pkg, w), _planes_buf(pkg, N, K), _dev(pkg, b)
        _launch(pkg, "vh_launch_split3_rows", None, d_w.ptr, d_w3.ptr, N, K)
        want = x[:, perm]
        d_o = pkg.DeviceBuffer(M * N)
        _launch(pkg, "vh_launch_linear_p3", None, d_o.ptr, 0, d_w3.ptr, d_x3.ptr, d_b.ptr, M, K, N, 0, None)
        assert np.array_equal(d_o.to_numpy((M, N)), want), f"fp32 rows out, N={N}"
        d_o3 = _planes_buf(pkg, M, N)
        _launch(pkg, "vh_launch_linear_p3", None, d_o3.ptr, 1, d_w3.ptr, d_x3.ptr, d_b.ptr, M, K, N, 0, None)
        parts = _planes_to_parts(d_o3, M, N)
        assert np.array_equal((parts[0] + parts[1]) + parts[2], want), f"planes out, N={N}"


@pytest.mark.parametrize("precision", ["bf16", "fp8"])
def test_reduced_modes_at_batch_512_are_batch_position_independent(pkg, device, weights, precision):
    """BASELINE config 3's size in its own precision (and the fp8 mode): big-tile launches, tail launches and the
    persistent grids give an image the same logits, bit for bit, as a batch of four does."""
    cfg = pkg.preset("vit_b_16")
    imgs = pkg.synth_images(cfg, 0, 512)
    big = pkg.ViTHip(cfg, weights, device=0, max_batch=512, precision=precision)
    lb, _ = big.forward(imgs)
    big.close()
    pick = [0, 255, 499, 511]
    small = pkg.ViTHip(cfg, weights, device=0, max_batch=4, precision=precision)
    ls, _ = small.forward(imgs[pick])
    small.close()
    assert np.isfinite(lb).all() and np.array_equal(lb[pick], ls)


# ---- the same kernel with ONE part per value: the bf16-operand mode (BASELINE config 3) -----------


def _planes1_to_f32(buf, rows, cols):
    raw = buf.to_numpy().view(np.uint16)[:rows * cols].reshape(cols // 32, 1, rows, 32)
    return (raw.astype(np.uint32) << 16).view(np.float32).transpose(1, 2, 0, 3).reshape(rows, cols)


@pytest.mark.parametrize("M,K,N,gelu,resid,planes_out", [
    (197, 768, 2304, 0, False, False),     # QKV: planes in, fp32 out
    (197, 768, 768, 0, True, False),       # out-projection + residual
    (300, 768, 3072, 1, False, True),      # fc1 + GELU, one-part planes out
    (300, 3072, 768, 0, True, False),      # fc2 + residual
    (5, 128, 128, 0, False, True),         # smallest legal shape
    (19700, 768, 2304, 0, False, False),   # 100 images: 256x256 tiles + a tail launch of 128x128 tiles
    (60000, 3072, 768, 0, True, False),    # long K, big tiles + tail
])
def test_linear_planes_one_part_vs_oracle_on_bf16_rounded_operands(pkg, device, oracle, M, K, N, gelu, resid, planes_out):
    """vh_launch_linear_planes(parts = 1): operands rounded to bf16 by vh_launch_split_rows(parts = 1),
    fp32 accumulation.  The oracle's fp32 loop (ViT_seq.c:295-309) on the SAME rounded operands differs
    only by summation order, so the fp32 operator tolerance applies (rows sampled around tile and launch
    boundaries for the large shapes)."""
    x = oracle.synth_fill(M * K, 700 + M, 1.0, 0.1).reshape(M, K)
    w = oracle.synth_fill(N * K, 701 + N, 0.04, 0.0)
    b = oracle.synth_fill(N, 702, 0.1, 0.0)
    r = oracle.synth_fill(M * N, 703, 1.0, 0.0).reshape(M, N)
    d_x, d_w, d_b = _dev(pkg, x), _dev(pkg, w), _dev(pkg, b)
    d_w1, d_x1 = pkg.DeviceBuffer((N * K + 1) // 2), pkg.DeviceBuffer((M * K + 1) // 2)
    _launch(pkg, "vh_launch_split_rows", None, d_w.ptr, d_w1.ptr, N, K, 1)
    _launch(pkg, "vh_launch_split_rows", None, d_x.ptr, d_x1.ptr, M, K, 1)
    xr, wr = _planes1_to_f32(d_x1, M, K), _planes1_to_f32(d_w1, N, K)
    assert np.array_equal(xr, _bf16_rne(x)) and np.array_equal(wr.ravel(), _bf16_rne(w))     # format + rounding
    if planes_out:
        d_o1 = pkg.DeviceBuffer((M * N + 1) // 2)
        _launch(pkg, "vh_launch_linear_planes", None, d_o1.ptr, 1, d_w1.ptr, d_x1.ptr, 1, d_b.ptr, M, K, N, gelu, None)
        got = _planes1_to_f32(d_o1, M, N)
    else:
        d_o = _dev(pkg, r) if resid else pkg.DeviceBuffer(M * N)
        _launch(pkg, "vh_launch_linear_planes", None, d_o.ptr, 0, d_w1.ptr, d_x1.ptr, 1, d_b.ptr, M, K, N, gelu,
                d_o.ptr if resid else None)
        got = d_o.to_numpy((M, N))
    rb = _rows_big(M, K, N, resid)
    rows = np.arange(M) if M <= 300 else _sample_rows(M, extra=(256, 4096, rb, rb + 128) if rb else (256, 4096))
    want = oracle.linear(xr[rows], wr.ravel(), b, N)
    if gelu:
        want = oracle.gelu(want.ravel()).reshape(len(rows), N)
    if resid:
        want = r[rows] + want
    tol = OP_TOL + (2.0 ** -8 * np.abs(want).max() if planes_out else 0.0)          # one bf16 rounding of the output
    assert np.abs(got[rows] - want).max() <= tol


def test_layer_norm_and_attention_one_part_planes_are_the_rounded_fp32_results(pkg, device, oracle, weights):
    """vh_launch_layer_norm_planes(parts = 1) writes vh_launch_layer_norm's values rounded to bf16 (nearest even),
    vh_launch_attention_planes_bf16 those of vh_launch_attention_f16 (the reduced modes' attention), in plane order."""
    rows, E, H = 1000, 768, 12
    x = oracle.synth_fill(rows * E, 31, 3.0, 0.5).reshape(rows, E)
    d_x, d_g, d_b = _dev(pkg, x), _dev(pkg, weights[4]), _dev(pkg, weights[5])
    d_y, d_p1 = pkg.DeviceBuffer(rows * E), pkg.DeviceBuffer(rows * E // 2)
    _launch(pkg, "vh_launch_layer_norm", None, d_x.ptr, d_g.ptr, d_b.ptr, d_y.ptr, rows, E, E, E, 1e-6)
    _launch(pkg, "vh_launch_layer_norm_planes", None, d_x.ptr, d_g.ptr, d_b.ptr, d_p1.ptr, 1, rows, E, E, 1e-6)
    assert np.array_equal(_planes1_to_f32(d_p1, rows, E), _bf16_rne(d_y.to_numpy((rows, E))))
    n_images, tokens = 3, 197
    rows = n_images * tokens
    qkv = oracle.synth_fill(rows * 3 * E, 78, 1.0, 0.0)
    d_q, d_o, d_o1 = _dev(pkg, qkv), pkg.DeviceBuffer(rows * E), pkg.DeviceBuffer(rows * E // 2 + 1)
    _launch(pkg, "vh_launch_attention_f16", None, d_q.ptr, d_o.ptr, n_images, tokens, E, H)
    _launch(pkg, "vh_launch_attention_planes_bf16", None, d_q.ptr, d_o1.ptr, n_images, tokens, E, H)
    assert np.array_equal(_planes1_to_f32(d_o1, rows, E), _bf16_rne(d_o.to_numpy((rows, E))))


def _planes_f16_to_f32(buf, rows, cols):
    """one-part fp16 planes [cols/32][rows][32] -> float32 [rows][cols]"""
    raw = buf.to_numpy().view(np.float16)[:rows * cols].reshape(cols // 32, rows, 32)
    return raw.astype(np.float32).transpose(1, 0, 2).reshape(rows, cols)


def _f16_planes_dev(pkg, x):
    """float32 [rows][cols] -> device buffer of one-part fp16 planes [cols/32][rows][32] (numpy rounds to nearest even)"""
    rows, cols = x.shape
    planes = np.ascontiguousarray(x.astype(np.float16).reshape(rows, cols // 32, 32).transpose(1, 0, 2))
    pad = (-planes.size) % 2
    raw = np.concatenate([planes.ravel().view(np.uint16), np.zeros(pad, np.uint16)]).view(np.float32)
    return pkg.DeviceBuffer.from_numpy(raw)


@pytest.mark.parametrize("M,K,N", [(197, 768, 2304), (5, 128, 128), (19700, 768, 2304)])
def test_linear_planes_fp16_output_is_the_fp32_result_rounded_to_fp16(pkg, device, oracle, M, K, N):
    """output_planes = 2 (the reduced modes' Q|K|V): the values of output_planes = 0 rounded to nearest-even fp16,
    in one-part plane order -- on small tiles, big tiles and the tail launch."""
    x = oracle.synth_fill(M * K, 900 + M, 1.0, 0.1).reshape(M, K)
    w = oracle.synth_fill(N * K, 901 + N, 0.04, 0.0)
    b = oracle.synth_fill(N, 902, 0.1, 0.0)
    d_x, d_w, d_b = _dev(pkg, x), _dev(pkg, w), _dev(pkg, b)
    d_w1, d_x1 = pkg.DeviceBuffer((N * K + 1) // 2), pkg.DeviceBuffer((M * K + 1) // 2)
    _launch(pkg, "vh_launch_split_rows", None, d_w.ptr, d_w1.ptr, N, K, 1)
    _launch(pkg, "vh_launch_split_rows", None, d_x.ptr, d_x1.ptr, M, K, 1)
    d_o, d_h = pkg.DeviceBuffer(M * N), pkg.DeviceBuffer((M * N + 1) // 2)
    _launch(pkg, "vh_launch_linear_planes", None, d_o.ptr, 0, d_w1.ptr, d_x1.ptr, 1, d_b.ptr, M, K, N, 0, None)
    _launch(pkg, "vh_launch_linear_planes", None, d_h.ptr, 2, d_w1.ptr, d_x1.ptr, 1, d_b.ptr, M, K, N, 0, None)
    assert np.array_equal(_planes_f16_to_f32(d_h, M, N), d_o.to_numpy((M, N)).astype(np.float16).astype(np.float32))
    L = pkg.lib()
    assert L.vh_launch_linear_planes(None, d_h.ptr, 2, d_w1.ptr, d_x1.ptr, 1, d_b.ptr, M, K, N, 1, None) != 0      # no GELU
    assert L.vh_launch_linear_planes(None, d_h.ptr, 2, d_w1.ptr, d_x1.ptr, 3, d_b.ptr, M, K, N, 0, None) != 0      # one part only
    assert L.vh_launch_linear_planes(None, d_h.ptr, 3, d_w1.ptr, d_x1.ptr, 1, d_b.ptr, M, K, N, 0, None) != 0


@pytest.mark.parametrize("n_images,tokens", [(1, 197), (3, 197), (2, 5), (1, 208), (40, 33), (300, 64), (2, 1)])
def test_attention_on_fp16_planes_equals_the_fp16_operand_attention_on_rows_bitwise(pkg, device, oracle, n_images, tokens):
    """vh_launch_attention_planes_f16 (Q|K|V already rounded to fp16 by the projection, as planes) against
    vh_launch_attention_f16 / vh_launch_attention_planes_bf16 (fp32 rows in, rounded inside the kernel): the same
    products in the same order -- fp32 rows out and one-part bf16 planes out, bit for bit."""
    E, H = 768, 12
    rows = n_images * tokens
    qkv = oracle.synth_fill(rows * 3 * E, 178 + tokens, 1.0, 0.0).reshape(rows, 3 * E)
    d_q, d_qh = _dev(pkg, qkv), _f16_planes_dev(pkg, qkv)
    d_a, d_b = pkg.DeviceBuffer(rows * E), pkg.DeviceBuffer(rows * E)
    _launch(pkg, "vh_launch_attention_f16", None, d_q.ptr, d_a.ptr, n_images, tokens, E, H)
    _launch(pkg, "vh_launch_attention_planes_f16", None, d_qh.ptr, d_b.ptr, 0, n_images, tokens, E, H)
    a, b = d_a.to_numpy((rows, E)), d_b.to_numpy((rows, E))
    assert np.isfinite(b).all() and np.array_equal(a, b)
    d_a1, d_b1 = pkg.DeviceBuffer(rows * E // 2 + 1), pkg.DeviceBuffer(rows * E // 2 + 1)
    _launch(pkg, "vh_launch_attention_planes_bf16", None, d_q.ptr, d_a1.ptr, n_images, tokens, E, H)
    _launch(pkg, "vh_launch_attention_planes_f16", None, d_qh.ptr, d_b1.ptr, 1, n_images, tokens, E, H)
    assert np.array_equal(_planes1_to_f32(d_a1, rows, E), _planes1_to_f32(d_b1, rows, E))
    L = pkg.lib()
    assert L.vh_launch_attention_planes_f16(None, d_qh.ptr, d_b.ptr, 0, n_images, 209, E, H) != 0
    assert L.vh_launch_attention_planes_f16(None, None, d_b.ptr, 0, n_images, tokens, E, H) != 0



def test_gather_rows_picks_every_nth_row_of_matrices_and_planes(pkg, device, oracle):
    rows, cols, stride = 985, 768, 197
    x = oracle.synth_fill(rows * cols, 41, 1.0, 0.0).reshape(rows, cols)
    d_x, d_o = _dev(pkg, x), pkg.DeviceBuffer(5 * cols)
    _launch(pkg, "vh_launch_gather_rows", None, d_x.ptr, d_o.ptr, 1, rows, 5, 4 * cols, stride)
    assert np.array_equal(d_o.to_numpy((5, cols)), x[::stride])
    d_p, d_g = _planes_buf(pkg, rows, cols), _planes_buf(pkg, 5, cols)
    _launch(pkg, "vh_launch_split3_rows", None, d_x.ptr, d_p.ptr, rows, cols)
    _launch(pkg, "vh_launch_gather_rows", None, d_p.ptr, d_g.ptr, 3 * (cols // 32), rows, 5, 64, stride)
    assert np.array_equal(_planes_to_parts(d_g, 5, cols), _planes_to_parts(d_p, rows, cols)[:, ::stride])
    L = pkg.lib()
    assert L.vh_launch_gather_rows(None, d_x.ptr, d_o.ptr, 1, rows, 6, 4 * cols, stride) != 0       # row 985 does not exist
    assert L.vh_launch_gather_rows(None, d_x.ptr, d_o.ptr, 1, rows, 5, 4 * cols + 4, stride) != 0


def test_last_layer_on_class_token_rows_only_gives_identical_logits(pkg, device, weights, golden_full):
    """vit_hip_set_last_layer_cls_only: the last layer's output projection, LayerNorm and MLP evaluated for the rows
    the classifier reads.  Same kernels, same k order: logits and probabilities bit for bit those of the full
    evaluation (and within 1e-4 of ViT_seq.c's), for one image and for a batch that spans several tiles."""
    cfg = pkg.preset("vit_b_16")
    imgs = pkg.synth_images(cfg, 0, 40)
    m = pkg.ViTHip(cfg, weights, device=0, max_batch=40)
    full_l, full_p = m.forward(imgs)
    assert m.set_last_layer_cls_only(True) is False
    cls_l, cls_p = m.forward(imgs)
    one_l, _ = m.forward(imgs[3:4])
    assert m.set_last_layer_cls_only(False) is True
    again_l, _ = m.forward(imgs)
    m.close()
    assert np.array_equal(cls_l, full_l) and np.array_equal(cls_p, full_p) and np.array_equal(again_l, full_l)
    assert np.array_equal(one_l[0], full_l[3])
    assert np.abs(full_l[:4] - golden_full["logits"][:4]).max() <= 1e-4


@pytest.mark.parametrize("preset,n", [("vit_b_16", 3), ("vit_h_14", 2), ("vit_b_16", 30)])
def test_patch_embed_on_one_part_planes_vs_oracle_on_bf16_rounded_operands(pkg, device, preset, n):
    """vh_launch_patch_embed_planes -- the reduced modes' conv_proj (conv2d.cl:1-80): im2row producer writing one-part
    bf16 planes, planes GEMM with the token-row epilogue.  The port's conv loop (ViT_seq.c:25-57 restated) on the SAME
    bf16-rounded pixels and weights differs by summation order only: the fp32 operator tolerance applies.  ViT-H/14's
    patch 14 (K = 588, padded to 640 with zeros) takes the element-wise gather; 30 images of ViT-B/16 (M = 5880) run
    several 128x128 tiles and a ragged last one."""
    from oracle.oracle import Oracle
    orc = Oracle(preset)
    cfg = pkg.preset(preset)
    E, T, P = cfg.embed_dim, pkg.binding.tokens(cfg), cfg.patch_size
    W = [orc.synth_fill(orc.tensor_size(i), 40 + i, 0.05, 0.0) for i in range(4)]
    imgs = pkg.synth_images(cfg, 20, n)
    L = pkg.lib()
    Kp = L.vh_patch_planes_k(3, P)
    assert Kp == (768 if P == 16 else 640)
    d = [_dev(pkg, a) for a in (imgs, W[1], W[2], W[0], W[3])]
    d_wp = pkg.DeviceBuffer(E * Kp // 2)
    _launch(pkg, "vh_launch_conv_weight_planes", None, d[1].ptr, d_wp.ptr, E, 3, P)
    wr = _planes1_to_f32(d_wp, E, Kp)
    K = 3 * P * P
    assert np.array_equal(wr[:, :K], _bf16_rne(W[1]).reshape(E, K)) and not wr[:, K:].any()
    need = n * (T - 1) * Kp * 2
    d_ws, d_tok = pkg.DeviceBuffer(need // 4), pkg.DeviceBuffer(n * T * E)
    assert L.vh_launch_patch_embed_planes(None, d[0].ptr, d_wp.ptr, d[2].ptr, d[3].ptr, d[4].ptr, d_tok.ptr, n, 3, 224, P, E,
                                          d_ws.ptr, need - 16) != 0          # workspace too small
    _launch(pkg, "vh_launch_patch_embed_planes", None, d[0].ptr, d_wp.ptr, d[2].ptr, d[3].ptr, d[4].ptr, d_tok.ptr, n, 3, 224, P, E,
            d_ws.ptr, need)
    got = d_tok.to_numpy((n, T, E))
    for i in sorted({0, n // 2, n - 1}):
        want = orc.tokens_from_conv(orc.conv2d(_bf16_rne(imgs[i]), _bf16_rne(W[1]), W[2]), W[0], W[3])
        assert np.abs(got[i] - want).max() <= OP_TOL, f"image {i}"


@pytest.mark.parametrize("preset,n", [("vit_b_16", 3), ("vit_h_14", 2), ("vit_b_16", 30)])
def test_patch_embed_on_three_part_planes_vs_oracle_and_the_in_loop_split(pkg, device, preset, n):
    """vh_launch_patch_embed_planes3 -- the fp32 path's conv_proj on the planes kernel (conv2d.cl:1-80): the im2row producer
    writes the exact three-part split of the pixels, the GEMM forms the six products per block like every other fp32
    projection.  Against the port's conv loop (ViT_seq.c:25-57) at the fp32 operator tolerance, and against round 1's
    kernel that splits both operands inside its K loop (vh_launch_patch_embed_ws): the same products on the same k
    assignment in the same order -- equal bit for bit."""
    from oracle.oracle import Oracle
    orc = Oracle(preset)
    cfg = pkg.preset(preset)
    E, T, P = cfg.embed_dim, pkg.binding.tokens(cfg), cfg.patch_size
    W = [orc.synth_fill(orc.tensor_size(i), 40 + i, 0.05, 0.0) for i in range(4)]
    imgs = pkg.synth_images(cfg, 20, n)
    L = pkg.lib()
    Kp = L.vh_patch_planes_k(3, P)
    d = [_dev(pkg, a) for a in (imgs, W[1], W[2], W[0], W[3])]
    d_wp = pkg.DeviceBuffer(E * Kp * 6 // 4)
    _launch(pkg, "vh_launch_conv_weight_planes_parts", None, d[1].ptr, d_wp.ptr, E, 3, P, 3)
    need = n * (T - 1) * Kp * 6
    d_ws, d_tok = pkg.DeviceBuffer(need // 4), pkg.DeviceBuffer(n * T * E)
    assert L.vh_launch_patch_embed_planes3(None, d[0].ptr, d_wp.ptr, d[2].ptr, d[3].ptr, d[4].ptr, d_tok.ptr, n, 3, 224, P, E,
                                           d_ws.ptr, need - 16) != 0          # workspace too small
    _launch(pkg, "vh_launch_patch_embed_planes3", None, d[0].ptr, d_wp.ptr, d[2].ptr, d[3].ptr, d[4].ptr, d_tok.ptr, n, 3, 224, P, E,
            d_ws.ptr, need)
    got = d_tok.to_numpy((n, T, E))
    for i in sorted({0, n // 2, n - 1}):
        want = orc.tokens_from_conv(orc.conv2d(imgs[i], W[1], W[2]), W[0], W[3])
        assert np.abs(got[i] - want).max() <= OP_TOL, f"image {i}"
    ws2 = L.vh_patch_embed_workspace(n, 3, 224, P, E)
    d_ws2, d_tok2 = pkg.DeviceBuffer(max(ws2 // 4, 4)), pkg.DeviceBuffer(n * T * E)
    _launch(pkg, "vh_launch_patch_embed_ws", None, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, d[4].ptr, d_tok2.ptr, n, 3, 224, P, E,
            d_ws2.ptr, ws2)
    assert np.array_equal(got, d_tok2.to_numpy((n, T, E)))


def test_fc1_gelu_of_the_reduced_modes_holds_its_error_bound_for_every_x(pkg, device, oracle):
    """The format-matched GELU of the bf16 mode's fc1 epilogue (gelu_lowp2<0>, csrc/gemm_common.h) through the GEMM with an
    identity weight: outputs for x from -100 to 100 against erf-GELU (ViT_seq.c:283-287) -- within the stated absolute
    1.8e-5 plus one bf16 rounding of the result EVERYWHERE, also far outside the fitted range (the clamp of the
    polynomial's argument used to leave -0.5 |x| erfc(3.5): the error grew linearly with |x|)."""
    from math import erf
    M, K = 512, 128
    xs = np.concatenate([np.linspace(-100.0, 100.0, 40001), np.linspace(-6.0, 6.0, 24000), [0.0] * 1535]).astype(np.float32)
    xs = _bf16_rne(xs).reshape(M, K)                     # exactly representable operands: the product with 1.0 is exact
    w = np.eye(K, dtype=np.float32)
    d_x, d_w, d_b = _dev(pkg, xs), _dev(pkg, w), _dev(pkg, np.zeros(K, np.float32))
    d_x1, d_w1, d_o1 = pkg.DeviceBuffer(M * K // 2), pkg.DeviceBuffer(K * K // 2), pkg.DeviceBuffer(M * K // 2)
    _launch(pkg, "vh_launch_split_rows", None, d_x.ptr, d_x1.ptr, M, K, 1)
    _launch(pkg, "vh_launch_split_rows", None, d_w.ptr, d_w1.ptr, K, K, 1)
    _launch(pkg, "vh_launch_linear_planes", None, d_o1.ptr, 1, d_w1.ptr, d_x1.ptr, 1, d_b.ptr, M, K, K, 1, None)
    got = _planes1_to_f32(d_o1, M, K).astype(np.float64)
    x64 = xs.astype(np.float64)
    want = 0.5 * x64 * (1.0 + np.vectorize(erf)(x64 / np.sqrt(2.0)))
    err = np.abs(got - want)
    bound = 1.8e-5 + 2.0 ** -8 * np.abs(want)
    worst = np.unravel_index(np.argmax(err - bound), err.shape)
    assert (err <= bound).all(), f"x = {xs[worst]}: got {got[worst]}, want {want[worst]}"
    assert np.abs(got[xs <= -8.0]).max() <= 1e-6          # the negative tail is zero, not -0.5 |x| erfc(3.5)
