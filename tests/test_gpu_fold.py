"""GPU tests of the LayerNorm fold of the reduced-precision modes (csrc/norm_fold.h; replaces the `layerNorm` launches
layer_norm.cl:3-53 in front of the QKV projection and fc1 for those modes):

    LN(x) W^T + b = rstd (x (gamma.W)^T - mean colsum(gamma.W)) + (beta W^T + b)

Checked per operator: the producers (output projection / fc2 / patch embedding) leave the SAME fp32 rows as their unfolded
twins plus those rows rounded to the next operand format (bit for bit: bf16 nearest-even, or the numpy statement of the MX
quantiser) plus per-row, per-128-column partial sums; the consumers reproduce (a) a float64 evaluation of the folded
formula on the same rounded operands at the fp32 operator tolerance and (b) the CPU statement of LayerNorm followed by the
projection (ViT_seq.c:120-142, 295-309) at the precision the rounded operands leave.  Per model: the folded path stays
inside the modes' own tolerances, one LayerNorm launch per forward remains, and max |mean| / std of the residual rows --
the cancellation the fold is exposed to -- is measured and printed."""
import numpy as np
import pytest

import mx_ref
from test_gpu_p3 import OP_TOL, _bf16_rne, _dev, _launch, _planes1_to_f32, _planes_f16_to_f32

pytestmark = pytest.mark.gpu

EPS = 1e-6


def _ln_rows(x, gamma, beta, eps=EPS):
    """layer_norm_seq (ViT_seq.c:120-142) in float64 (the fp32 statement is within 1e-6 of it)"""
    x = x.astype(np.float64)
    mean = x.mean(1, keepdims=True)
    var = (x * x).mean(1, keepdims=True) - mean * mean
    return (x - mean) / np.sqrt(var + eps) * gamma.astype(np.float64) + beta.astype(np.float64)


def _folded_reference(x_oper, stats_rows, w_oper, colsum, bias_f, K, eps=EPS):
    """float64 evaluation of what the consuming kernel computes: operands as rounded, row terms from the fp32 partials"""
    s = stats_rows.astype(np.float32)
    lane = []                          # fixed order, fp32, as row_norm_terms: lane q adds partials q, q+4, q+8, q+12 ...
    for q in range(4):
        acc = np.zeros_like(s[0])
        for g in range(q, s.shape[0], 4):
            acc = acc + s[g]
        lane.append(acc)
    tot = (lane[0] + lane[1]) + (lane[2] + lane[3])     # ... and two shuffles add the four lane sums
    mean = tot[:, 0] / np.float32(K)
    var = tot[:, 1] / np.float32(K) - mean * mean
    rstd = (np.float32(1.0) / np.sqrt((var.astype(np.float64) + eps).astype(np.float32))).astype(np.float64)
    acc = x_oper.astype(np.float64) @ w_oper.astype(np.float64).T
    return rstd[:, None] * acc - (rstd * mean.astype(np.float64))[:, None] * colsum.astype(np.float64)[None, :] + bias_f.astype(np.float64)[None, :]


def _gelu64(v):
    from math import erf
    return 0.5 * v * (1.0 + np.vectorize(erf)(v / np.sqrt(2.0)))


def _residual_rows(oracle, M, E, seed):
    """fp32 rows with per-row offsets and scales: mean/std between -1 and 1, a few rows well outside"""
    x = oracle.synth_fill(M * E, seed, 1.0, 0.0).reshape(M, E)
    off = oracle.synth_fill(M, seed + 1, 0.6, 0.0)
    sc = 0.5 + np.abs(oracle.synth_fill(M, seed + 2, 2.0, 0.0))
    x = (x * sc[:, None] + off[:, None]).astype(np.float32)
    if M > 4:
        x[1] += 3.0            # |mean| / std ~ 5: the cancellation case
        x[3, 7] = 40.0         # one dominant channel
    return x


@pytest.mark.parametrize("M,K,N", [(300, 768, 768), (300, 3072, 768), (19700, 768, 768), (131, 128, 256)])
def test_bf16_producer_leaves_rows_operand_and_partial_sums(pkg, device, oracle, M, K, N):
    """vh_launch_linear_planes_resid_norm = vh_launch_linear_planes(residual) for the fp32 rows (same k order: the same bits
    up to the column permutation, which does not change any sum) + bf16(rows) as planes + partial sums per 128 columns."""
    a = oracle.synth_fill(M * K, 800 + M, 1.0, 0.1).reshape(M, K)
    w = oracle.synth_fill(N * K, 801 + N, 0.04, 0.0)
    b = oracle.synth_fill(N, 802, 0.1, 0.0)
    r = _residual_rows(oracle, M, N, 803)
    d_a, d_w, d_b = _dev(pkg, a), _dev(pkg, w), _dev(pkg, b)
    d_w1, d_a1 = pkg.DeviceBuffer((N * K + 1) // 2), pkg.DeviceBuffer((M * K + 1) // 2)
    _launch(pkg, "vh_launch_split_rows", None, d_w.ptr, d_w1.ptr, N, K, 1)
    _launch(pkg, "vh_launch_split_rows", None, d_a.ptr, d_a1.ptr, M, K, 1)
    d_plain, d_x = _dev(pkg, r), _dev(pkg, r)
    _launch(pkg, "vh_launch_linear_planes", None, d_plain.ptr, 0, d_w1.ptr, d_a1.ptr, 1, d_b.ptr, M, K, N, 0, d_plain.ptr)
    d_op, d_st = pkg.DeviceBuffer((M * N + 1) // 2), pkg.DeviceBuffer((N // 128) * M * 2)
    _launch(pkg, "vh_launch_linear_planes_resid_norm", None, d_x.ptr, d_w1.ptr, d_a1.ptr, d_b.ptr, d_x.ptr, M, K, N, d_op.ptr, None,
            d_st.ptr)
    x = d_x.to_numpy((M, N))
    assert np.array_equal(x, d_plain.to_numpy((M, N)))                    # the fp32 rows: bit-identical to the unfolded launch
    assert np.array_equal(_planes1_to_f32(d_op, M, N), _bf16_rne(x))      # the operand: those rows, nearest-even bf16
    st = d_st.to_numpy((N // 128, M, 2))
    x64 = x.astype(np.float64).reshape(M, N // 128, 128)
    assert np.abs(st[:, :, 0].T - x64.sum(2)).max() <= 1e-5 * max(1.0, np.abs(x64).sum(2).max())
    assert np.abs(st[:, :, 1].T - (x64 * x64).sum(2)).max() <= 1e-5 * (x64 * x64).sum(2).max()


@pytest.mark.parametrize("M,N,gelu,out", [(300, 2304, 0, 2), (300, 3072, 1, 1), (300, 2304, 0, 0), (19700, 3072, 1, 1), (19700, 2304, 0, 2)])
def test_bf16_consumer_applies_the_folded_layernorm(pkg, device, oracle, weights, M, N, gelu, out):
    """vh_launch_linear_planes_norm on bf16(x) planes, gamma-scaled weight planes and the column terms made by the
    context-creation helpers, against float64 on the same operands (operator tolerance) and against LayerNorm -> Linear
    on the fp32 values (bf16 operand precision, scaled by the row's |x| / |x - mean|)."""
    K = 768
    x = _residual_rows(oracle, M, K, 810 + M)
    gamma = (1.0 + oracle.synth_fill(K, 811, 0.3, 0.0)).astype(np.float32)
    beta = oracle.synth_fill(K, 812, 0.2, 0.0)
    w = oracle.synth_fill(N * K, 813 + N, 0.04, 0.0).reshape(N, K)
    b = oracle.synth_fill(N, 814, 0.1, 0.0)
    # the producer's outputs, made here by its building blocks: planes of bf16(x) and the partial sums
    d_x, d_w, d_b, d_g, d_be = (_dev(pkg, v) for v in (x, w, b, gamma, beta))
    d_x1 = pkg.DeviceBuffer((M * K + 1) // 2)
    _launch(pkg, "vh_launch_split_rows", None, d_x.ptr, d_x1.ptr, M, K, 1)
    xg = x.astype(np.float32).reshape(M, K // 128, 128)
    stats = np.stack([xg.sum(2, dtype=np.float32), (xg * xg).sum(2, dtype=np.float32)], axis=2).transpose(1, 0, 2)   # [K/128][M][2]
    d_st = _dev(pkg, stats)
    # context-creation side: gamma into W, rounded; colsum of the rounded values; folded bias
    d_ws, d_w1 = pkg.DeviceBuffer(N * K), pkg.DeviceBuffer((N * K + 1) // 2)
    d_cs, d_bf = pkg.DeviceBuffer(N), pkg.DeviceBuffer(N)
    _launch(pkg, "vh_launch_fold_gamma", None, d_w.ptr, d_g.ptr, d_ws.ptr, N, K)
    assert np.array_equal(d_ws.to_numpy((N, K)), w * gamma[None, :])
    _launch(pkg, "vh_launch_split_rows", None, d_ws.ptr, d_w1.ptr, N, K, 1)
    _launch(pkg, "vh_launch_colsum_operand", None, d_w1.ptr, None, d_cs.ptr, N, K)
    _launch(pkg, "vh_launch_fold_bias", None, d_w.ptr, d_be.ptr, d_b.ptr, d_bf.ptr, N, K)
    wr = _planes1_to_f32(d_w1, N, K)
    cs, bf = d_cs.to_numpy(), d_bf.to_numpy()
    assert np.abs(cs - wr.astype(np.float64).sum(1)).max() <= 1e-6 * np.abs(wr).sum(1).max()
    assert np.abs(bf - (b + w.astype(np.float64) @ beta.astype(np.float64))).max() <= 1e-6
    if out == 0:
        d_o = pkg.DeviceBuffer(M * N)
    else:
        d_o = pkg.DeviceBuffer((M * N + 1) // 2)
    _launch(pkg, "vh_launch_linear_planes_norm", None, d_o.ptr, out, d_w1.ptr, d_x1.ptr, d_st.ptr, d_cs.ptr, d_bf.ptr, EPS, M, K, N, gelu)
    got = d_o.to_numpy((M, N)) if out == 0 else _planes1_to_f32(d_o, M, N) if out == 1 else _planes_f16_to_f32(d_o, M, N)
    rows = np.arange(M) if M <= 300 else np.r_[0:24, 240:272, 4090:4102, M - 24:M]
    want = _folded_reference(_bf16_rne(x[rows]), stats[:, rows], wr, cs, bf, K)
    if gelu:
        want = _gelu64(want)
    out_round = {0: 0.0, 1: 2.0 ** -8, 2: 2.0 ** -10}[out] * np.abs(want).max()
    assert np.abs(got[rows] - want).max() <= 4 * OP_TOL + out_round
    # against the unfolded statement on the fp32 values: bf16 operand noise, amplified where |mean| >> std
    ref = _ln_rows(x[rows], gamma, beta) @ w.astype(np.float64).T + b
    if gelu:
        ref = _gelu64(ref)
    xr = x[rows].astype(np.float64)
    amp = np.sqrt((xr * xr).mean(1)) / xr.std(1)
    print(f"\nfold vs LayerNorm->Linear (bf16): max |mean|/std of the test rows {np.abs(xr.mean(1) / xr.std(1)).max():.2f}, "
          f"max error {np.abs(got[rows] - ref).max():.3e}")
    assert (np.abs(got[rows] - ref).max(1) <= (2.0 ** -7 * amp + 2.0 ** -8) * max(1.0, np.abs(ref).max())).all()


def _act_buf(pkg, rows, cols):
    return pkg.DeviceBuffer(mx_ref.act_scale_bytes(rows, cols) // 4 + 4)


@pytest.mark.parametrize("M,E,N", [(1, 128, 128), (17, 256, 384), (129, 384, 128), (255, 640, 256), (513, 1280, 384), (1000, 2048, 256)])
def test_bf16_producer_then_consumer_on_ragged_shapes(pkg, device, oracle, M, E, N):
    """The two halves of the fold chained as the model chains them, on shapes that are not ViT-B/16's: a residual producer
    (K = E) leaves x, bf16(x) and 1 .. 16 partial sums per row (E = 128 .. 2048: one to four partials per lane of a row,
    lanes without any), the consumer (N columns) applies them; M from a single row to ragged multiples of the tiles.
    Against float64 of the folded formula on the rounded operands, and against LayerNorm -> Linear of the fp32 rows."""
    a = oracle.synth_fill(M * E, 900 + M, 1.0, 0.1).reshape(M, E)
    w1 = oracle.synth_fill(E * E, 901 + E, 0.04, 0.0).reshape(E, E)
    b1 = oracle.synth_fill(E, 902, 0.1, 0.0)
    r = _residual_rows(oracle, M, E, 903)
    gamma = (1.0 + oracle.synth_fill(E, 904, 0.3, 0.0)).astype(np.float32)
    beta = oracle.synth_fill(E, 905, 0.2, 0.0)
    w2 = oracle.synth_fill(N * E, 906 + N, 0.04, 0.0).reshape(N, E)
    b2 = oracle.synth_fill(N, 907, 0.1, 0.0)
    d_a, d_w1, d_b1, d_x, d_g, d_be, d_w2, d_b2 = (_dev(pkg, v) for v in (a, w1, b1, r, gamma, beta, w2, b2))
    d_a1, d_w11 = pkg.DeviceBuffer((M * E + 1) // 2 + 4), pkg.DeviceBuffer((E * E + 1) // 2)
    _launch(pkg, "vh_launch_split_rows", None, d_a.ptr, d_a1.ptr, M, E, 1)
    _launch(pkg, "vh_launch_split_rows", None, d_w1.ptr, d_w11.ptr, E, E, 1)
    d_op, d_st = pkg.DeviceBuffer((M * E + 1) // 2 + 4), pkg.DeviceBuffer((E // 128) * M * 2)
    _launch(pkg, "vh_launch_linear_planes_resid_norm", None, d_x.ptr, d_w11.ptr, d_a1.ptr, d_b1.ptr, d_x.ptr, M, E, E, d_op.ptr, None, d_st.ptr)
    x = d_x.to_numpy((M, E))
    assert np.array_equal(_planes1_to_f32(d_op, M, E), _bf16_rne(x))
    stats = d_st.to_numpy((E // 128, M, 2))
    d_ws, d_w21, d_cs, d_bf = pkg.DeviceBuffer(N * E), pkg.DeviceBuffer((N * E + 1) // 2), pkg.DeviceBuffer(N), pkg.DeviceBuffer(N)
    _launch(pkg, "vh_launch_fold_gamma", None, d_w2.ptr, d_g.ptr, d_ws.ptr, N, E)
    _launch(pkg, "vh_launch_split_rows", None, d_ws.ptr, d_w21.ptr, N, E, 1)
    _launch(pkg, "vh_launch_colsum_operand", None, d_w21.ptr, None, d_cs.ptr, N, E)
    _launch(pkg, "vh_launch_fold_bias", None, d_w2.ptr, d_be.ptr, d_b2.ptr, d_bf.ptr, N, E)
    d_o = pkg.DeviceBuffer(M * N)
    _launch(pkg, "vh_launch_linear_planes_norm", None, d_o.ptr, 0, d_w21.ptr, d_op.ptr, d_st.ptr, d_cs.ptr, d_bf.ptr, EPS, M, E, N, 0)
    got = d_o.to_numpy((M, N))
    want = _folded_reference(_bf16_rne(x), stats, _planes1_to_f32(d_w21, N, E), d_cs.to_numpy(), d_bf.to_numpy(), E)
    assert np.abs(got - want).max() <= 4 * OP_TOL * max(1.0, np.abs(want).max())
    ref = _ln_rows(x, gamma, beta) @ w2.astype(np.float64).T + b2
    xr = x.astype(np.float64)
    amp = np.sqrt((xr * xr).mean(1)) / np.maximum(xr.std(1), 1e-3)
    assert (np.abs(got - ref).max(1) <= (2.0 ** -7 * amp + 2.0 ** -8) * max(1.0, np.abs(ref).max())).all()


def _mx_dev(pkg, x, act):
    """fp32 rows -> (values, scales) device buffers through the library's quantiser -- act: an activation tensor (a GEMM's A
    operand, scale bytes in the activation order), else a weight matrix -- and their numpy dequantisation"""
    M, K = x.shape
    d_x = _dev(pkg, x)
    d_v, d_s = pkg.DeviceBuffer(M * K // 4), _act_buf(pkg, M, K) if act else pkg.DeviceBuffer((M * K // 32 + 3) // 4 + 4)
    _launch(pkg, "vh_launch_quantize_mx_act" if act else "vh_launch_quantize_mx_rows", None, d_x.ptr, d_v.ptr, d_s.ptr, M, K)
    v = d_v.to_numpy().view(np.uint8)[:M * K].reshape(K // 128, M, 128)
    raw = d_s.to_numpy().view(np.uint8)
    s = mx_ref.from_act_layout(raw, M, K) if act else raw[:M * K // 32].reshape(K // 128, 4, M)
    return d_v, d_s, mx_ref.dequantize(v, s)


@pytest.mark.parametrize("M,K,N", [(300, 768, 768), (300, 3072, 768), (19700, 768, 768)])
def test_mx_producer_leaves_rows_operand_and_partial_sums(pkg, device, oracle, M, K, N):
    a = oracle.synth_fill(M * K, 820 + M, 1.0, 0.1).reshape(M, K)
    w = oracle.synth_fill(N * K, 821 + N, 0.04, 0.0).reshape(N, K)
    b = oracle.synth_fill(N, 822, 0.1, 0.0)
    r = _residual_rows(oracle, M, N, 823)
    d_av, d_as, _ = _mx_dev(pkg, a, True)
    d_wv, d_ws, _ = _mx_dev(pkg, w, False)
    d_b, d_plain, d_x = _dev(pkg, b), _dev(pkg, r), _dev(pkg, r)
    _launch(pkg, "vh_launch_linear_mx", None, d_plain.ptr, None, d_wv.ptr, d_ws.ptr, d_av.ptr, d_as.ptr, d_b.ptr, M, K, N, 0, d_plain.ptr)
    d_ov, d_os = pkg.DeviceBuffer(M * N // 4), _act_buf(pkg, M, N)
    d_st = pkg.DeviceBuffer((N // 128) * M * 2)
    _launch(pkg, "vh_launch_linear_mx_resid_norm", None, d_x.ptr, d_wv.ptr, d_ws.ptr, d_av.ptr, d_as.ptr, d_b.ptr, d_x.ptr, M, K, N,
            d_ov.ptr, d_os.ptr, d_st.ptr)
    x = d_x.to_numpy((M, N))
    assert np.array_equal(x, d_plain.to_numpy((M, N)))
    qv, qs = mx_ref.quantize(x)                                           # the operand: the numpy quantiser, byte for byte
    assert np.array_equal(d_ov.to_numpy().view(np.uint8)[:M * N].reshape(N // 128, M, 128), qv)
    assert np.array_equal(mx_ref.from_act_layout(d_os.to_numpy().view(np.uint8), M, N), qs)
    st = d_st.to_numpy((N // 128, M, 2))
    x64 = x.astype(np.float64).reshape(M, N // 128, 128)
    assert np.abs(st[:, :, 0].T - x64.sum(2)).max() <= 1e-5 * max(1.0, np.abs(x64).sum(2).max())
    assert np.abs(st[:, :, 1].T - (x64 * x64).sum(2)).max() <= 1e-5 * (x64 * x64).sum(2).max()


@pytest.mark.parametrize("M,N,gelu,kind", [(300, 2304, 0, 2), (300, 3072, 1, 1), (300, 2304, 0, 0), (19700, 3072, 1, 1)])
def test_mx_consumer_applies_the_folded_layernorm(pkg, device, oracle, M, N, gelu, kind):
    K = 768
    x = _residual_rows(oracle, M, K, 830 + M)
    gamma = (1.0 + oracle.synth_fill(K, 831, 0.3, 0.0)).astype(np.float32)
    beta = oracle.synth_fill(K, 832, 0.2, 0.0)
    w = oracle.synth_fill(N * K, 833 + N, 0.04, 0.0).reshape(N, K)
    b = oracle.synth_fill(N, 834, 0.1, 0.0)
    d_xv, d_xs, xq = _mx_dev(pkg, x, True)
    xg = x.reshape(M, K // 128, 128)
    stats = np.stack([xg.sum(2, dtype=np.float32), (xg * xg).sum(2, dtype=np.float32)], axis=2).transpose(1, 0, 2)
    d_st = _dev(pkg, stats)
    d_wv, d_wsc, wq = _mx_dev(pkg, (w * gamma[None, :]).astype(np.float32), False)
    d_w, d_b, d_be = _dev(pkg, w), _dev(pkg, b), _dev(pkg, beta)
    d_cs, d_bf = pkg.DeviceBuffer(N), pkg.DeviceBuffer(N)
    _launch(pkg, "vh_launch_colsum_operand", None, d_wv.ptr, d_wsc.ptr, d_cs.ptr, N, K)
    _launch(pkg, "vh_launch_fold_bias", None, d_w.ptr, d_be.ptr, d_b.ptr, d_bf.ptr, N, K)
    cs, bf = d_cs.to_numpy(), d_bf.to_numpy()
    assert np.abs(cs - wq.astype(np.float64).sum(1)).max() <= 1e-6 * np.abs(wq).sum(1).max()     # of the DEQUANTISED weights
    if kind == 1:
        d_o, d_os = pkg.DeviceBuffer(M * N // 4), _act_buf(pkg, M, N)
    elif kind == 2:
        d_o, d_os = pkg.DeviceBuffer((M * N + 1) // 2), None
    else:
        d_o, d_os = pkg.DeviceBuffer(M * N), None
    _launch(pkg, "vh_launch_linear_mx_norm", None, d_o.ptr, d_os.ptr if d_os else None, kind, d_wv.ptr, d_wsc.ptr, d_xv.ptr, d_xs.ptr,
            d_st.ptr, d_cs.ptr, d_bf.ptr, EPS, M, K, N, gelu)
    rows = np.arange(M) if M <= 300 else np.r_[0:24, 240:272, 4090:4102, M - 24:M]
    want = _folded_reference(xq[rows], stats[:, rows], wq, cs, bf, K)
    if gelu:
        want = _gelu64(want)
    if kind == 1:
        got = mx_ref.dequantize(d_o.to_numpy().view(np.uint8)[:M * N].reshape(N // 128, M, 128),
                                mx_ref.from_act_layout(d_os.to_numpy().view(np.uint8), M, N))
        tol = 2.0 ** -3 * np.abs(want).max()          # one e4m3 rounding (3 significand bits, block-scaled) + the format-matched GELU
    else:
        # v_mfma_scale_f32_16x16x128_f8f6f4 sums the 128 products of an instruction in a fixed-point frame hung on the largest
        # one: a row with one dominant channel (x = 40 here) loses the low bits of its small products -- measured 2^-13 of
        # the row's largest |x w| per instruction (the unfolded path sees the same loss on LayerNorm outputs; uniform
        # operands do not show it: test_linear_mx_vs_float64_products_of_the_dequantised_operands).  Scaled by the row's 1/std.
        s32 = stats[:, rows].sum(0)
        mean = s32[:, 0] / K
        rstd = 1.0 / np.sqrt(np.maximum(s32[:, 1] / K - mean * mean, 0.0) + EPS)
        big = rstd * np.abs(xq[rows]).max(1) * np.abs(wq).max()
        tol_rows = 4 * OP_TOL + 2.0 ** -11 * big
        if kind == 2:
            got, tol_rows = _planes_f16_to_f32(d_o, M, N), tol_rows + 2.0 ** -10 * np.abs(want).max()
        else:
            got = d_o.to_numpy((M, N))
        err = np.abs(got[rows] - want).max(1)
        worst = int(np.argmax(err / tol_rows))
        print(f"\nMX fold consumer kind {kind}: worst row {rows[worst]} error {err[worst]:.3e} (bound {tol_rows[worst]:.3e}, largest |x w| "
              f"scaled {big[worst]:.2f}); median row error {np.median(err):.3e}")
        assert (err <= tol_rows).all()
        return
    assert np.abs(got[rows] - want).max() <= tol


@pytest.mark.parametrize("preset,n,mx", [("vit_b_16", 3, False), ("vit_b_16", 30, True), ("vit_h_14", 2, False), ("vit_h_14", 2, True)])
def test_patch_embedding_leaves_the_first_operand_and_sums_for_every_token_row(pkg, device, preset, n, mx):
    """vh_launch_patch_embed_planes_norm: the token rows of vh_launch_patch_embed_planes (bit for bit), and -- class-token
    rows included -- those rows as bf16 planes / MX tensor and their partial sums."""
    from oracle.oracle import Oracle
    orc = Oracle(preset)
    cfg = pkg.preset(preset)
    E, T, P = cfg.embed_dim, pkg.binding.tokens(cfg), cfg.patch_size
    W = [orc.synth_fill(orc.tensor_size(i), 40 + i, 0.05, 0.0) for i in range(4)]
    imgs = pkg.synth_images(cfg, 20, n)
    L = pkg.lib()
    Kp = L.vh_patch_planes_k(3, P)
    d = [_dev(pkg, a) for a in (imgs, W[1], W[2], W[0], W[3])]
    d_wp = pkg.DeviceBuffer(E * Kp // 2)
    _launch(pkg, "vh_launch_conv_weight_planes", None, d[1].ptr, d_wp.ptr, E, 3, P)
    need = n * (T - 1) * Kp * 2
    d_ws, d_plain, d_tok = pkg.DeviceBuffer(need // 4), pkg.DeviceBuffer(n * T * E), pkg.DeviceBuffer(n * T * E)
    _launch(pkg, "vh_launch_patch_embed_planes", None, d[0].ptr, d_wp.ptr, d[2].ptr, d[3].ptr, d[4].ptr, d_plain.ptr, n, 3, 224, P, E,
            d_ws.ptr, need)
    R = n * T
    d_op = pkg.DeviceBuffer(R * E // 4 if mx else (R * E + 1) // 2)
    d_os = _act_buf(pkg, R, E) if mx else None
    d_st = pkg.DeviceBuffer((E // 128) * R * 2)
    _launch(pkg, "vh_launch_patch_embed_planes_norm", None, d[0].ptr, d_wp.ptr, d[2].ptr, d[3].ptr, d[4].ptr, d_tok.ptr, n, 3, 224, P, E,
            d_ws.ptr, need, d_op.ptr, d_os.ptr if mx else None, d_st.ptr)
    x = d_tok.to_numpy((R, E))
    assert np.array_equal(x, d_plain.to_numpy((R, E)))
    if mx:
        qv, qs = mx_ref.quantize(x)
        assert np.array_equal(d_op.to_numpy().view(np.uint8)[:R * E].reshape(E // 128, R, 128), qv)
        assert np.array_equal(mx_ref.from_act_layout(d_os.to_numpy().view(np.uint8), R, E), qs)
    else:
        assert np.array_equal(_planes1_to_f32(d_op, R, E), _bf16_rne(x))
    st = d_st.to_numpy((E // 128, R, 2))
    x64 = x.astype(np.float64).reshape(R, E // 128, 128)
    assert np.abs(st[:, :, 0].T - x64.sum(2)).max() <= 1e-5 * max(1.0, np.abs(x64).sum(2).max())
    assert np.abs(st[:, :, 1].T - (x64 * x64).sum(2)).max() <= 1e-5 * (x64 * x64).sum(2).max()


def _model_logits(pkg, cfg, weights, images, precision, fold, monkeypatch, tokens_out=None):
    monkeypatch.setenv("VIT_HIP_LN_FOLD", "1" if fold else "0")
    m = pkg.ViTHip(cfg, weights, device=0, max_batch=images.shape[0], precision=precision)
    assert bool(pkg.lib().vit_hip_ln_fold(m.ctx)) == fold
    m.profile_enable(1)
    logits, _ = m.forward(images)
    prof = m.profile_read()
    if tokens_out is not None:
        tokens_out.append(m.read_tokens(images.shape[0]))
    m.close()
    return logits, prof


@pytest.mark.parametrize("precision,bound", [("bf16", 2e-2), ("fp8", 0.35)])
def test_model_with_folded_layernorms_one_launch_left_and_inside_the_modes_tolerance(pkg, device, weights, golden_full, precision, bound,
                                                                                      monkeypatch):
    """ViT-B/16, 4 golden images: the folded path (default) against the separate-LayerNorm path of the same precision
    and against the reference's logits: 25 LayerNorm launches become 1, the logit error against ViT_seq.c stays inside
    the mode's stated tolerance, and the cancellation exposure max |mean| / std of the residual rows is printed."""
    cfg = pkg.preset("vit_b_16")
    images = np.stack([pkg.synth_images(cfg, i, 1)[0] for i in range(4)])
    toks = []
    lf, pf = _model_logits(pkg, cfg, weights, images, precision, True, monkeypatch, toks)
    lu, pu = _model_logits(pkg, cfg, weights, images, precision, False, monkeypatch)
    assert pf["layer_norm"][1] == 1 and pu["layer_norm"][1] == 25
    assert all(pf[k][1] == pu[k][1] for k in ("qkv_gemm", "attention", "out_proj_gemm", "fc1_gemm", "fc2_gemm", "head_gemm"))
    want = golden_full["logits"][:4]
    ef, eu = np.abs(lf - want).max(), np.abs(lu - want).max()
    x = toks[0].reshape(-1, cfg.embed_dim).astype(np.float64)
    ratio = np.abs(x.mean(1)) / x.std(1)
    print(f"\n{precision}: max |dlogit| vs ViT_seq.c folded {ef:.3e}, separate LayerNorms {eu:.3e}; folded vs separate "
          f"{np.abs(lf - lu).max():.3e}; residual rows after the last layer: max |mean|/std {ratio.max():.3f}, median {np.median(ratio):.3f}")
    if precision == "bf16":
        assert ef <= 4e-2                                   # the mode's stated tolerance, unchanged
    else:
        rel = np.linalg.norm(lf - want) / np.linalg.norm(want)
        assert rel <= 0.15                                  # the mode's stated tolerance, unchanged
    assert np.abs(lf - lu).max() <= bound


def test_fp32_path_with_folded_layernorms_lab_variant_keeps_the_parity(pkg, device, weights, golden_full, monkeypatch):
    """$VIT_HIP_LN_FOLD=1 on the fp32 path (a LAB VARIANT, off by default: docs/LABBOOK.md R4.6): the same fold on the exact
    three-part planes -- x itself is split, gamma sits in the three-part weights, QKV and fc1 apply the row terms.  The
    reference's goldens (4 synthetic images and its one real image) within the north star's 1e-4 with the same arg-max,
    one LayerNorm launch left; the difference to the default path is printed."""
    cfg = pkg.preset("vit_b_16")
    real = np.load(__import__("pathlib").Path(__file__).resolve().parent / "golden" / "b16_real_image.npz")
    images = np.concatenate([np.stack([pkg.synth_images(cfg, i, 1)[0] for i in range(4)]), real["image"][None]])
    want = np.concatenate([golden_full["logits"][:4], real["logits"]])
    lf, pf = _model_logits(pkg, cfg, weights, images, "f32", True, monkeypatch)
    monkeypatch.delenv("VIT_HIP_LN_FOLD")
    m = pkg.ViTHip(cfg, weights, device=0, max_batch=5)
    assert not pkg.lib().vit_hip_ln_fold(m.ctx)                  # the default fp32 path never folds
    ld, _ = m.forward(images)
    m.close()
    assert pf["layer_norm"][1] == 1
    ef, ed = np.abs(lf - want).max(axis=1), np.abs(ld - want).max(axis=1)
    print(f"\nfp32 path, max |dlogit| vs ViT_seq.c per image (4 synthetic + the real one): folded {ef}, default {ed}; "
          f"folded vs default {np.abs(lf - ld).max():.3e}")
    assert ef.max() <= 1e-4 and np.array_equal(lf.argmax(1), want.argmax(1))


def test_fold_launchers_reject_bad_arguments(pkg, device):
    L = pkg.lib()
    buf = pkg.DeviceBuffer(4096)
    p = buf.ptr
    assert L.vh_launch_linear_planes_norm(None, p, 1, p, p, None, p, p, EPS, 16, 128, 128, 0) != 0          # no row statistics
    assert L.vh_launch_linear_planes_norm(None, p, 2, p, p, p, p, p, EPS, 16, 128, 128, 1) != 0             # GELU into fp16 planes
    assert L.vh_launch_linear_planes_norm(None, p, 1, p, p, p, p, p, EPS, 16, 96, 128, 0) != 0              # K step
    assert L.vh_launch_linear_planes_resid_norm(None, p, p, p, p, p, 16, 128, 128, None, None, p) != 0      # no operand
    assert L.vh_launch_linear_mx_norm(None, p, None, 1, p, p, p, p, p, p, p, EPS, 16, 256, 128, 0) != 0     # MX out without scales
    assert L.vh_launch_linear_mx_resid_norm(None, p, p, p, p, p, p, p, 16, 256, 128, p, None, p) != 0       # no operand scales
    assert L.vh_launch_colsum_operand(None, p, None, p, 16, 48) != 0
    assert b"" != L.vh_last_error()
