"""numpy statement of the block-scaled fp8 ("MX", OCP Microscaling v1.0) format csrc/gemm_mx.hip consumes:
e4m3 elements (tests/fp8_ref.py) with one e8m0 power-of-two scale per 32 consecutive elements of a row.

    scale exponent  E = ceil(log2(max |block| / 448))    (the smallest power of two that brings the block inside
                                                           e4m3's range: floor(log2 max) - 8, plus one where the maximum's
                                                           significand exceeds 1.75 -- OCP MX v1.0's plain floor(...) - 8 would
                                                           let those maxima saturate; clamped to >= -126; all-zero or
                                                           subnormal-maximum blocks take -126)
    scale byte        = E + 127
    elements          = e4m3_nearest_even_saturating(x * 2^-E)

Storage of a tensor [rows][cols], cols % 128 == 0:
    values[cols/128][rows][128]   scales[cols/128][4][rows], scales[k][g][r] = scale of block 2 (g & 1) + (g >> 1)
of row r's K step k (the order the four lane groups of v_mfma_scale_f32_16x16x128_f8f6f4 take them in)."""
import numpy as np

import fp8_ref

GROUP_OF_BLOCK = [0, 2, 1, 3]          # block b is stored in lane-group slot 2 (b & 1) + (b >> 1)

# ACTIVATION tensors carry the same scale bytes in another order (csrc/vit_kernels.h mx_act_scale_index):
#     act_scales[ceil(K/512)][4 lane groups][rows][4]   -- K steps in groups of four; the GEMM's lane takes its scales
# of four K steps with one dword load.  Bytes of K steps beyond K/128 in the last group are unspecified.


def act_scale_bytes(rows: int, cols: int) -> int:
    return ((cols // 128 + 3) // 4) * 16 * rows


def to_act_layout(scales: np.ndarray) -> np.ndarray:
    """scales[K/128][4][rows] (the weights' order, what quantize() returns) -> flat uint8 in the activation order"""
    ks, _, rows = scales.shape
    out = np.zeros(((ks + 3) // 4, 4, rows, 4), dtype=np.uint8)
    for k in range(ks):
        out[k >> 2, :, :, k & 3] = scales[k]
    return out.ravel()


def from_act_layout(raw: np.ndarray, rows: int, cols: int) -> np.ndarray:
    """flat uint8 in the activation order -> scales[K/128][4][rows] (padding bytes dropped)"""
    ks = cols // 128
    a = np.asarray(raw, dtype=np.uint8)[:act_scale_bytes(rows, cols)].reshape((ks + 3) // 4, 4, rows, 4)
    return np.stack([a[k >> 2, :, :, k & 3] for k in range(ks)])


def quantize(x: np.ndarray):
    """fp32 [rows][cols] -> (values uint8 [cols/128][rows][128], scales uint8 [cols/128][4][rows])."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    rows, cols = x.shape
    assert cols % 128 == 0
    blocks = x.reshape(rows, cols // 32, 32)
    amax = np.abs(blocks).max(axis=2)
    bits = amax.view(np.uint32)
    expo = ((bits >> 23) & 0xff).astype(np.int32) - 127 - 8 + ((bits & 0x7fffff) > 0x600000)   # significand > 1.75: one up
    expo = np.maximum(expo, -126)
    mult = np.ldexp(np.float32(1.0), -expo).astype(np.float32)                  # 2^-E, exact
    q = fp8_ref.quantize((blocks * mult[:, :, None]).astype(np.float32)).reshape(rows, cols // 128, 128)
    values = np.ascontiguousarray(q.transpose(1, 0, 2))
    sb = (expo + 127).astype(np.uint8).reshape(rows, cols // 128, 4)             # [row][K step][block]
    scales = np.empty((cols // 128, 4, rows), dtype=np.uint8)
    for b in range(4):
        scales[:, GROUP_OF_BLOCK[b], :] = sb[:, :, b].T
    return values, scales


def dequantize(values: np.ndarray, scales: np.ndarray) -> np.ndarray:
    """-> float32 [rows][cols] (exact: e4m3 value times a power of two)."""
    ks, rows, _ = values.shape
    v = fp8_ref.dequantize(values).reshape(ks, rows, 4, 32)
    out = np.empty((rows, ks, 4, 32), dtype=np.float32)
    for b in range(4):
        e = scales[:, GROUP_OF_BLOCK[b], :].astype(np.int32) - 127               # [K step][row]
        out[:, :, b, :] = (v[:, :, b, :] * np.ldexp(np.float32(1.0), e)[:, :, None]).transpose(1, 0, 2)
    return out.reshape(rows, ks * 128)
