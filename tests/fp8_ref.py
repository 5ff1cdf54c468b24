"""numpy statement of OCP e4m3 (e4m3fn: 4 exponent bits, bias 7, 3 mantissa bits, no infinities,
0x7f/0xff = NaN, largest finite 448) used to check the GPU's fp8 casts byte for byte."""
import numpy as np


def _positive_table() -> np.ndarray:
    vals = np.zeros(127, dtype=np.float64)          # codes 0x00..0x7e
    for c in range(127):
        e, m = c >> 3, c & 7
        vals[c] = (m / 8.0) * 2.0 ** -6 if e == 0 else (1.0 + m / 8.0) * 2.0 ** (e - 7)
    return vals


TABLE = _positive_table()
FP8_MAX = 448.0


def quantize(x: np.ndarray) -> np.ndarray:
    """fp32 -> e4m3 bytes: saturate to +-448, round to nearest, ties to the even code."""
    x = np.asarray(x, dtype=np.float32)
    a = np.minimum(np.abs(x.astype(np.float64)), FP8_MAX)
    hi = np.searchsorted(TABLE, a, side="left").clip(0, 126)       # first code with value >= a
    lo = np.maximum(hi - 1, 0)
    d_lo, d_hi = a - TABLE[lo], TABLE[hi] - a
    pick_hi = (d_hi < d_lo) | ((d_hi == d_lo) & (hi % 2 == 0))
    code = np.where(pick_hi, hi, lo).astype(np.uint8)
    return code | (np.signbit(x).astype(np.uint8) << 7)


def dequantize(code: np.ndarray) -> np.ndarray:
    code = np.asarray(code, dtype=np.uint8)
    mag = TABLE[np.minimum(code & 0x7f, 126)].astype(np.float32)
    return np.where(code & 0x80, -mag, mag).astype(np.float32)
