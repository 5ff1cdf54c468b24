#!/usr/bin/env python3
"""Same-box A/B of differently built copies of the library on ONE launcher: the ViT-H/14 attention
(vh_launch_attention_planes_f16_hd80_operand, 256 images, 257 tokens, 16 heads of 80) timed alternately, in one process,
on the same device buffers.  Box-to-box spread of bench.py is +-2 %; differences between kernel variants of that size only
show this way.  Usage: attn_h16_ab.py name=path/to/libvit_hip.so [name=path ...] [kind: 1 = bf16 planes out (default), 2 = MX out] [b16]
`b16`: the head-dim-64 kernel instead (attention_p3.hip, one-part form: ViT-B/16, 512 images, 197 tokens, 12 heads);
`b16f32`: its three-part form, the fp32 path's attention (vh_launch_attention_planes; exact splits of N(0,1) values)."""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as graft  # noqa: E402


def main():
    libs, kind, b16, f32 = [], 1, False, False
    for a in sys.argv[1:]:
        if "=" in a:
            name, path = a.split("=", 1)
            libs.append((name, path))
        elif a in ("b16", "b16f32"):
            b16 = True
            f32 = a == "b16f32"
        else:
            kind = int(a)
    pkg = graft.load_package()
    L0 = pkg.lib()
    assert L0.vh_init(0) == 0, L0.vh_last_error()
    n, T, E, H = (512, 197, 768, 12) if b16 else (256, 257, 1280, 16)
    rows = n * T
    rng = np.random.default_rng(5)
    if f32:
        def bf16_round(x):          # round-to-nearest-even to bf16, returned as float32
            u = x.view(np.uint32).astype(np.uint64)
            return (((u + 0x7fff + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)
        x = rng.standard_normal((3 * E // 32, 1, rows, 32), dtype=np.float32)
        p0 = bf16_round(x)
        p1 = bf16_round(x - p0)
        p2 = bf16_round(x - p0 - p1)
        parts = np.concatenate([p0, p1, p2], axis=1)               # [3E/32][3][rows][32]
        planes = (parts.view(np.uint32) >> 16).astype(np.uint16)
        d_q = pkg.DeviceBuffer.from_numpy(planes.ravel().view(np.float32))
        d_o = pkg.DeviceBuffer(rows * E * 3 // 2)
    else:
        planes = rng.standard_normal((3 * E // 32, rows, 32), dtype=np.float32).astype(np.float16)
        d_q = pkg.DeviceBuffer.from_numpy(planes.ravel().view(np.float32))
        d_o = pkg.DeviceBuffer(rows * E)
    d_s = pkg.DeviceBuffer(rows * E // 32 + 64)
    voidp, i = C.c_void_p, C.c_int
    handles = []
    for name, path in libs:
        L = C.CDLL(path)
        L.vh_init.argtypes = [i]
        L.vh_last_error.restype = C.c_char_p
        assert L.vh_init(0) == 0, L.vh_last_error()
        if f32:
            f0 = L.vh_launch_attention_planes
            f0.argtypes = [voidp, voidp, voidp, i, i, i, i]

            def f(st, q, o, sc, knd, n_, t_, e_, h_, f0=f0):
                return f0(st, q, o, n_, t_, e_, h_)
        elif b16:
            f1, f2 = L.vh_launch_attention_planes_f16, L.vh_launch_attention_planes_f16_mx
            f1.argtypes = [voidp, voidp, voidp, i, i, i, i, i]
            f2.argtypes = [voidp, voidp, voidp, voidp, i, i, i, i]

            def f(st, q, o, sc, knd, n_, t_, e_, h_, f1=f1, f2=f2):
                return f1(st, q, o, 1, n_, t_, e_, h_) if knd == 1 else f2(st, q, o, sc, n_, t_, e_, h_)
        else:
            f = L.vh_launch_attention_planes_f16_hd80_operand
            f.argtypes = [voidp, voidp, voidp, voidp, i, i, i, i, i]
        handles.append((name, L, f))
    outs = {}
    for name, L, f in handles:      # warm-up + the bytes each variant writes
        assert f(None, d_q.ptr, d_o.ptr, d_s.ptr if kind == 2 else None, kind, n, T, E, H) == 0, L.vh_last_error()
        assert L.vh_device_sync() == 0
        outs[name] = d_o.to_numpy()[: rows * E * 3 // 2 if f32 else rows * E // (2 if kind == 1 else 4)].copy()
    ROUNDS, REPS = 8, 20
    t = {name: [] for name, _, _ in handles}
    for r in range(ROUNDS):
        for name, L, f in (handles if r % 2 == 0 else handles[::-1]):
            L.vh_device_sync()
            t0 = time.perf_counter()
            for _ in range(REPS):
                f(None, d_q.ptr, d_o.ptr, d_s.ptr if kind == 2 else None, kind, n, T, E, H)
            L.vh_device_sync()
            t[name].append((time.perf_counter() - t0) / REPS * 1e3)
    first = handles[0][0]
    print(f"{'ViT-B/16' if b16 else 'ViT-H/14'} attention, {n} images, output kind {kind}; {ROUNDS} rounds x {REPS} launches per variant, order alternating")
    for name, _, _ in handles:
        a = np.array(t[name])
        same = "bit-identical to " + first if np.array_equal(outs[name], outs[first]) else \
            f"differs from {first} in {int((outs[name] != outs[first]).sum())} of {outs[first].size} words"
        print(f"  {name:28s} mean {a.mean():.4f} ms  min {a.min():.4f}  max {a.max():.4f}   ({same})")


if __name__ == "__main__":
    main()
