#!/usr/bin/env python3
"""Time the planes GEMM entry points (three-part fp32 split, one-part bf16, block-scaled fp8) on the four
projection shapes of ViT-B/16 at batch 512, interleaved in one process on random operands.
Usage (GPU box): python tools/gemm_rates.py [rounds]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as graft  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    pkg = graft.load_package()
    L = pkg.lib()
    pkg.binding.check(L.vh_init(0), "vh_init")
    M = 197 * 512
    rng = np.random.default_rng(0)
    shapes = [("qkv", 768, 2304, 0, False), ("out_proj", 768, 768, 0, True), ("fc1", 768, 3072, 1, False), ("fc2", 3072, 768, 0, True)]
    ev0, ev1 = C.c_void_p(), C.c_void_p()
    L.vh_event_create(C.byref(ev0)); L.vh_event_create(C.byref(ev1))
    for name, K, N, gelu, resid in shapes:
        x = pkg.DeviceBuffer.from_numpy(rng.standard_normal((M, K), dtype=np.float32))
        w = pkg.DeviceBuffer.from_numpy(rng.standard_normal((N, K), dtype=np.float32) * np.float32(0.03))
        b = pkg.DeviceBuffer.from_numpy(np.zeros(N, np.float32))
        out = pkg.DeviceBuffer(M * N * 3 // 2 + 16)
        res = pkg.DeviceBuffer(M * N) if resid else None
        x3, w3 = pkg.DeviceBuffer(M * K * 3 // 2 + 16), pkg.DeviceBuffer(N * K * 3 // 2 + 16)
        x1, w1 = pkg.DeviceBuffer(M * K // 2 + 16), pkg.DeviceBuffer(N * K // 2 + 16)
        # activations carry their MX scale bytes in the activation order (vh_mx_act_scale_bytes), weights in theirs
        xv, xs, wv, ws = (pkg.DeviceBuffer(M * K // 4 + 16), pkg.DeviceBuffer(L.vh_mx_act_scale_bytes(M, K) // 4 + 16),
                          pkg.DeviceBuffer(N * K // 4 + 16), pkg.DeviceBuffer(N * K // 128 + 16))
        os_ = pkg.DeviceBuffer(L.vh_mx_act_scale_bytes(M, N) // 4 + 16)
        chk = pkg.binding.check
        chk(L.vh_launch_split_rows(None, x.ptr, x3.ptr, M, K, 3)); chk(L.vh_launch_split_rows(None, w.ptr, w3.ptr, N, K, 3))
        chk(L.vh_launch_split_rows(None, x.ptr, x1.ptr, M, K, 1)); chk(L.vh_launch_split_rows(None, w.ptr, w1.ptr, N, K, 1))
        chk(L.vh_launch_quantize_mx_act(None, x.ptr, xv.ptr, xs.ptr, M, K)); chk(L.vh_launch_quantize_mx_rows(None, w.ptr, wv.ptr, ws.ptr, N, K))
        planes_out = 1 if gelu else 0          # fc1 writes its consumer's format, the others fp32 rows
        r = res.ptr if resid else None
        legs = {
            "fp32 (3 parts)": lambda: L.vh_launch_linear_planes(None, out.ptr, planes_out, w3.ptr, x3.ptr, 3, b.ptr, M, K, N, gelu, r),
            "bf16 (1 part) ": lambda: L.vh_launch_linear_planes(None, out.ptr, planes_out, w1.ptr, x1.ptr, 1, b.ptr, M, K, N, gelu, r),
            "MX fp8        ": lambda: L.vh_launch_linear_mx(None, out.ptr, os_.ptr if planes_out else None, wv.ptr, ws.ptr, xv.ptr, xs.ptr, b.ptr, M, K, N, gelu, r),
        }
        best = {k: 1e9 for k in legs}
        for k, f in legs.items():
            chk(f(), k)
        chk(L.vh_device_sync())
        for _ in range(rounds):
            for k, f in legs.items():
                L.vh_event_record(ev0, None)
                for _ in range(10):
                    f()
                L.vh_event_record(ev1, None)
                L.vh_event_sync(ev1)
                ms = C.c_float()
                L.vh_event_elapsed_ms(C.byref(ms), ev0, ev1)
                best[k] = min(best[k], ms.value / 10)
        fl = 2.0 * M * N * K
        print(f"{name:9s} M={M} K={K} N={N}: " + "  ".join(f"{k.strip()} {v:.3f} ms ({fl / v / 1e9:.0f} TFLOP/s)" for k, v in best.items()), flush=True)
        for d in (x, w, b, out, x3, w3, x1, w1, xv, xs, wv, ws, os_):
            d.free()
        if res:
            res.free()


if __name__ == "__main__":
    main()
