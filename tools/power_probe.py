#!/usr/bin/env python3
"""Board power and shader clock while one GEMM shape runs back to back.

Answers one question for DESIGN.md's GEMM section: is the split-bf16 fp32 GEMM
(51 % MFMA-busy at 1.99 GHz) held by the board's power loop, or by its own
instruction schedule?  Runs the fc1 shape (M = 197*512, K = 768, N = 3072) for a
few seconds in each arithmetic (split3 / native fp32 MFMA / bf16 operands) while a
thread samples the hwmon power and sclk files (falling back to `rocm-smi`), and
prints mean/max per leg.
"""
import ctypes as C
import glob
import json
import os
import subprocess
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as graft  # noqa: E402

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0


def sysfs_sources():
    out = []
    if os.environ.get("POWER_PROBE_HWMON") != "1":   # hwmon lists every GPU of the host, not only ours
        return out
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        p = [f for f in (hw + "/power1_average", hw + "/power1_input") if os.path.exists(f)]
        f = hw + "/freq1_input"
        if p:
            out.append((p[0], f if os.path.exists(f) else None))
    return out


def read_num(path):
    try:
        return float(open(path).read().split()[0])
    except Exception:
        return None


def smi_sample():
    try:
        txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True,
                             text=True, timeout=10).stdout
        d = json.loads(txt)
        card = d[sorted(d)[0]]
        pw = [float(v) for k, v in card.items() if "ower" in k and "(W)" in k]
        ck = [v for k, v in card.items() if k.lower().startswith("sclk")]
        mhz = float(ck[0].strip("()Mhz ")) if ck else None
        return (max(pw) if pw else None), mhz
    except Exception:
        return None, None


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.src = sysfs_sources()
        self.stop = False
        self.rows = []

    def run(self):
        while not self.stop:
            if self.src:
                pw = [read_num(p) for p, _ in self.src]
                ck = [read_num(f) for _, f in self.src if f]
                pw = [v / 1e6 for v in pw if v is not None]
                ck = [v / 1e6 for v in ck if v is not None]
                self.rows.append((max(pw) if pw else None, max(ck) if ck else None))
                time.sleep(0.05)
            else:
                self.rows.append(smi_sample())


def leg(name, launch, lib, stream):
    launch()
    lib.vh_stream_sync(stream)
    s = Sampler()
    s.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < SECONDS:
        for _ in range(20):
            launch()
        lib.vh_stream_sync(stream)
        n += 20
    dt = time.perf_counter() - t0
    s.stop = True
    s.join()
    rows = s.rows[len(s.rows) // 4:]                      # drop the ramp
    pw = [r[0] for r in rows if r[0] is not None]
    ck = [r[1] for r in rows if r[1] is not None]
    fl = 2.0 * 197 * 512 * 768 * 3072
    print(f"{name:26s} {dt / n * 1e3:7.3f} ms/launch {fl * n / dt / 1e12:7.1f} TFLOP/s | power "
          f"mean {np.mean(pw) if pw else float('nan'):7.1f} W max {max(pw) if pw else float('nan'):7.1f} W | "
          f"sclk mean {np.mean(ck) if ck else float('nan'):7.1f} MHz min {min(ck) if ck else float('nan'):7.1f} "
          f"({len(rows)} samples, source {'hwmon' if s.src else 'rocm-smi'})", flush=True)


def main():
    pkg = graft.load_package()
    lib = pkg.lib()
    pkg.binding.check(lib.vh_init(0), "vh_init")
    M, K, N = 197 * 512, 768, 3072
    rng = np.random.default_rng(0)
    a = pkg.DeviceBuffer.from_numpy(rng.standard_normal((M, K), dtype=np.float32))
    w = pkg.DeviceBuffer.from_numpy((rng.standard_normal((N, K), dtype=np.float32) * 0.02))
    b = pkg.DeviceBuffer.from_numpy(np.zeros(N, np.float32))
    out = pkg.DeviceBuffer(M * N)
    a16, w16 = pkg.DeviceBuffer(M * K // 2), pkg.DeviceBuffer(N * K // 2)
    stream = C.c_void_p()
    pkg.binding.check(lib.vh_stream_create(C.byref(stream)), "stream")
    pkg.binding.check(lib.vh_launch_split_rows(stream, a.ptr, a16.ptr, M, K, 1), "planes")     # one-part (bf16) planes
    pkg.binding.check(lib.vh_launch_split_rows(stream, w.ptr, w16.ptr, N, K, 1), "planes")
    print(f"power sources: {sysfs_sources() or 'rocm-smi'}", flush=True)

    def f32():
        pkg.binding.check(lib.vh_launch_linear(stream, out.ptr, w.ptr, a.ptr, b.ptr, M, K, N, 1, None), "linear")

    def b16():
        pkg.binding.check(lib.vh_launch_linear_planes(stream, out.ptr, 0, w16.ptr, a16.ptr, 1, b.ptr, M, K, N, 1, None),
                  "linear_planes")

    leg("idle", lambda: time.sleep(0.01), lib, stream)
    leg("fp32 (" + os.environ.get("VIT_HIP_GEMM_FP32", "split3") + ", split in the K loop)", f32, lib, stream)
    leg("bf16 operands", b16, lib, stream)


if __name__ == "__main__":
    main()
