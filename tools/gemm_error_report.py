#!/usr/bin/env python3
"""Error of each fp32-product arithmetic of the GEMM against a float64 reference, measured on the
GPU's own outputs (not a numpy model of them): the exact three-part bf16 split (default), the
fp16-pair emulation (opt-in), the native fp32 MFMA, and -- for scale -- the reference's own
sequential fp32 loop (the oracle port).  Errors are relative to the RMS of the exact results."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

CHILD = os.environ.get("GEMM_ERR_CHILD")
import __graft_entry__ as graft  # noqa: E402

SHAPES = [(512, 768, 2304), (512, 768, 3072), (512, 3072, 768)]


def run_child(which):
    pkg = graft.load_package()
    L = pkg.lib()
    pkg.binding.check(L.vh_init(0), "vh_init")
    rng = np.random.default_rng(11)
    out = {}
    for M, K, N in SHAPES:
        x = rng.standard_normal((M, K)).astype(np.float32)
        w = (rng.standard_normal((N, K)) * 0.03).astype(np.float32)
        b = (rng.standard_normal(N) * 0.1).astype(np.float32)
        d_x, d_w, d_b = (pkg.DeviceBuffer.from_numpy(a) for a in (x, w, b))
        d_o = pkg.DeviceBuffer(M * N)
        if which == "fp16x2":
            scale = 2.0 ** (14 - int(np.frexp(np.abs(w).max())[1]))
            d_p = pkg.DeviceBuffer(N * K)
            pkg.binding.check(L.vh_launch_split2h_planes(None, d_w.ptr, d_p.ptr, N, K, scale), "split")
            pkg.binding.check(L.vh_launch_linear_h2(None, d_o.ptr, d_p.ptr, scale, d_x.ptr, d_b.ptr, M, K, N, 0, None), "h2")
        else:                      # "split3" or "native" (VIT_HIP_GEMM_FP32 is read once per process)
            pkg.binding.check(L.vh_launch_linear(None, d_o.ptr, d_w.ptr, d_x.ptr, d_b.ptr, M, K, N, 0, None), "linear")
        pkg.binding.check(L.vh_device_sync(), "sync")
        out[f"{M}x{K}x{N}"] = d_o.to_numpy((M, N))
    np.savez(os.environ["GEMM_ERR_OUT"], **out)


def main():
    if CHILD:
        return run_child(CHILD)
    from oracle.oracle import Oracle
    orc = Oracle("vit_b_16")
    rng = np.random.default_rng(11)
    res = {}
    for which, env in (("split3", {}), ("fp16x2", {}), ("native", {"VIT_HIP_GEMM_FP32": "native"})):
        path = f"/tmp/gemm_err_{which}.npz"
        subprocess.run([sys.executable, __file__], check=True,
                       env=dict(os.environ, GEMM_ERR_CHILD=which, GEMM_ERR_OUT=path, **env))
        res[which] = np.load(path)
    print("relative error against float64 (max | rms), per shape M x K x N")
    for M, K, N in SHAPES:
        x = rng.standard_normal((M, K)).astype(np.float32)
        w = (rng.standard_normal((N, K)) * 0.03).astype(np.float32)
        b = (rng.standard_normal(N) * 0.1).astype(np.float32)
        exact = x.astype(np.float64) @ w.astype(np.float64).T + b
        scale = np.sqrt((exact ** 2).mean())
        seq = orc.linear(x[:64], w, b, N)
        line = f"  {M}x{K}x{N}:"
        for which, label in (("split3", "3 x bf16, 6 products (default)"), ("fp16x2", "2 x fp16, 3 products"),
                             ("native", "fp32 MFMA")):
            e = (res[which][f"{M}x{K}x{N}"] - exact) / scale
            line += f"  {label}: {np.abs(e).max():.2e} | {np.sqrt((e ** 2).mean()):.2e};"
        e = (seq - exact[:64]) / scale
        line += f"  reference's sequential fp32 loop (64 rows): {np.abs(e).max():.2e} | {np.sqrt((e ** 2).mean()):.2e}"
        print(line)


if __name__ == "__main__":
    main()
