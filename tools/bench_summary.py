#!/usr/bin/env python3
"""One line per bench.py JSON file: throughput and the per-operator average times."""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        print(path, "unreadable:", e)
        continue
    k = d.get("kernels", {})
    emu = d.get("fp32_fp16x2_emulation_mode", {}).get("value")
    ops = " ".join(f"{n.replace('_gemm', '')}={v['avg_ms']:.3f}" for n, v in k.items()
                   if n not in ("softmax", "head_gemm"))
    b16 = d.get("bf16_gemm_mode", {}).get("value")
    par = d.get("parity", {})
    print(f"{path.split('/')[-1]:24s} {d['value']:8.1f} img/s {d['ms_per_step']:7.2f} ms | {ops} | bf16 {b16} | fp16x2-emu {emu} | "
          f"dlogit {par.get('max_abs_dlogit') if isinstance(par, dict) else par}")
