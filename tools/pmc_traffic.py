#!/usr/bin/env python3
"""profiles/rNN_pmc_traffic.json from a tools/pmc_summary.py summary: the fc1 launch pair (256x256-tile launch over
rows [0, 98304) + 128x128-tile launch over the rest) of the default fp32 path, counters of the two launches added.
Usage: pmc_traffic.py <summary.txt> > profiles/r02_pmc_traffic.json
Corrections as MI355X_MICROARCH.md (HBM / rocprofv3) prescribes: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports half the bytes of wide coalesced reads."""
import hashlib
import json
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
KERNEL_SOURCE = ROOT / "vit-with-opencl_amd" / "csrc" / "gemm_p3.hip"

M, N, K, SPLIT_ROW = 100864, 3072, 768, 98304


def blocks(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"(\S.*?)\s+grid=(\d+)\s+dispatches=(\d+)\s+avg_us=([\d.]+)", line)
        if m:
            cur = out.setdefault((m.group(1), int(m.group(2))), {"avg_us": float(m.group(4))})
        elif cur is not None and line.startswith("    "):
            k, v = line.split()
            cur[k] = float(v)
    return out


def main(path):
    b = blocks(path)
    big = b[("gemm_p3_kernel<8, 256, 1, 1, 3>", (SPLIT_ROW // 256) * (N // 256) * 512)]
    tail = b[("gemm_p3_kernel<4, 128, 1, 1, 3>", ((M - SPLIT_ROW) // 128) * (N // 128) * 256)]
    fetch, write = big["FETCH_SIZE"] + tail["FETCH_SIZE"], big["WRITE_SIZE"] + tail["WRITE_SIZE"]
    alg = (M * (K + N) + N * K) * 6
    try:
        head = subprocess.run(["git", "-C", str(ROOT), "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
    except OSError:
        head = ""
    out = {
        # what the figures were measured on: bench.py withholds `traffic` when the kernel source has changed since
        "kernel_source": "vit-with-opencl_amd/csrc/gemm_p3.hip",
        "kernel_source_sha256": hashlib.sha256(KERNEL_SOURCE.read_bytes()).hexdigest(),
        "git_head": head or None,
        "kernel": "fc1 = gemm_p3_kernel<8,256,EPI_GELU,OUT_PLANES,NPL=3> on rows [0, 98304) + gemm_p3_kernel<4,128,...> "
                  "for the last 2560 rows (M=100864 N=3072 K=768); counters of the two launches added",
        "source": "rocprofv3 --kernel-trace --output-format csv, separate --pmc passes (tools/pmc_passes.sh) on `python3 "
                  "bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end`; per-dispatch means in "
                  "the round's profiles/rNN_pmc_summary.txt; this file by tools/pmc_traffic.py",
        "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
        "correction": "traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE reports half the bytes of wide "
                      "coalesced reads (MI355X_MICROARCH.md, HBM); the counters sit on the L2's memory side, so "
                      "Infinity-Cache hits are included (an upper bound on HBM bytes)",
        "traffic_bytes_per_launch": (2 * fetch + write) * 1024,
        "algorithmic_bytes_per_launch": alg,
        "write_bytes": write * 1024, "write_bytes_algorithmic": M * N * 6,
        "read_bytes": 2 * fetch * 1024, "read_bytes_algorithmic": (M * K + N * K) * 6,
        "mfma_busy_fraction": big["SQ_VALU_MFMA_BUSY_CYCLES"] / big["SQ_BUSY_CU_CYCLES"] / 4,
        "clock_ghz_in_kernel": big["GRBM_GUI_ACTIVE"] / 8 / (big["avg_us"] * 1e3),
        "valu_active_quad_cycles_per_launch": big["SQ_ACTIVE_INST_VALU"] + tail["SQ_ACTIVE_INST_VALU"],
        "wait_any_fraction_of_wave_cycles": big["SQ_WAIT_ANY"] / big["SQ_WAVE_CYCLES"],
        "lds_bank_conflict_cycles": big["SQ_LDS_BANK_CONFLICT"] + tail["SQ_LDS_BANK_CONFLICT"],
        "l2_hit_rate": big["TCC_HIT_sum"] / (big["TCC_HIT_sum"] + big["TCC_MISS_sum"]),
        "big_part_avg_us": big["avg_us"], "tail_part_avg_us": tail["avg_us"],
    }
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1])
