#!/usr/bin/env python3
"""A wider parity point than bench.py's 17 images: N synthetic images through the reference's own ViT_seq.c (oracle/_ref,
one process per usable core) against the library's class logits for the same images at batch 512 -- the fp32 path
(tolerance 1e-4, arg-max equal) and, for the record, the reduced modes against the SAME reference logits.
Checker use of oracle/ only (a tool, like the tests).  Usage: parity_sweep.py [N = 256]   -> text report on stdout."""
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as graft  # noqa: E402


def reference_logits(indices, procs):
    from oracle.oracle import read_records
    harness = ROOT / "oracle" / "_ref" / "ref_harness"
    assert harness.exists(), "oracle/_ref/ref_harness is not built (python -c 'import __graft_entry__ as g; g.build()' where /root/reference exists)"
    out = {}
    t0 = time.perf_counter()
    with tempfile.TemporaryDirectory() as td:
        for lo in range(0, len(indices), procs):
            chunk = indices[lo:lo + procs]
            ps = [subprocess.Popen([str(harness), "full", str(i), "1", "0", str(Path(td) / f"img{i}.bin")],
                                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for i in chunk]
            assert all(p.wait(timeout=1800) == 0 for p in ps)
            for i in chunk:
                out[i] = read_records(Path(td) / f"img{i}.bin")["logits"]
                os.unlink(Path(td) / f"img{i}.bin")
            print(f"# reference: {lo + len(chunk)} of {len(indices)} images, {time.perf_counter() - t0:.0f} s", file=sys.stderr, flush=True)
    return np.stack([out[i] for i in indices]), time.perf_counter() - t0


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    B = 512
    pkg = graft.load_package()
    cfg = pkg.preset("vit_b_16")
    weights = pkg.synth_weights(cfg, 0)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    procs = max(1, min(cores, 32))
    idx = list(range(n))
    ref, dt = reference_logits(idx, procs)
    images = pkg.synth_images(cfg, 0, B)
    print(f"ViT-B/16, synthetic weights (seed 0), synthetic images 0..{n - 1} evaluated at their positions in a batch of {B}; reference = the "
          f"reference's own ViT_seq.c built in place (oracle/_ref), {procs} processes, {dt:.0f} s")
    spread = np.linalg.norm(ref - ref.mean(1, keepdims=True), axis=1)
    for precision in ("f32", "f32_fp16x2", "bf16", "fp8"):
        m = pkg.ViTHip(cfg, weights, device=0, max_batch=B, precision=precision)
        logits, _ = m.forward(images)
        m.close()
        got = logits[:n]
        d = np.abs(got - ref)
        rel = np.linalg.norm(got - ref, axis=1) / spread
        print(f"  {precision:10s} max |dlogit| {d.max():.3e}   mean over images of max |dlogit| {d.max(1).mean():.3e}   relative L2 mean {rel.mean():.2e} "
              f"max {rel.max():.2e}   arg-max equal on {int((got.argmax(1) == ref.argmax(1)).sum())} of {n}")
        if precision == "f32":
            assert d.max() <= 1e-4 and (got.argmax(1) == ref.argmax(1)).all(), "fp32 path outside its stated tolerance"


if __name__ == "__main__":
    main()
