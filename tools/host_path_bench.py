#!/usr/bin/env python3
"""PCIe-inclusive throughput of the host-pointer entry (vit_hip_forward): n separately
allocated host images in, host logits + probabilities out, weights resident.  This is the
number DESIGN.md quotes beside (never instead of) bench.py's device-resident `value`."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as graft  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 512
pkg = graft.load_package()
cfg = pkg.preset("vit_b_16")
model = pkg.ViTHip(cfg, pkg.synth_weights(cfg, 0), device=0, max_batch=chunk)
base = pkg.synth_images(cfg, 0, 64)
images = np.concatenate([base] * (n // 64))
model.forward(images[:chunk])                      # warm-up
t0 = time.perf_counter()
logits, probs = model.forward(images)
dt = time.perf_counter() - t0
assert np.array_equal(logits[:64], logits[64:128]) and np.isfinite(logits).all()
print(f"host path: {n} images in {dt * 1e3:.1f} ms = {n / dt:.1f} images/sec (chunk {chunk}, "
      f"{n * 602112 / dt / 1e9:.2f} GB/s of image bytes over PCIe)")
model.close()
