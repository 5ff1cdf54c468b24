#!/usr/bin/env python3
"""Probe: does running two half-batches on two streams of one GPU, out of phase, beat one batch of 512?  (Kernels that sit
under the power cap -- attention, LayerNorm -- overlapping with the capped GEMMs of the other half.)"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as graft
pkg = graft.load_package()
L = pkg.lib()
cfg = pkg.preset("vit_b_16")
w = pkg.synth_weights(cfg, 0)
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
per = cfg.in_chans * cfg.img_size * cfg.img_size
def make(B):
    m = pkg.ViTHip(cfg, w, device=0, max_batch=B, precision=prec)
    d_img = pkg.DeviceBuffer(B * per)
    img = pkg.synth_images(cfg, 0, min(B, 64))
    for lo in range(0, B, 64):
        n = min(64, B - lo)
        pkg.binding.check(L.vh_h2d(d_img.ptr.value + lo * per * 4, img.ctypes.data, n * per * 4, None), "h2d")
    pkg.binding.check(L.vh_device_sync(), "sync")
    return m, d_img, pkg.DeviceBuffer(B * cfg.num_classes), pkg.DeviceBuffer(B * cfg.num_classes)
def run(ctxs, steps, offset_first=False):
    for m, di, dl, dp, B in ctxs:      # warm-up
        m.forward_device(di.ptr, B, dl.ptr, dp.ptr, m.stream)
    pkg.binding.check(L.vh_device_sync(), "sync")
    if offset_first and len(ctxs) > 1:  # put the second stream a fraction of a step behind
        m, di, dl, dp, B = ctxs[1]
        m.forward_device(di.ptr, max(B // 8, 1), dl.ptr, dp.ptr, m.stream)
    t0 = time.perf_counter()
    for _ in range(steps):
        for m, di, dl, dp, B in ctxs:
            m.forward_device(di.ptr, B, dl.ptr, dp.ptr, m.stream)
    pkg.binding.check(L.vh_device_sync(), "sync")
    dt = time.perf_counter() - t0
    return sum(c[4] for c in ctxs) * steps / dt
one = make(512)
two = [make(256), make(256)]
four = [make(128) for _ in range(4)]
res = {}
for rnd in range(3):
    res.setdefault("1 x 512", []).append(run([(*one, 512)], 10))
    res.setdefault("2 x 256, two streams", []).append(run([(*c, 256) for c in two], 10))
    res.setdefault("2 x 256, second stream offset", []).append(run([(*c, 256) for c in two], 10, True))
    res.setdefault("4 x 128, four streams", []).append(run([(*c, 128) for c in four], 10))
    res.setdefault("1 x 256 alone", []).append(run([(*two[0], 256)], 10))
print(f"ViT-B/16 {prec}, images/s, three rounds each:")
for k, v in res.items():
    print(f"  {k:32s} " + "  ".join(f"{x:8.1f}" for x in v))
