#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE ITSELF.

Runs oracle/_ref/ref_harness -- the reference's unmodified ViT_seq.c compiled in
place by oracle/Makefile -- on the deterministic synthetic inputs of
vit_synth_* (weights: seed_base 0; images: index i -> seed 1000+i) and stores
its outputs as small fixtures.  Only possible in the build container (needs
/root/reference); the fixtures are committed, the reference never travels.

Why synthetic inputs: 36 of the reference's 152 weight files and
Data/input-100.bin are absent from the snapshot (/root/reference/.MISSING_LARGE_BLOBS),
so its pretrained goldens (Data/answer_result*.txt) cannot be reproduced.

Fixtures (inputs are regenerated from seeds, never stored):
  b16_full.npz    logits[4][1000], probs[4][1000] for synthetic images 0..3
  b16_stages.npz  per-stage outputs for image 0 / layer 0: small tensors in full,
                  large ones as a stride-37 sample plus fp64 sum and abs-sum
  b16_answer_result.txt  Main.c-format result lines ("[i] label: L / prob: P")
  b16_full_rounded.npz / b16_answer_result_rounded.txt  the same with weights rounded to 1e-6
                  as the reference loader delivers them from disk (Network.c:208-211)
  b16_answer_result_100_rounded.txt  (`make_golden.py answers100`) 100 result lines -- comparator.c's
                  IMAGE_COUNT (comparator.c:9) -- for synthetic images 0..99 on rounded weights, written
                  with Main.c's own arg-max loop (Main.c:59-71: pred_idx is NOT reset per image and class 0
                  is never visited, so the line is what the reference's Main.c would print around ViT_seq)
  b16_seed1.npz   (`make_golden.py other_seed`) logits / probabilities of synthetic images 100, 101 on a SECOND weight set
                  (seed_base 1), from the reference's ViT_seq.c
  b16_real_image.npz  (`make_golden.py real_image`) the reference's ONE real input -- Data/input-1.bin, a normalised
                  224x224 photograph (spatially correlated pixels, unlike every synthetic image) -- kept as a data fixture,
                  with the logits / probabilities the reference's ViT_seq.c gives for it on the seed-0 synthetic weights
                  (its pretrained answer Data/answer_result_1.txt cannot be reproduced: 36 weight files are absent)
"""
from __future__ import annotations

import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import oracle as orc  # noqa: E402

GOLD = ROOT / "tests" / "golden"
STRIDE = 37
N_IMAGES = 4


def summarize(a: np.ndarray) -> dict[str, np.ndarray]:
    a = a.ravel()
    return {
        "sample": a[::STRIDE].copy(),
        "sum": np.array(a.astype(np.float64).sum()),
        "abssum": np.array(np.abs(a.astype(np.float64)).sum()),
        "count": np.array(a.size),
    }


def main_c_lines(probs: np.ndarray) -> list[str]:
    """The result lines exactly as Main.c:59-72 produces them from `probabilities`."""
    lines, pred = [], 0
    for i in range(probs.shape[0]):
        for j in range(1, probs.shape[1]):          # Main.c:62: j starts at 1; pred carries over (Main.c:59)
            if probs[i][j] > probs[i][pred]:
                pred = j
        lines.append("[%d] label: %d / prob: %.6f\n" % (i, pred, probs[i][pred]))
    return lines


def answers100(workers: int = 8) -> None:
    """100 images through the reference's ViT_seq.c on 1e-6-rounded weights, `workers` processes at a time."""
    import subprocess
    orc.build()
    if not orc.have_reference():
        sys.exit("oracle/_ref/ref_harness missing: run in the build container (needs /root/reference)")
    n = 100
    probs = np.empty((n, 1000), dtype=np.float32)
    with tempfile.TemporaryDirectory() as td:
        for lo in range(0, n, workers):
            idx = list(range(lo, min(n, lo + workers)))
            procs = [subprocess.Popen([str(orc.REF_HARNESS), "full_rounded", str(i), "1", "0", str(Path(td) / f"{i}.bin")],
                                      stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for i in idx]
            assert all(p.wait() == 0 for p in procs)
            for i in idx:
                probs[i] = orc.read_records(Path(td) / f"{i}.bin")["probs"]
            print("images", idx[0], "..", idx[-1], "done", flush=True)
    (GOLD / "b16_answer_result_100_rounded.txt").write_text("".join(main_c_lines(probs)))
    # cross-check against the 4-image rounded fixture
    small = np.load(GOLD / "b16_full_rounded.npz")["probs"]
    assert np.array_equal(small, probs[:4]), "100-image run disagrees with b16_full_rounded.npz"
    print("wrote b16_answer_result_100_rounded.txt; labels:", sorted(set(int(l.split()[2]) for l in main_c_lines(probs))))


def real_image() -> None:
    """Data/input-1.bin through the reference's ViT_seq.c on the seed-0 synthetic weights -> b16_real_image.npz."""
    src = Path("/root/reference/MulticoreMainProject/Data/input-1.bin")
    orc.build()
    if not orc.have_reference() or not src.exists():
        sys.exit("needs the build container (oracle/_ref/ref_harness and the reference's Data/input-1.bin)")
    raw = src.read_bytes()
    n, c, h, w = np.frombuffer(raw, dtype=np.int32, count=4)
    assert (n, c, h, w) == (1, 3, 224, 224) and len(raw) == 16 + 4 * c * h * w
    image = np.frombuffer(raw, dtype=np.float32, offset=16).reshape(c, h, w).copy()
    with tempfile.TemporaryDirectory() as td:
        rec = orc.run_reference("full_file", src, 0, out_path=Path(td) / "real.bin")
    np.savez_compressed(GOLD / "b16_real_image.npz", image=image, logits=rec["logits"].reshape(1, 1000),
                        probs=rec["probs"].reshape(1, 1000), seed_base=np.array(0),
                        note=np.array("image = the reference's Data/input-1.bin (data, its only real input); logits/probs = "
                                      "the reference's own ViT_seq.c on it with vit_synth_tensor weights, seed_base 0"))
    lg = rec["logits"]
    print("real image: mean %.3f std %.3f  argmax %d  top-2 margin %.4f" % (image.mean(), image.std(), int(lg.argmax()),
                                                                           float(np.sort(lg)[-1] - np.sort(lg)[-2])))


def other_seed() -> None:
    """A second weight set (seed_base 1) and other images (100, 101) through the reference's ViT_seq.c -> b16_seed1.npz:
    the pinning does not rest on one draw of the synthetic weights."""
    orc.build()
    if not orc.have_reference():
        sys.exit("needs the build container (oracle/_ref/ref_harness)")
    with tempfile.TemporaryDirectory() as td:
        rec = orc.run_reference("full", 100, 2, 1, out_path=Path(td) / "seed1.bin")
    np.savez(GOLD / "b16_seed1.npz", logits=rec["logits"].reshape(2, 1000), probs=rec["probs"].reshape(2, 1000),
             seed_base=np.array(1), first_image=np.array(100))
    print("seed 1: argmax", rec["logits"].reshape(2, 1000).argmax(1))


def main() -> None:
    if len(sys.argv) > 1 and sys.argv[1] == "other_seed":
        other_seed()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "answers100":
        answers100()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "real_image":
        real_image()
        return
    orc.build()
    if not orc.have_reference():
        sys.exit("oracle/_ref/ref_harness missing: run in the build container (needs /root/reference)")
    GOLD.mkdir(parents=True, exist_ok=True)
    with tempfile.TemporaryDirectory() as td:
        full = orc.run_reference("full", 0, N_IMAGES, 0, out_path=Path(td) / "full.bin")
        stages = orc.run_reference("stages", 0, out_path=Path(td) / "stages.bin")
        rounded = orc.run_reference("full_rounded", 0, N_IMAGES, 0, out_path=Path(td) / "rounded.bin")

    logits = full["logits"].reshape(N_IMAGES, 1000)
    probs = full["probs"].reshape(N_IMAGES, 1000)
    np.savez(GOLD / "b16_full.npz", logits=logits, probs=probs,
             seed_base=np.array(0), first_image=np.array(0),
             seconds_per_image=full["seconds_per_image"])

    out = {}
    for name in ("conv", "tokens", "ln", "mha", "mlp", "enc0"):
        for k, v in summarize(stages[name]).items():
            out[f"{name}_{k}"] = v
    for name in ("head", "softmax", "gelu"):
        out[name] = stages[name]
    out["stride"] = np.array(STRIDE)
    np.savez(GOLD / "b16_stages.npz", **out)

    # The file-based flow (Main.c: load_weights rounds every value to 1e-6, Network.c:208-211).
    rl = rounded["logits"].reshape(N_IMAGES, 1000)
    rp = rounded["probs"].reshape(N_IMAGES, 1000)
    np.savez(GOLD / "b16_full_rounded.npz", logits=rl, probs=rp)
    with open(GOLD / "b16_answer_result_rounded.txt", "w") as f:
        for i in range(N_IMAGES):
            k = int(np.argmax(rp[i]))
            f.write("[%d] label: %d / prob: %.6f\n" % (i, k, rp[i][k]))

    # Result lines exactly as Main.c:59-72 prints them, but with a per-image argmax
    # (Main.c:59 never resets pred_idx).
    with open(GOLD / "b16_answer_result.txt", "w") as f:
        for i in range(N_IMAGES):
            k = int(np.argmax(probs[i]))
            f.write("[%d] label: %d / prob: %.6f\n" % (i, k, probs[i][k]))
    print("wrote", sorted(p.name for p in GOLD.iterdir()))
    print("argmax", logits.argmax(1), "top-2 margin", np.sort(logits, 1)[:, -1] - np.sort(logits, 1)[:, -2])


if __name__ == "__main__":
    main()
