#!/usr/bin/env python3
"""Generate tests/golden/{h14,l16}_port_logits.npz from the CPU PORT (oracle/vit_seq_port.c).

"port, parity UNPINNED": the reference hard-codes ViT-B/16 (ViT_seq.c:10-21) and has no code for
ViT-L/16 or ViT-H/14, so these vectors are NOT outputs of the reference itself -- they are the port
(bit-identical to the reference's ViT_seq.c on ViT-B/16, tests/test_oracle.py) run with other loop
bounds.  They exist so that the GPU tests of BASELINE configs 4 and 5 can check a FULL-DEPTH image
(24 / 32 layers) without spending 1-2 minutes of CPU per image on the test box.

Inputs are regenerated from seeds, never stored:
    h14_port_logits.npz   ViT-H/14, weights seed_base 3, synthetic images 5 and 6
    l16_port_logits.npz   ViT-L/16, weights seed_base 7, synthetic images 3 and 4
each with logits[2][1000], probs[2][1000], images[2] (the indices) and seed_base.

    python oracle/make_golden_port.py            # both files, one process per image
"""
from __future__ import annotations

import sys
from multiprocessing import Pool
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle.oracle import Oracle  # noqa: E402

GOLD = ROOT / "tests" / "golden"
CASES = {"h14": ("vit_h_14", 3, [5, 6]), "l16": ("vit_l_16", 7, [3, 4])}


def one_image(job):
    preset, seed_base, index = job
    orc = Oracle(preset)
    logits, probs, _ = orc.forward(orc.synth_image(index), orc.synth_weights(seed_base))
    return logits, probs


def main() -> None:
    jobs = [(preset, seed, i) for preset, seed, idx in CASES.values() for i in idx]
    with Pool(len(jobs)) as pool:
        res = pool.map(one_image, jobs)
    k = 0
    for tag, (preset, seed, idx) in CASES.items():
        logits = np.stack([res[k + j][0] for j in range(len(idx))])
        probs = np.stack([res[k + j][1] for j in range(len(idx))])
        k += len(idx)
        np.savez_compressed(GOLD / f"{tag}_port_logits.npz", logits=logits, probs=probs,
                            images=np.array(idx), seed_base=np.array(seed),
                            note=np.array(f"{preset}: oracle/vit_seq_port.c (the port; parity unpinned -- the reference has "
                                          f"no {preset} code), oracle/make_golden_port.py"))
        print(tag, "argmax", logits.argmax(1), "prob", probs.max(1))


if __name__ == "__main__":
    main()
