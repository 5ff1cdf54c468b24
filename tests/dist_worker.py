"""Worker for tests/test_dist.py: one rank of a world-size-N gloo group (CPU).
Exercises the host-side multi-GPU logic -- contiguous batch shards, ragged gather of
logits rows to rank 0, max-over-ranks timing -- with synthetic rows (no GPU here)."""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as graft  # noqa: E402


def fake_logits(lo, hi, classes):
    idx = torch.arange(lo, hi, dtype=torch.float32).unsqueeze(1)
    return idx * 1000.0 + torch.arange(classes, dtype=torch.float32).unsqueeze(0)


def main():
    total, classes, out_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    pkg = graft.load_package()
    from vit_with_opencl_amd.host.dist import Comm
    comm = Comm(backend="gloo")
    lo, hi = pkg.shard_range(total, comm.rank, comm.world)
    counts = [b - a for a, b in (pkg.shard_range(total, r, comm.world) for r in range(comm.world))]
    parts = comm.gather_rows(fake_logits(lo, hi, classes), counts)
    slowest = comm.max_over_ranks(1.0 + comm.rank)
    if comm.rank == 0:
        full = torch.cat(parts)
        ok = bool(torch.equal(full, fake_logits(0, total, classes)))
        Path(out_path).write_text(json.dumps({"ok": ok, "rows": full.shape[0], "slowest": slowest,
                                              "world": comm.world, "counts": counts}))
    comm.close()


if __name__ == "__main__":
    main()
