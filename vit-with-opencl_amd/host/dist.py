"""One-process-per-GPU plumbing for the batch-sharded forward pass.

The path shards by image (ViT_opencl.c:926 processes images strictly one at a time;
nothing crosses images), so there is no data-path collective.  The only exchange is
the gather of each rank's [n_local][num_classes] logits to rank 0, over RCCL
(`backend="nccl"` is RCCL on ROCm) -- or gloo for CPU rehearsal and tests.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


class Comm:
    def __init__(self, backend: str | None = None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.backend = backend or os.environ.get("VIT_DIST_BACKEND", "nccl")
        # VIT_DIST_FORCE=1 initialises the process group even for one rank (single-GPU rehearsal of the RCCL path)
        self.active = self.world > 1 or os.environ.get("VIT_DIST_FORCE", "0") == "1"
        if self.active:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            kw = {}
            if self.backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                kw["device_id"] = torch.device("cuda", self.local_rank)
            dist.init_process_group(backend=self.backend, rank=self.rank, world_size=self.world, **kw)

    @property
    def device(self) -> torch.device:
        return torch.device("cuda", self.local_rank) if self.backend == "nccl" else torch.device("cpu")

    def barrier(self) -> None:
        if self.active:
            dist.barrier()

    def gather_rows(self, local: torch.Tensor, counts: list[int] | None = None):
        """Gather per-rank row blocks to rank 0 -> list of tensors in rank order (None elsewhere).
        `counts[r]` = rows of rank r (defaults to equal shards)."""
        if not self.active:
            return [local]
        counts = counts or [local.shape[0]] * self.world
        if len(set(counts)) == 1:
            out = [torch.empty_like(local) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(local, out, dst=0)
            return out
        # ragged shards: pad to the largest, trim on rank 0
        width, big = local.shape[1], max(counts)
        padded = torch.zeros(big, width, dtype=local.dtype, device=local.device)
        padded[: local.shape[0]] = local
        out = [torch.empty_like(padded) for _ in range(self.world)] if self.rank == 0 else None
        dist.gather(padded, out, dst=0)
        return [t[:c] for t, c in zip(out, counts)] if self.rank == 0 else None

    def max_over_ranks(self, value: float) -> float:
        if not self.active:
            return value
        t = torch.tensor([value], dtype=torch.float64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def rank0_says_done(self, timeout_s: int = 1200) -> None:
        """Host-side hand-over through the rendezvous store: rank 0 calls it when its rank-0-only work (CPU baseline,
        checks) is finished, every other rank blocks in it until then -- on the CPU, with no collective kernel
        spinning on its GPU meanwhile (an RCCL barrier would)."""
        if not self.active or self.world == 1:
            return
        from datetime import timedelta
        store = dist.distributed_c10d._get_default_store()
        if self.rank == 0:
            store.set("vit_rank0_done", "1")
        else:
            store.wait(["vit_rank0_done"], timedelta(seconds=timeout_s))

    def close(self) -> None:
        if self.active:
            dist.barrier()
            dist.destroy_process_group()
