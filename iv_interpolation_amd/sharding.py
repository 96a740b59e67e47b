"""Batch partitioning across the GPUs of one node: one process per GPU, contiguous shard per rank, no
data-path collective (surfaces -- like the reference's symbols -- are independent; the reference's own
parallelism is one OS process per symbol, batch_processor.py:234-239).  Ragged batches are split by
cumulative strike count so every rank streams about the same number of bytes (the idea behind the reference's
greedy complexity batching, optimized_batch_processor.py:123-164).  ``gather_outputs`` is the optional
final all-gather over RCCL/xGMI (gloo on CPU); the default is to leave results sharded."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`; sizes differ by at most one."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def ragged_shard_bounds(k_off: Sequence[int], world: int) -> List[Tuple[int, int]]:
    """Split surfaces [0, B) into `world` contiguous blocks with ~equal total strike count (k_off is CSR)."""
    k_off = np.asarray(k_off, np.int64)
    B = len(k_off) - 1
    total = int(k_off[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        cuts.append(int(np.clip(np.searchsorted(k_off, target, side="left"), cuts[-1], B)))
    cuts.append(B)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def gather_outputs(local_out, counts: Optional[Sequence[int]] = None):
    """All-gather the per-rank output blocks (first dim = surfaces) into the full batch on every rank.
    Uses torch.distributed (backend "nccl" == RCCL over xGMI on ROCm, "gloo" on CPU).  Uneven shards
    are padded to the largest block for the collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    if counts is None:
        n = torch.tensor([local_out.shape[0]], dtype=torch.int64, device=local_out.device)
        allc = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(allc, n)
        counts = [int(c) for c in allc]
    mx = max(counts)
    pad = local_out
    if local_out.shape[0] < mx:
        pad = torch.cat([local_out, local_out.new_zeros((mx - local_out.shape[0],) + tuple(local_out.shape[1:]))])
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad.contiguous())
    return torch.cat([p[:c] for p, c in zip(parts, counts)])
