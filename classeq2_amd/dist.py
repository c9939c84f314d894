"""Query sharding across the GPUs of one node.

Queries are independent (the reference's only parallelism is rayon over
queries, core/src/use_cases/place_sequences/mod.rs:123-126) and the index is
read-only, so: contiguous blocks of ceil(N/G) reads per rank (rank order ==
input order), the index replicated on every GPU, and ONE collective at the end --
a gather of the fixed 24-byte placement records to rank 0.  With the "nccl"
backend that gather is RCCL over xGMI (each peer's block lands over its own
direct link); "gloo" runs the same code on CPU tensors (tests).
"""

from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np

from . import _abi


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of ceil(n/world) reads (the last blocks may be short or empty)."""
    per = -(-n // world) if world > 0 else n
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def shard_reads(bases: np.ndarray, offsets: np.ndarray, rank: int, world: int):
    """This rank's reads as (bases, offsets rebased to 0, first read index)."""
    n = len(offsets) - 1
    lo, hi = shard_range(n, rank, world)
    b0, b1 = int(offsets[lo]), int(offsets[hi])
    return bases[b0:b1], (offsets[lo:hi + 1] - offsets[lo]).astype(np.uint64), lo


def gather_records(local: np.ndarray, n_total: int, group=None, device=None) -> Optional[np.ndarray]:
    """Gather every rank's cls_placement block to rank 0 (None elsewhere).

    `local` holds this rank's shard_range() records.  One `dist.gather` of
    ceil(n/world) * 24 bytes per rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    per = -(-n_total // world)
    buf = np.zeros(per, dtype=_abi.PLACEMENT_DTYPE)
    buf[: len(local)] = local
    t = torch.from_numpy(buf.view(np.uint8).copy())
    if device is not None:
        t = t.to(device)
    parts = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, parts, dst=0, group=group)
    if rank != 0:
        return None
    out = np.concatenate([p.cpu().numpy().view(_abi.PLACEMENT_DTYPE) for p in parts])
    # drop the padding of the short / empty trailing blocks
    keep = np.concatenate([np.arange(r * per, r * per + (shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0]))
                           for r in range(world)]).astype(np.int64)
    return out[keep]


def place_sharded(place_fn: Callable[[np.ndarray, np.ndarray], np.ndarray], bases: np.ndarray, offsets: np.ndarray,
                  group=None, device=None) -> Optional[np.ndarray]:
    """Every rank places its block with `place_fn(bases, offsets) -> records`
    (the engine's `PlacementDb.place_batch` in production); rank 0 returns all
    records in input order."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    b, o, _ = shard_reads(bases, offsets, rank, world)
    local = place_fn(b, o) if len(o) > 1 else np.zeros(0, dtype=_abi.PLACEMENT_DTYPE)
    return gather_records(local, len(offsets) - 1, group=group, device=device)
