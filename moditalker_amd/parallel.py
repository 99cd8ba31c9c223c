"""Clip-sharded sampling across the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference samples strictly on one GPU, looping over identities and 16-frame chunks
(MToV/sample.py:267,305); chunks are independent unless --use_last_as_reference chains them
(sample.py:344-362).  So the path shards by clip with NO collective inside the denoising loop;
the only exchange is one all_gather of the finished latents ([1,4,L] fp32 = 32 KiB per clip at
(32,16)) at the end.  Each clip owns its noise stream (explicit per-clip noise), so a clip's result
does not depend on how clips are distributed over ranks.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_indices(n_items: int, rank: int, world_size: int) -> List[int]:
    """Round-robin: clip i -> rank i % world_size (chains of dependent chunks must be passed as one item)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank outside world")
    return list(range(rank, n_items, world_size))


def sample_clips_sharded(sample_fn: Callable[[int], torch.Tensor], n_clips: int,
                         group: Optional[dist.ProcessGroup] = None) -> List[torch.Tensor]:
    """Run `sample_fn(clip_index) -> [1,C,L]` for this rank's clips and all_gather the results.

    Returns the list of all n_clips results in clip order on every rank.  Works on any backend
    (`nccl` = RCCL on the GPU box, `gloo` in the CPU tests); without an initialised process group it
    degenerates to a plain loop."""
    if not (dist.is_available() and dist.is_initialized()):
        return [sample_fn(i) for i in range(n_clips)]
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    mine = shard_indices(n_clips, rank, world)
    if n_clips < world:
        raise ValueError(f"sample_clips_sharded needs n_clips >= world_size ({n_clips} < {world})")
    per_rank = (n_clips + world - 1) // world
    outs = [sample_fn(i) for i in mine]
    ref = outs[0]
    slab = torch.zeros((per_rank,) + tuple(ref.shape), dtype=ref.dtype, device=ref.device)
    for k, o in enumerate(outs):
        slab[k].copy_(o)
    gathered = [torch.empty_like(slab) for _ in range(world)]
    dist.all_gather(gathered, slab, group=group)      # the ONLY collective of the path
    result: List[Optional[torch.Tensor]] = [None] * n_clips
    for r in range(world):
        for k, i in enumerate(shard_indices(n_clips, r, world)):
            result[i] = gathered[r][k]
    return result  # type: ignore[return-value]
