"""Forest sharding for N > 1 GPUs: independent decoding trees share nothing, so the path
partitions by tree with NO data-path collective.  One process per GPU; each rank
owns the KV pools, metadata and plans of its trees.  The only communication is control-plane:
the timing barrier / max-over-ranks in bench.py and, if a consumer wants every rank's outputs,
one all-gather of <= 1 MB per rank (not on the hot path)."""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist


def shard_trees(kv_tokens_per_tree: Sequence[int], world_size: int) -> List[List[int]]:
    """Greedy longest-processing-time partition of trees over ranks, balanced by unique KV tokens
    (attention time is proportional to KV bytes).  Deterministic: every rank computes the same
    assignment locally, nothing is exchanged."""
    order = sorted(range(len(kv_tokens_per_tree)), key=lambda i: (-kv_tokens_per_tree[i], i))
    loads = [0] * world_size
    shards: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += kv_tokens_per_tree[i]
    return [sorted(s) for s in shards]


def cfg5_shard(world_size: int, rank: int, trees_per_gpu: int = 8, kv_tokens_per_tree: int = 8704) -> List[int]:
    """BASELINE configs[4]: the batch has `trees_per_gpu` x world_size equal trees (64 on 8 GPUs); this rank's share."""
    return shard_trees([kv_tokens_per_tree] * (trees_per_gpu * world_size), world_size)[rank]


def max_over_ranks(seconds: float, device: torch.device) -> float:
    """Slowest rank's time (bench contract: barrier, time, MAX over ranks)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_gather_outputs(o_local: torch.Tensor) -> List[torch.Tensor]:
    """Optional, off the hot path: every rank receives every rank's attention outputs
    (RCCL all-gather on GPUs, gloo on CPU).  Shapes may differ per rank (ragged forests)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [o_local]
    world, rank = dist.get_world_size(), dist.get_rank()
    shapes = [None] * world
    dist.all_gather_object(shapes, tuple(o_local.shape))
    outs = [torch.empty(s, dtype=o_local.dtype, device=o_local.device) for s in shapes]
    if len(set(shapes)) == 1:
        dist.all_gather(outs, o_local.contiguous())
        return outs
    # ragged (forests of different sizes per rank): ONE all-gather of row-padded buffers -- a single ring collective over xGMI
    # instead of a broadcast per rank; a rank with no rows takes part with padding only.  (Rows differ, the trailing shape
    # -- heads x head_dim -- is the model's; tensors whose trailing shapes differ fall back to one broadcast per rank.)
    if len({tuple(sh[1:]) for sh in shapes}) == 1 and all(len(sh) >= 1 for sh in shapes):
        most = max(sh[0] for sh in shapes)
        padded = torch.zeros((most,) + tuple(shapes[rank][1:]), dtype=o_local.dtype, device=o_local.device)
        padded[: o_local.shape[0]].copy_(o_local)
        bufs = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(bufs, padded)
        return [bufs[r][: shapes[r][0]] for r in range(world)]
    outs[rank].copy_(o_local)
    for src in range(world):
        dist.broadcast(outs[src], src=src)
    return outs
