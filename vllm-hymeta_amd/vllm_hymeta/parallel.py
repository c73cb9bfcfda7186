"""Tensor-parallel plumbing around the attention kernels: head sharding rules and the two
collectives of a layer, on torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in
the CPU tests).  The attention kernels themselves need no communication: every (sequence, head)
is independent, so a rank simply runs them on its own heads."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class HeadShard:
    """Heads owned by one tensor-parallel rank."""
    tp_size: int
    tp_rank: int
    num_heads: int        # q-heads on this rank
    num_kv_heads: int     # kv-heads on this rank (>= 1: replicated when tp_size > total kv-heads)
    head_start: int       # first global q-head
    kv_head_start: int    # first global kv-head


def shard_heads(total_heads: int, total_kv_heads: int, tp_size: int, tp_rank: int) -> HeadShard:
    """The reference's sharding rules (``vllm_hymeta/models/modeling_hymeta.py:401-420``): q-heads
    split evenly (``H % tp == 0``); kv-heads split evenly when there are at least ``tp`` of them,
    otherwise each kv-head is replicated on ``tp / KVH`` ranks.  With H = 28, KVH = 4 this admits
    tp in {1, 2, 4, 28}; tp = 8 is rejected, as in the reference."""
    if total_heads % tp_size != 0:
        raise ValueError(f"num_heads {total_heads} must be divisible by tensor-parallel size {tp_size}")
    if total_kv_heads >= tp_size:
        if total_kv_heads % tp_size != 0:
            raise ValueError(f"num_kv_heads {total_kv_heads} must be divisible by tensor-parallel size {tp_size}")
        kv_local = total_kv_heads // tp_size
        kv_start = tp_rank * kv_local
    else:
        if tp_size % total_kv_heads != 0:
            raise ValueError(f"tensor-parallel size {tp_size} must be a multiple of num_kv_heads {total_kv_heads}")
        kv_local = 1
        kv_start = tp_rank // (tp_size // total_kv_heads)
    h_local = total_heads // tp_size
    return HeadShard(tp_size, tp_rank, h_local, kv_local, tp_rank * h_local, kv_start)


# Rehearsal switch: run the collectives over a ONE-rank group too (a multi-GPU step exercised on one GPU, graph
# capture of the collective calls included).  Off in serving: a one-rank all-reduce is pure overhead.
SINGLE_RANK_COLLECTIVES = False


def _active(group: Optional[dist.ProcessGroup]) -> bool:
    """``group is None`` means "no tensor parallelism" and is a strict no-op, whatever process group
    the program holds for other purposes (replicas, data or pipeline parallelism): a collective only
    runs over a group that was passed explicitly (``dist.group.WORLD`` for the whole job)."""
    if group is None or not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or SINGLE_RANK_COLLECTIVES


def all_reduce_sum(x: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Sum over the tensor-parallel group, in place (the reduction behind ``RowParallelLinear``:
    ``out_proj`` / ``o_proj`` / ``down_proj``).  No-op for a single rank or without a group."""
    if _active(group):
        dist.all_reduce(x, op=dist.ReduceOp.SUM, group=group)
    return x


def all_reduce_variance(stats: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """``stats [T, 2]`` = per-rank mean of squares of the two branches over the local columns;
    returns the mean over the full hidden size (sum over ranks / tp), one message for both norms
    where the reference sends two ``[T, 1]`` ones (``modeling_hymeta.py:158-160``)."""
    if _active(group):
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
        stats /= dist.get_world_size(group)
    return stats


def tp_rank_and_size(group: Optional[dist.ProcessGroup] = None) -> Tuple[int, int]:
    """Rank and size of the tensor-parallel group; ``(0, 1)`` without one (see ``_active``)."""
    if group is not None and dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1
