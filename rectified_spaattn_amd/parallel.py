"""Head sharding across the GPUs of one node (one process per GPU, torch.distributed: "nccl" == RCCL over xGMI
on ROCm, "gloo" in the CPU tests).

The algorithm has no cross-head dependency (softmax / sort / cumsum are per (b, h, q-block) row; top_k, p and
the neighbour matrix are head-independent), so mask selection and the sparse pass need no collective.  The
only exchange is optional and sits at the layer boundary: an all-gather of O along the head axis, needed iff
the consumer (the to_out GEMM) is not head-sharded itself."""
from typing import List, Tuple

import torch
import torch.distributed as dist


def head_shard(num_heads: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous head blocks: returns (first_head, heads_on_this_rank).  Requires world_size | num_heads."""
    if num_heads % world_size:
        raise ValueError(f"{num_heads} heads do not split evenly over {world_size} ranks")
    per = num_heads // world_size
    return rank * per, per


def gather_heads(out_local: torch.Tensor, group=None) -> torch.Tensor:
    """out_local [B, S, H_local, D] (or [B, S, H_local*D]) on every rank -> [B, S, H*D] on every rank, heads
    in rank order (== the unsharded layout of the reference's output, hunyuan :383-387)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    B, S = out_local.shape[:2]
    flat = out_local.reshape(B, S, -1).contiguous()
    if world == 1:
        return flat
    parts: List[torch.Tensor] = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(parts, flat, group=group)
    return torch.cat(parts, dim=-1)


def reduce_step_stats(elapsed_s: float, flops: float, pairs: float, k5_ms: float, device, group=None,
                      busy_s: float = None):
    """(max elapsed, sum flops, sum kept pairs, max K5 ms, per-rank list) over all ranks.  The per-rank list holds
    each rank's own busy time `busy_s` (measured before the closing barrier; default: elapsed_s)."""
    busy_s = elapsed_s if busy_s is None else busy_s
    t = torch.tensor([busy_s, flops, pairs, k5_ms, elapsed_s], dtype=torch.float64, device=device)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return elapsed_s, flops, pairs, k5_ms, [busy_s]
    tmax, tsum = t.clone(), t.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(tsum, op=dist.ReduceOp.SUM, group=group)
    per_rank = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(per_rank, t, group=group)
    return tmax[4].item(), tsum[1].item(), tsum[2].item(), tmax[3].item(), [p[0].item() for p in per_rank]
