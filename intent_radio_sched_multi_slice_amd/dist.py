"""Multi-GPU: shard independent episodes over ranks; one small gather for metrics only.

The env step has no cross-env term, so the data path needs no
collective.  Each rank owns a contiguous slice of the global batch; per reporting interval the
ranks all_gather a short float64 accumulator vector (RCCL over xGMI on GPUs, gloo on CPU).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.distributed as dist

METRIC_NAMES = ("env_steps", "reward_inter_sum", "violations", "pkts_sent", "pkts_dropped", "pkts_incoming",
                "queue_pkts", "done_envs")


def shard_range(global_batch: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of the global batch owned by ``rank`` (remainder to the low ranks)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank / world_size")
    base, rem = divmod(global_batch, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def local_metrics(reward: torch.Tensor, views: Dict[str, torch.Tensor], done: torch.Tensor,
                  n_steps: int) -> torch.Tensor:
    """Per-rank accumulator vector (float64, on the tensors' device) from the last step."""
    b = reward.shape[0]
    r0 = reward[:, 0].to(torch.float64)
    vec = torch.stack([
        torch.tensor(float(b * n_steps), dtype=torch.float64, device=reward.device),
        r0.sum(),
        (r0 < 0).sum().to(torch.float64),
        views["pkt_effective_thr"].sum(dtype=torch.float64),
        views["dropped_pkts"].sum(dtype=torch.float64),
        views["pkt_incoming"].sum(dtype=torch.float64),
        views["queue_pkts"].sum(dtype=torch.float64),
        done.sum(dtype=torch.float64),
    ])
    return vec


def gather_metrics(vec: torch.Tensor) -> torch.Tensor:
    """[world, len(vec)] on every rank; identity (1 row) without an initialised process group."""
    if not (dist.is_available() and dist.is_initialized()):
        return vec[None, :].clone()
    out: List[torch.Tensor] = [torch.empty_like(vec) for _ in range(dist.get_world_size())]
    dist.all_gather(out, vec)
    return torch.stack(out)


def summarize(gathered: torch.Tensor) -> Dict[str, float]:
    tot = gathered.sum(dim=0).tolist()
    return dict(zip(METRIC_NAMES, tot))
