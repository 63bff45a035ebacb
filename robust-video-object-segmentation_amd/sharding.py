"""Sequence sharding for multi-GPU evaluation.

Video sequences are independent units (the reference drops all per-sequence state between
sequences, eval_manager_mm.py:376-382), so the hot path shards by sequence: one process per GPU,
no data-path collective.  The only message is one all-reduce(SUM) of a small float64 vector of
metric accumulators at the end (RCCL over xGMI when the backend is "nccl"; "gloo" on CPU tests).
"""
from typing import List, Sequence

import torch
import torch.distributed as dist

METRIC_FIELDS = ("frames", "objects", "gpu_seconds", "sum_iou", "iou_count", "sum_f")


def lpt_partition(costs: Sequence[float], n_ranks: int) -> List[List[int]]:
    """Longest-processing-time-first: sort sequences by cost (frames x objects) descending and give
    each to the least loaded rank.  Deterministic (ties -> lower index / lower rank)."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * n_ranks
    parts: List[List[int]] = [[] for _ in range(n_ranks)]
    for i in order:
        r = min(range(n_ranks), key=lambda j: (loads[j], j))
        parts[r].append(i)
        loads[r] += costs[i]
    return parts


def _reduce_device(device):
    """RCCL reduces device tensors; the gloo backend (CPU tests, several ranks on one GPU) reduces on the host."""
    if dist.is_available() and dist.is_initialized() and dist.get_backend() != "nccl":
        return None
    return device


def allreduce_metrics(local: dict, device=None) -> dict:
    """Sum the metric accumulators over all ranks (no-op without an initialised process group; with one, the collective runs even for a
    single rank -- that is how a one-GPU box exercises RCCL)."""
    vec = torch.tensor([float(local.get(k, 0.0)) for k in METRIC_FIELDS], dtype=torch.float64, device=_reduce_device(device))
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    return dict(zip(METRIC_FIELDS, vec.tolist()))


def allreduce_max(value: float, device=None) -> float:
    """Maximum of a scalar over the ranks (the slowest rank's time: load imbalance)."""
    t = torch.tensor([float(value)], dtype=torch.float64, device=_reduce_device(device))
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def mask_iou_sums(pred: torch.Tensor, ref: torch.Tensor, n_obj: int):
    """Sum over objects of IoU(pred == o, ref == o) (utils/metric.py:3-34 formula, empty-vs-empty = 1)
    and the number of objects counted."""
    total, count = 0.0, 0
    for o in range(n_obj):
        p, r = pred == o, ref == o
        union = int((p | r).sum())
        total += 1.0 if union == 0 else int((p & r).sum()) / union
        count += 1
    return total, count
