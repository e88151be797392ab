"""Multi-GPU glue: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm, "gloo" in CPU tests).

The path shards naturally: rollouts are independent per env (no data-path collective); training is pure
data parallelism with ONE gradient all-reduce per optimizer step plus a few scalars:
  * `n` (the masked token count the losses divide by) is all-reduced BEFORE the loss kernel, so every rank's local
    gradient is already its share of the global-mean loss and the gradient reduction is a plain SUM;
  * logged sums / mins / maxes are reduced so the log dict equals the single-process one;
  * PPO's whole-batch advantage whitening needs the three moments (sum, sum of squares, count) all-reduced.
Gradients are packed into large flat buckets: xGMI links are point-to-point, so few large messages beat many small ones.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence, Tuple


def is_distributed() -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world() -> Tuple[int, int]:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n_total: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous env / sample range of `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_sum_(t, group=None):
    import torch.distributed as dist
    if is_distributed():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def allreduce_grads(grad_dicts: Sequence[Dict[str, "torch.Tensor"]], bucket_bytes: int = 256 << 20, average: bool = False, group=None) -> int:
    """In-place SUM (or mean) all-reduce of every tensor in the given gradient dicts, packed into flat buckets.
    Returns the number of collectives issued."""
    import torch
    import torch.distributed as dist
    if not is_distributed():
        return 0
    ws = dist.get_world_size(group)
    tensors: List["torch.Tensor"] = [g for d in grad_dicts for _, g in sorted(d.items())]
    n_coll = 0
    i = 0
    while i < len(tensors):
        j, size = i, 0
        while j < len(tensors) and (j == i or size + tensors[j].numel() * tensors[j].element_size() <= bucket_bytes):
            size += tensors[j].numel() * tensors[j].element_size()
            j += 1
        chunk = tensors[i:j]
        flat = torch.cat([t.reshape(-1) for t in chunk])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat.div_(ws)
        off = 0
        for t in chunk:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()
        n_coll += 1
        i = j
    return n_coll


def reduce_stat_partials(sums, mins, maxs, group=None):
    """All-reduce host-side stat vectors (numpy): additive entries, minima, maxima."""
    import numpy as np
    import torch
    import torch.distributed as dist
    if not is_distributed():
        return sums, mins, maxs
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    out = []
    for arr, op in ((sums, dist.ReduceOp.SUM), (mins, dist.ReduceOp.MIN), (maxs, dist.ReduceOp.MAX)):
        t = torch.as_tensor(np.asarray(arr, dtype=np.float64), device=dev).clone()
        dist.all_reduce(t, op=op, group=group)
        out.append(t.cpu().numpy())
    return tuple(out)


def whiten_distributed(x, mask, shift_mean: bool = True, group=None):
    """`whiten` over the action tokens of ALL ranks (ppo/base_interface.py:609-615): local moments on the device
    (lmrl_whiten_moments), one 3-double all-reduce, local apply (lmrl_whiten_apply)."""
    import torch
    from . import _lib
    L = _lib.lib()
    mom = torch.zeros(3, dtype=torch.float64, device=x.device)
    _lib.check(L.lmrl_whiten_moments(x.data_ptr(), _lib.ptr(mask), mom.data_ptr(), x.numel(), _lib.stream_ptr()))
    allreduce_sum_(mom, group)
    y = torch.empty_like(x)
    _lib.check(L.lmrl_whiten_apply(x.data_ptr(), _lib.ptr(mask), mom.data_ptr(), y.data_ptr(), x.numel(), int(shift_mean), _lib.stream_ptr()))
    return y
