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


def _all_reduce_inplace(t, group=None, async_op: bool = False):
    """SUM all-reduce of a device tensor in place.  RCCL ("nccl") reduces device memory directly; the gloo backend of the CPU / single-GPU
    test tier stages a device tensor through the host (gloo has no ROCm device path)."""
    import torch.distributed as dist
    if dist.get_backend(group) != "gloo" or not t.is_cuda:
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    h = t.cpu()
    dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
    t.copy_(h)
    return None


def allreduce_sum_(t, group=None):
    if is_distributed():
        _all_reduce_inplace(t, group)
    return t


def _flat_chunks(grad_dicts, bucket_bytes):
    """Yield flat fp32 views to reduce: a GradArena (train/gpt2_f32.py) is ONE flat buffer and is reduced in place in `bucket_bytes`
    slices; a plain dict of tensors falls back to per-tensor reduction (tensors are not assumed adjacent in memory)."""
    for d in grad_dicts:
        flat = getattr(d, "flat", None)
        if flat is not None:
            step = max(1, bucket_bytes // flat.element_size())
            for lo in range(0, flat.numel(), step):
                yield flat[lo:lo + step]
        else:
            for _, g in sorted(d.items()):
                yield g.reshape(-1) if g.is_contiguous() else g


def allreduce_grads(grad_dicts: Sequence[Dict[str, "torch.Tensor"]], bucket_bytes: int = 256 << 20, average: bool = False, group=None) -> int:
    """In-place SUM (or mean) all-reduce of every gradient in the given dicts.  Gradient arenas are reduced in place on slices of their
    flat buffer (no torch.cat / copy-back: xGMI links are point-to-point, so few large messages); returns the number of collectives."""
    import torch.distributed as dist
    if not is_distributed():
        return 0
    ws = dist.get_world_size(group)
    n_coll = 0
    works: list = []
    for chunk in _flat_chunks(grad_dicts, bucket_bytes):
        _reduce_grad_slice(chunk, group, works)
        _finish_works(works)
        if average:
            chunk.div_(ws)
        n_coll += 1
    return n_coll


# bench-only switch: with the gradient reduction off, a data-parallel step runs the same kernels without the data-path collective, so
# `bench.py` can report the EXPOSED all-reduce time (step with - step without).  Never off in a real run: ranks would diverge.
_GRAD_REDUCE_ENABLED = True
LAST_REDUCE_BYTES = 0          # bytes handed to the gradient all-reduce by the most recent GradReducer.finish() (per rank)


def set_grad_reduce(enabled: bool) -> None:
    global _GRAD_REDUCE_ENABLED
    _GRAD_REDUCE_ENABLED = bool(enabled)


# Optional bf16-compressed gradient all-reduce (SURVEY.md §5.8 allows one): the fp32 arena slice is rounded to bf16 (RNE), the bf16 copy is
# SUM-all-reduced (half the bytes on the xGMI links: 815 -> 408 MB per ILQL step and rank) and widened back into the fp32 arena; parameters,
# optimizer state and the local gradients stay fp32.  Every rank receives the same reduced bf16 values, so ranks stay bit-identical to each
# other; against the fp32 reduction each element carries <= 2^-8 relative rounding of every addend and of the sum (tests/test_dist_cpu.py,
# tests/test_gpu_dist.py hold the written tolerance).  OFF by default: the north star asks for 1e-4 train-step parity, which is the fp32 wire.
_GRAD_COMPRESSION: Optional[str] = None


def set_grad_compression(mode: Optional[str]) -> None:
    """None / "f32": exact fp32 wire (default).  "bf16": bf16 wire format for the gradient all-reduce only."""
    global _GRAD_COMPRESSION
    assert mode in (None, "f32", "bf16"), mode
    _GRAD_COMPRESSION = None if mode in (None, "f32") else mode


def grad_compression() -> Optional[str]:
    return _GRAD_COMPRESSION


def _reduce_grad_slice(sl, group, works: list) -> int:
    """Enqueue the SUM all-reduce of one flat fp32 gradient slice (async where the backend allows it) -> bytes put on the wire.  `works`
    collects (work handle or None, completion callback or None); `_finish_works` waits and runs the callbacks (the bf16 wire format
    needs one: widen the reduced copy back into the fp32 arena)."""
    import torch
    if _GRAD_COMPRESSION == "bf16":
        wire = sl.to(torch.bfloat16)
        w = _all_reduce_inplace(wire, group, async_op=True)
        works.append((w, lambda: sl.copy_(wire)))
        return wire.numel() * wire.element_size()
    w = _all_reduce_inplace(sl, group, async_op=True)
    if w is not None:
        works.append((w, None))
    return sl.numel() * sl.element_size()


def _finish_works(works: list) -> None:
    for w, done in works:
        if w is not None:
            w.wait()
        if done is not None:
            done()
    works.clear()


class GradReducer:
    """Overlaps the data-parallel gradient all-reduce with the backward pass.  `GPT2F32.backward(..., on_final=reducer.ready(arena))`
    reports parameter groups whose gradients are final (ln_f, then block after block, then the embeddings — the arena is laid out in that
    order); whenever `bucket_bytes` of finished gradients have piled up, an asynchronous all-reduce of that arena slice is enqueued — RCCL
    runs it on its own stream behind the kernels already launched, concurrently with the rest of the backward.  `finish()` reduces what is
    left (plus any further dicts, e.g. the heads) and waits.  Without a process group every call is a no-op."""

    def __init__(self, bucket_bytes: int = 64 << 20, group=None):
        self.bucket_bytes, self.group = bucket_bytes, group
        self.works, self.n_coll, self.n_bytes = [], 0, 0
        self._arena, self._lo, self._hi = None, 0, 0

    def _flush(self):
        if self._arena is not None and self._hi > self._lo:
            self.n_bytes += _reduce_grad_slice(self._arena.flat[self._lo:self._hi], self.group, self.works)
            self.n_coll += 1
            self._lo = self._hi

    def ready(self, arena):
        """The `on_final` callback for one arena."""
        if not is_distributed() or not _GRAD_REDUCE_ENABLED:
            return None
        self._arena, self._lo, self._hi = arena, 0, 0

        def on_final(names):
            lo, hi = arena.span(names)
            assert lo == self._hi, "gradients must become final in arena order"
            self._hi = hi
            if (self._hi - self._lo) * arena.flat.element_size() >= self.bucket_bytes:
                self._flush()
        return on_final

    def early(self, done: Sequence[Dict[str, "torch.Tensor"]]):
        """Gradient arenas that are ALREADY final before the transformer backward starts (the heads': their backward runs first) — their all-reduce
        is enqueued now and runs under the whole base backward instead of after it (`finish(more=...)` reduced them last, fully exposed: 310 MB of
        the 815 MB of an ILQL step).  A plain dict (no `.flat`) is left to `finish`."""
        if not is_distributed() or not _GRAD_REDUCE_ENABLED:
            return []
        later = []
        for d in done:
            flat = getattr(d, "flat", None)
            if flat is None:
                later.append(d)
                continue
            for lo in range(0, flat.numel(), max(1, self.bucket_bytes // flat.element_size())):
                sl = flat[lo:lo + max(1, self.bucket_bytes // flat.element_size())]
                self.n_bytes += _reduce_grad_slice(sl, self.group, self.works)
                self.n_coll += 1
        return later

    def finish(self, more: Sequence[Dict[str, "torch.Tensor"]] = ()):
        global LAST_REDUCE_BYTES
        if not is_distributed() or not _GRAD_REDUCE_ENABLED:
            return 0
        if self._arena is not None:
            self._hi = self._arena.flat.numel()              # whatever has not been handed over yet (normally the last partial bucket)
            self._flush()
        for chunk in _flat_chunks(more, 256 << 20):
            self.n_bytes += _reduce_grad_slice(chunk, self.group, self.works)
            self.n_coll += 1
        _finish_works(self.works)
        LAST_REDUCE_BYTES = self.n_bytes
        return self.n_coll


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


def whiten_distributed(x, mask, shift_mean: bool = True, group=None, partials=None):
    """`whiten` over the action tokens of ALL ranks (ppo/base_interface.py:609-615): local moments on the device
    (lmrl_whiten_moments), one 3-double all-reduce, local apply (lmrl_whiten_apply).
    partials (float64 [n, 3], optional): the per-workgroup partial moments `lmrl_gae_moments` left beside these advantages — no moments pass
    re-reads x: one rank applies straight from the partials (one launch), several ranks finish them (lmrl_whiten_finish) and all-reduce."""
    import torch
    from . import _lib
    L = _lib.lib()
    if partials is not None and not is_distributed():
        y = torch.empty_like(x)
        _lib.check(L.lmrl_whiten_apply_partials(x.data_ptr(), _lib.ptr(mask), partials.data_ptr(), int(partials.shape[0]), y.data_ptr(), x.numel(),
                                                int(shift_mean), _lib.stream_ptr()), "lmrl_whiten_apply_partials")
        return y
    mom = torch.zeros(3, dtype=torch.float64, device=x.device)
    if partials is not None:
        _lib.check(L.lmrl_whiten_finish(partials.data_ptr(), int(partials.shape[0]), mom.data_ptr(), _lib.stream_ptr()), "lmrl_whiten_finish")
    else:
        _lib.check(L.lmrl_whiten_moments(x.data_ptr(), _lib.ptr(mask), mom.data_ptr(), x.numel(), _lib.stream_ptr()))
    allreduce_sum_(mom, group)
    y = torch.empty_like(x)
    _lib.check(L.lmrl_whiten_apply(x.data_ptr(), _lib.ptr(mask), mom.data_ptr(), y.data_ptr(), x.numel(), int(shift_mean), _lib.stream_ptr()))
    return y
