"""Distinct GEMM shapes of one ILQL train step (bf16-matmul mode) with per-shape GPU time: monkeypatches MatmulBF16.gemm / ops.sgemm with
event-timed wrappers (serialises the step; the sum is not the step time).  usage: python tools/log_train_shapes.py [ilql-step|ppo-step]"""
import collections
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from lmrl_gym_amd.train import ops

rec = collections.OrderedDict()


def timed(key, fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    e1.synchronize()
    c = rec.setdefault(key, [0, 0.0])
    c[0] += 1
    c[1] += e0.elapsed_time(e1) * 1e3
    return r


_g = ops.MatmulBF16.gemm
_s = ops.sgemm


def gemm(self, a, w, bias, c, m, n, k, ldc, n_store, accumulate=False, resid=None):
    return timed(("bf16", m, n, k, bias is not None, accumulate, resid is not None),
                 lambda: _g(self, a, w, bias, c, m, n, k, ldc, n_store, accumulate=accumulate, resid=resid))


def sgemm(a, b, c, m, n, k, **kw):
    return timed(("f32", m, n, k, bool(kw.get("trans_a")), bool(kw.get("trans_b")), kw.get("batch")), lambda: _s(a, b, c, m, n, k, **kw))


mode = sys.argv[1] if len(sys.argv) > 1 else "ilql-step"
matmul = sys.argv[2] if len(sys.argv) > 2 else "bf16"
world, rank, dev, backend, use_dist = bench._dist_setup(torch)
bench.run_train_step(mode, matmul, 32, 1, 1, dev, 0, 1, False, "nccl")
rec.clear()
ops.MatmulBF16.gemm = gemm
ops.sgemm = sgemm
import lmrl_gym_amd.train.gpt2_f32 as G
bench.run_train_step(mode, matmul, 32, 1, 0, dev, 0, 1, False, "nccl")
tot = sum(v[1] for v in rec.values())
for k, v in sorted(rec.items(), key=lambda kv: -kv[1][1]):
    fl = 2.0 * k[1] * k[2] * k[3] * v[0]
    print(f"{str(k):70s} n={v[0]:4d} total {v[1]:9.1f} us  avg {v[1] / v[0]:8.1f} us  {fl / v[1] / 1e6:7.1f} TFLOP/s")
print("sum", tot)
