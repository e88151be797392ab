"""A/B on one box: lmrl_gae at 65 536 / 4096 chains x 96 slots with the values pair of a lane read as ONE unaligned 8-byte streaming load (round 6)
vs two 4-byte default-policy loads (round 5, `lmrl_rl_reduce_set_variant(5)`; variant 6 forces the 8-byte form, the default picks by launch size); interleaved repetitions, one hipGraph of 40 launches over rotating
buffer sets each (as bench.py's rl_reduce leg).  -> profiles/r06_rl_reduce_vload_ab.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import lmrl_gym_amd  # noqa: F401,E402
from lmrl_gym_amd import _lib  # noqa: E402

dev = _lib.require_gpu()
Lb, sp = _lib.lib(), _lib.stream_ptr
L, iters = 96, 40
for B in (65536, 4096):
    rng = np.random.RandomState(B)
    lens = rng.randint(40, L + 1, size=B).astype(np.int32)
    t = np.arange(L)[None, :]
    sta = ((t >= 4) & (((t - 4) // 6) % 2 == 0) & (t < lens[:, None])).astype(np.uint8)
    n_sets = max(2, min(64, (640 << 20) // (B * L * 25)))
    sets = [dict(v=torch.randn(B, L + 1, device=dev), r=torch.randn(B, L, device=dev), s=torch.from_numpy(sta).to(dev), ln=torch.from_numpy(lens).to(dev),
                 adv=torch.empty(B, L, device=dev), ret=torch.empty(B, L, device=dev)) for _ in range(n_sets)]
    p = lambda x: x.data_ptr()
    fn = lambda d: _lib.check(Lb.lmrl_gae(p(d["v"]), p(d["r"]), p(d["s"]), p(d["ln"]), p(d["adv"]), p(d["ret"]), B, L, 0.99, 0.95, sp()))
    graphs = {}
    for variant in (6, 5):
        Lb.lmrl_rl_reduce_set_variant(variant)
        for d in sets[:2]:
            fn(d)
        torch.cuda.synchronize()
        st = torch.cuda.Stream(device=dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
                for i in range(iters):
                    fn(sets[i % n_sets])
        graphs[variant] = (g, st)
    Lb.lmrl_rl_reduce_set_variant(0)
    res = {6: [], 5: []}
    for rep in range(6):
        for variant in (6, 5):
            g, st = graphs[variant]
            with torch.cuda.stream(st):
                g.replay(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st); g.replay(); e1.record(st)
            torch.cuda.synchronize()
            res[variant].append(e0.elapsed_time(e1) * 1e3 / iters)
    used = int(lens.sum())
    nbytes = used * 9 + B * 8 + B * L * 8
    for variant, name in ((6, "8-byte unaligned nt pair (variant 6)"), (5, "2 x 4-byte loads (variant 5)")):
        us = float(np.median(res[variant]))
        print(f"chains {B:6d}  {name:38s} {us:7.2f} us / launch (median of 6; {' '.join(f'{x:.2f}' for x in res[variant])})  "
              f"{nbytes / us / 1e3:7.1f} GB/s = {nbytes / us / 1e3 / 8000:.3f} of 8 TB/s")
