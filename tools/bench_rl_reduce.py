"""M5 micro-benchmark: GAE / reward-to-go / whitening kernels on 4096 (and 65536) token chains of the
rollout's shape (L = 96 token slots, ~36 action tokens per chain), device-resident inputs.
Algorithmic bytes: GAE 4 B value + 4 B reward + 1 B flag read, 8 B written per token slot; RTG 5 B read, 4 B written;
whiten 5 B read per pass (2 passes) + 4 B written."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lmrl_gym_amd  # noqa
from lmrl_gym_amd import _lib
dev = _lib.require_gpu(); L_ = _lib.lib(); sp = _lib.stream_ptr


def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


for B in (4096, 65536, 1048576):
    L = 96
    rng = np.random.RandomState(0)
    lens = rng.randint(40, L + 1, size=B).astype(np.int32)
    sta = ((np.arange(L)[None, :] >= 4) & (((np.arange(L)[None, :] - 4) // 6) % 2 == 0) & (np.arange(L)[None, :] < lens[:, None])).astype(np.uint8)
    v = torch.randn(B, L + 1, device=dev); r = torch.randn(B, L, device=dev)
    s = torch.from_numpy(sta).to(dev); ln = torch.from_numpy(lens).to(dev)
    adv, ret, rtg = torch.empty(B, L, device=dev), torch.empty(B, L, device=dev), torch.empty(B, L, device=dev)
    mom = torch.zeros(3, dtype=torch.float64, device=dev)
    n = B * L
    t_gae = timeit(lambda: _lib.check(L_.lmrl_gae(v.data_ptr(), r.data_ptr(), s.data_ptr(), ln.data_ptr(), adv.data_ptr(), ret.data_ptr(), B, L, 0.99, 0.95, sp())))
    t_rtg = timeit(lambda: _lib.check(L_.lmrl_rtg(r.data_ptr(), s.data_ptr(), ln.data_ptr(), rtg.data_ptr(), B, L, 0.99, sp())))
    def wh():
        _lib.check(L_.lmrl_whiten_moments(adv.data_ptr(), s.data_ptr(), mom.data_ptr(), n, sp()))
        _lib.check(L_.lmrl_whiten_apply(adv.data_ptr(), s.data_ptr(), mom.data_ptr(), ret.data_ptr(), n, 1, sp()))
    t_wh = timeit(wh)
    print("chains %8d x %d slots: GAE %8.1f us %7.1f GB/s | RTG %8.1f us %7.1f GB/s | whiten %8.1f us %7.1f GB/s | %.1f M action tokens/s (GAE)" % (
        B, L, t_gae * 1e6, n * 17 / t_gae / 1e9, t_rtg * 1e6, n * 9 / t_rtg / 1e9, t_wh * 1e6, n * 14 / t_wh / 1e9, sta.sum() / t_gae / 1e6))
