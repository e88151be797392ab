"""LM-head sampler micro-benchmark: greedy (no RNG) vs Gumbel sampling at M=1024, GPT-2-small vocabulary."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lmrl_gym_amd  # noqa
from lmrl_gym_amd import _lib
from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine, SampleParams
dev = _lib.require_gpu()
cfg = GPT2Config(1, 12, 768, 3072, 50257, 64)
eng = GPT2Engine.random_init(cfg, seed=0, device=dev)
B = 1024
ses = eng.session(B, 8)
hid = torch.randn(B, 768, device=dev).to(torch.bfloat16)
L = _lib.lib()
for variant, vname in [(0, "persistent kernel (default)"), (301, "one tile per workgroup")]:
    L.lmrl_gemm_set_variant(variant)
    for name, temp, lp in [("greedy", 0.0, False), ("gumbel", 1.0, False), ("gumbel + log-prob", 1.0, True)]:
        for _ in range(3):
            ses.sample(SampleParams(temp, 0, 1, 0, 0.0, 0.0, 0), hidden=hid, want_logprob=lp)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(50):
            ses.sample(SampleParams(temp, 0, 1, k, 0.0, 0.0, 0), hidden=hid, want_logprob=lp)
        e1.record(); torch.cuda.synchronize()
        print("%-32s %-20s %.1f us per call (lm_head + reduce, back to back)" % (vname, name, e0.elapsed_time(e1) * 1e3 / 50))
L.lmrl_gemm_set_variant(0)
