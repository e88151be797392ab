"""The persistent LM-head kernel (1024 x 50257 x 768) by epilogue: Gumbel-max sampling without / with the log-prob, greedy without / with the log-prob
(no noise drawn: what an epilogue of one exp per logit costs), top_k = 40 (candidate epilogue).  HIP events over 40 launches, 4 hidden-state copies."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import lmrl_gym_amd  # noqa: E402,F401
from lmrl_gym_amd import _lib  # noqa: E402
from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine, SampleParams  # noqa: E402

dev = _lib.require_gpu()
eng = GPT2Engine.random_init(GPT2Config.gpt2_small(), seed=0, device=dev)
B = 1024
ses = eng.session(B, 16)
hid = [(torch.randn(B, 768, device=dev) * 3).to(torch.bfloat16) for _ in range(4)]
lo = torch.zeros(B, eng.cfg.vocab_padded, device=dev)
for rep in range(2):
    for name, temp, top_k, lp in (("sample", 1.0, 0, False), ("sample + logprob", 1.0, 0, True), ("greedy", 0.0, 0, False), ("greedy + logprob", 0.0, 0, True),
                                  ("top_k=40", 1.0, 40, False)):
        sp = SampleParams(temp, top_k, 5, 0, 0.0, 0.0, 0, None, 0.0, 0)
        kw = dict(logits_out=lo if top_k else None, want_logprob=lp)
        for i in range(5):
            ses.sample(sp, hidden=hid[i % 4], **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(40):
            ses.sample(sp, hidden=hid[i % 4], **kw)
        e1.record(); torch.cuda.synchronize()
        if rep:
            print(f"{name:18s}: {e0.elapsed_time(e1) / 40 * 1e3:7.1f} us per sampled token (LM head + reduce)", flush=True)
