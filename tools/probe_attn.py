import sys; sys.path.insert(0, ".")
import numpy as np, torch
import lmrl_gym_amd
from lmrl_gym_amd import _lib
from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine
dev = _lib.require_gpu(); L = _lib.lib()
cfg = GPT2Config(1, 2, 128, 512, 1000, 64)
eng = GPT2Engine.random_init(cfg, seed=1, device=dev)
B = 5
g = torch.Generator().manual_seed(0)
outs = {}
for variant in (1, 0):
    ses = eng.session(B, 48, flags=8 if variant else 0)      # FWD_ATTN_VALU: VALU chunk attention as the cross-check
    res = []
    for C, cnts in [(8, [8, 5, 1, 0, 7]), (8, [8, 8, 3, 8, 2])]:
        toks = torch.randint(0, 1000, (B * C,), generator=torch.Generator().manual_seed(C + sum(cnts))).to(torch.int32).to(dev)
        allh = torch.zeros(B * C, cfg.d_model, dtype=torch.bfloat16, device=dev)
        ses.forward(toks, torch.tensor(cnts, dtype=torch.int32, device=dev), C, all_hidden=allh)
        res.append(allh.float().cpu().reshape(B, C, -1))
    outs[variant] = res
for step in range(2):
    a, b = outs[1][step], outs[0][step]
    for bi in range(B):
        for j in range(8):
            d = (a[bi, j] - b[bi, j]).abs().max().item()
            nan = torch.isnan(b[bi, j]).sum().item()
            print(step, bi, j, "maxdiff %.4f nan %d" % (d, nan))
