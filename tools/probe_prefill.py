"""Chunk-prefill forwards alone (B = 1024 envs, C = 8 slots, 7 tokens each, contexts as in a Wordle episode) for rocprofv3 --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lmrl_gym_amd
from lmrl_gym_amd import _lib
from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine
dev = _lib.require_gpu()
eng = GPT2Engine.random_init(GPT2Config.gpt2_small(), seed=0, device=dev)
B, C = 1024, 8
ses = eng.session(B, 128, flags=int(os.environ.get("FLAGS", "0")))
tok = torch.randint(0, 50257, (B * C,), dtype=torch.int32, device=dev)
cnt = torch.full((B,), int(os.environ.get("CNT", "7")), dtype=torch.int32, device=dev)
for rep in range(4):
    ses.reset()
    for turn in range(6):
        ses.forward(tok, cnt, C)
torch.cuda.synchronize()
print("done")
