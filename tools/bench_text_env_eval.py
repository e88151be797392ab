import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import lmrl_gym_amd
from lmrl_gym_amd import _lib
from lmrl_gym_amd.envs import wordle as W
from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine
from lmrl_gym_amd.rollout import WordleRolloutEngine
from bench import scripted_guesses
dev = _lib.require_gpu()
eng = GPT2Engine.random_init(GPT2Config.gpt2_small(), seed=0, device=dev)
vocab = W.Vocabulary.builtin("wordle_official_400.txt")
B = 1024
ro = WordleRolloutEngine(eng, vocab, B, max_new_tokens=6)
g = torch.from_numpy(scripted_guesses(vocab.all_vocab, 8, 6, B, seed=1).view(np.int32)).to(dev)
kw = dict(scripted_guesses_fn=lambda bid: g[bid % 8], steer_strength=30.0, temperature=1.0, sample_seed=9, use_graph=True)
gen = iter(range(10**6, 10**9))
ro.text_env_eval(B, seed_generator=gen, **kw)
for nb in (4, 8, 4, 8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    inter, _ = ro.text_env_eval(nb * B, seed_generator=gen, **kw)
    dt = time.perf_counter() - t0
    print(nb, "batches:", round(sum(len(e) for e in inter) / dt), "env-steps/s", round(dt * 1e3 / nb, 1), "ms per batch", flush=True)
    del inter
