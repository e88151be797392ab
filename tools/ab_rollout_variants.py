"""In-situ A/B of GEMM tile-policy variants (`lmrl_gemm_set_variant`, tools-only hook of csrc/gemm_dispatch.h) on the bench episode: 1024 envs,
GPT-2-small, hipGraph replay; each variant gets a fresh engine + capture, two passes (the second is reported).

    python tools/ab_rollout_variants.py [variant ...]        (default: 0 201 203)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import lmrl_gym_amd  # noqa: E402,F401
from lmrl_gym_amd import _lib  # noqa: E402
from lmrl_gym_amd.envs import wordle as W  # noqa: E402
from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine  # noqa: E402
from lmrl_gym_amd.rollout import WordleRolloutEngine  # noqa: E402
from bench import scripted_guesses  # noqa: E402

variants = [int(v) for v in sys.argv[1:]] or [0, 201, 203]
dev = _lib.require_gpu()
L = _lib.lib()
eng = GPT2Engine.random_init(GPT2Config.gpt2_small(), seed=0, device=dev)
vocab = W.Vocabulary.builtin("wordle_official_400.txt")
B, N = 1024, 8
g = torch.from_numpy(scripted_guesses(vocab.all_vocab, N + 1, 6, B, seed=1).view(np.int32)).to(dev)
seeds = torch.arange((N + 1) * B, dtype=torch.int64, device=dev).view(N + 1, B)
base = None
for rep in range(2):
    for v in variants:
        L.lmrl_gemm_set_variant(v)
        ro = WordleRolloutEngine(eng, vocab, B, max_new_tokens=6, bad_word_reward=-10.0)
        ro.capture_episode(temperature=1.0, sample_seed=5, steer_strength=30.0, scripted=True)
        ro.replay_episode(seeds[0], g[0]); torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = []
        for i in range(1, N + 1):
            ro.replay_episode(seeds[i], g[i])
            n.append(ro.traj["n_steps"].sum())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        steps = int(torch.stack(n).sum().item())
        if v == variants[0]:
            base = dt
        if rep == 1:
            print(f"variant {v:4d}: {dt / N * 1e3:7.3f} ms per episode  {steps / dt:9.0f} env-steps/s  ({(base - dt) / N * 1e3:+.3f} ms vs variant {variants[0]})", flush=True)
        ro.close()
        del ro
        L.lmrl_gemm_set_variant(0)
