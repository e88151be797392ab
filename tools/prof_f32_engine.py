"""The bench workload (1024 envs, GPT-2-small, steered sampling) on `GPT2EngineF32` in one arithmetic mode — eager launches or hipGraph replay;
prints ms per episode and env-steps/s.  Run it under `rocprofv3 --kernel-trace --stats` for the per-kernel table of that mode
(the default bench line mixes three engines in one process).

    python tools/prof_f32_engine.py --matmul bf16x3|f32 [--graph 1] [--episodes 3] [--batch 1024]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import lmrl_gym_amd  # noqa: E402,F401
from lmrl_gym_amd import _lib  # noqa: E402
from lmrl_gym_amd.envs import wordle as W  # noqa: E402
from lmrl_gym_amd.gpt2 import GPT2Config  # noqa: E402
from lmrl_gym_amd.gpt2_f32_engine import GPT2EngineF32  # noqa: E402
from lmrl_gym_amd.rollout import WordleRolloutEngine  # noqa: E402
from bench import scripted_guesses  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--matmul", default="bf16x3", choices=["f32", "bf16x3"])
ap.add_argument("--graph", type=int, default=1)
ap.add_argument("--episodes", type=int, default=3)
ap.add_argument("--batch", type=int, default=1024)
a = ap.parse_args()
dev = _lib.require_gpu()
eng = GPT2EngineF32.random_init(GPT2Config.gpt2_small(), seed=0, device=dev, matmul=a.matmul)
vocab = W.Vocabulary.builtin("wordle_official_400.txt")
B, N = a.batch, a.episodes
g = torch.from_numpy(scripted_guesses(vocab.all_vocab, N + 1, 6, B, seed=1).view(np.int32)).to(dev)
seeds = torch.arange((N + 1) * B, dtype=torch.int64, device=dev).view(N + 1, B)
ro = WordleRolloutEngine(eng, vocab, B, max_new_tokens=6, bad_word_reward=-10.0)
kw = dict(temperature=1.0, sample_seed=5, steer_strength=30.0)
if a.graph:
    ro.capture_episode(scripted=True, **kw)
    run = lambda i: ro.replay_episode(seeds[i], g[i])
else:
    run = lambda i: ro.run_episode(seeds[i], scripted_guesses=g[i], **kw)
run(0); torch.cuda.synchronize()
t0 = time.perf_counter()
n = []
for i in range(1, N + 1):
    run(i)
    n.append(ro.traj["n_steps"].sum())
torch.cuda.synchronize()
dt = time.perf_counter() - t0
steps = int(torch.stack(n).sum().item())
print(f"GPT2EngineF32(matmul={a.matmul}) {'hipGraph replay' if a.graph else 'eager'}: {dt / N * 1e3:.2f} ms per {B}-env episode, {steps / dt:.0f} env-steps/s", flush=True)
ro.close()
