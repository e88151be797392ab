import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lmrl_gym_amd
from lmrl_gym_amd import _lib
from lmrl_gym_amd.envs import wordle as W
from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine
from lmrl_gym_amd.rollout import WordleRolloutEngine
from bench import scripted_guesses
dev = _lib.require_gpu()
cfg = GPT2Config.gpt2_medium()
eng = GPT2Engine.random_init(cfg, seed=0, device=dev)
vocab = W.Vocabulary.builtin("wordle_official_400.txt")
B = 1024
ro = WordleRolloutEngine(eng, vocab, B, max_new_tokens=6)
g = torch.from_numpy(scripted_guesses(vocab.all_vocab, 4, 6, B, seed=1).view(np.int32)).to(dev)
seeds = torch.arange(4 * B, dtype=torch.int64, device=dev).view(4, B)
ro.capture_episode(temperature=1.0, sample_seed=5, steer_strength=30.0, scripted=True)
ro.replay_episode(seeds[0], g[0]); torch.cuda.synchronize()
n = torch.zeros((), dtype=torch.int64, device=dev)
t0 = time.perf_counter()
for i in range(1, 4):
    ro.replay_episode(seeds[i], g[i]); n += ro.traj["n_steps"].sum()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("GPT-2-medium (24 x 1024) Wordle rollouts, 1024 envs: %.1f env-steps/s, %.1f ms per episode" % (int(n) / dt, dt / 3 * 1e3))
# sanity: steered episodes reproduce the scripted words (token record = canonical encoding of the scripted guess)
tr = ro.token_trajectories()
ok = sum(1 for t in tr if t[3])
print("episodes done:", ok, "of", B)
