"""One warm + two timed fp32 episodes (GPT2EngineF32, 1024 envs, steered) for `rocprofv3 --kernel-trace --stats`; prints the wall time too."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lmrl_gym_amd  # noqa
from lmrl_gym_amd import _lib
from lmrl_gym_amd.envs import wordle as W
from lmrl_gym_amd.gpt2 import GPT2Config
from lmrl_gym_amd.gpt2_f32_engine import GPT2EngineF32
from lmrl_gym_amd.rollout import WordleRolloutEngine
from bench import scripted_guesses
dev = _lib.require_gpu()
MM = os.environ.get("MATMUL", "f32")
eng = GPT2EngineF32.random_init(GPT2Config.gpt2_small(), seed=0, device=dev, matmul=MM)
vocab = W.Vocabulary.builtin("wordle_official_400.txt")
B = 1024
ro = WordleRolloutEngine(eng, vocab, B, max_new_tokens=6)
g = torch.from_numpy(scripted_guesses(vocab.all_vocab, 3, 6, B, seed=1).view(np.int32)).to(dev)
seeds = torch.arange(3 * B, dtype=torch.int64, device=dev).view(3, B)
ro.run_episode(seeds[0], scripted_guesses=g[0], steer_strength=30.0); torch.cuda.synchronize()
t0 = time.perf_counter()
n = 0
for i in (1, 2):
    ro.run_episode(seeds[i], scripted_guesses=g[i], steer_strength=30.0)
    n += int(ro.traj["n_steps"].sum())
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{MM} rollout: {n / dt:.0f} env-steps/s, {dt / 2 * 1e3:.1f} ms per episode")
