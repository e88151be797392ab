"""The partially observed Maze task on the DEVICE loop (round 6): MazeEnv(last_k = 40) — llm_rl_scripts/maze/bc/partially_observed_bc.py:241,
ppo/partially_observed_ppo_online.py:72-83 (max_input_length 512, max_output_length 10) — with the item window on a persistent per-env KV cache
(`MazeRolloutEngine`, csrc/maze_tokens.hip lmrl_maze_hist_*): append turns forward the action's tail + the new observation, re-prefill turns the
whole window.  GPT-2-small, byte tokenizer, the sampler steered towards random LEGAL moves (`set_scripted_actions`: a random-init policy never
spells one, and an illegal string restarts the window, env.py:179-180), per-turn hipGraphs.  Prints env-steps/s for
  (a) episodes that end before the window slides or overflows (every turn appends),
  (b) longer episodes (re-prefill turns once 512 tokens are reached),
and the generic host path (interact_environment + GPT2PPOPolicy with K/V reuse across act() calls, tools/bench_maze_partially_observed.py) beside it.

    python tools/bench_maze_history.py [--envs 1024]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import lmrl_gym_amd  # noqa: E402,F401
from lmrl_gym_amd import _lib, datasets as DS  # noqa: E402
from lmrl_gym_amd.envs import maze as M  # noqa: E402
from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine  # noqa: E402
from lmrl_gym_amd.maze_rollout import MazeRolloutEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, nargs="+", default=[1024, 256])
a = ap.parse_args()
dev = _lib.require_gpu()
tok = DS.ByteTokenizer()
eng = GPT2Engine.random_init(GPT2Config.gpt2_small(), seed=0, device=dev)
for B in a.envs:
    for max_steps, max_input, label in ((6, 512, "7 turns, window growing (append turns only)"), (24, 512, "25 turns, max_input_length 512 (re-prefill turns once the bound reaches it)"),
                                        (19, 1024, "20 turns, max_input_length 1024 (append turns only: 39 items, <= 1024 tokens)")):
        env = M.setup_maze_env("double_t_maze", "describe_observation_only_walls", "standard_reward", last_k=40, max_steps=max_steps)
        ro = MazeRolloutEngine(eng, tok, env, B, max_new_tokens=12, eos_token_id=tok.eos_token_id, max_input_length=max_input)
        rng = np.random.RandomState(B + max_steps)
        ro.set_scripted_actions(rng.randint(0, 4, size=(ro.T, B)), strength=30.0)
        n_app = sum(ro._append_turn[:ro.T])
        seeds = list(range(B))
        ro.run_episode(seeds, None, temperature=1.0, sample_seed=3, episode=0, use_graph=True, sync_every=0)      # capture + warm-up
        torch.cuda.synchronize()
        reps, steps = 3, 0
        t0 = time.perf_counter()
        for r in range(reps):
            ro.run_episode([s + 1000 * (r + 1) for s in seeds], None, temperature=1.0, sample_seed=3, episode=r + 1, use_graph=True, sync_every=0)
            steps += int(ro.traj["n_turns"].sum().item())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        kinds = ro.traj["kind"].cpu().numpy()
        nt = ro.traj["n_turns"].cpu().numpy()
        legal = float(np.mean([np.mean(kinds[b, :nt[b]] != 3) for b in range(B) if nt[b] > 0]))
        print(f"last_k = 40, B = {B:5d}, {label}: {steps / dt:9.0f} env-steps/s ({dt / reps / ro.T * 1e3:6.2f} ms per lock-step turn; {n_app} append / {ro.T - n_app} "
              f"re-prefill turns; legal moves {legal:.2f}; flags {ro.history_flags()})", flush=True)
        ro.close()
