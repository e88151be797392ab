"""The partially observed Maze task (llm_rl_scripts/maze/ppo/partially_observed_ppo_online.py:72-83,341: `last_k = 40`, max_input_length 512,
max_output_length 10) through the text-policy path with K/V reuse across `act()` calls: how much of a turn is re-prefill?

The prompt of a turn is the concatenation of the last 40 history items, LEFT-truncated to 512 tokens.  While it only grows, a turn forwards its new
tokens; once it is truncated (byte-level stand-in tokenizer: after ~8 turns; GPT-2 BPE: once 40 items are held) every token shifts position, and
GPT-2's learned ABSOLUTE position embeddings enter every layer's K / V: nothing of the cache can be kept or re-based exactly — the turn costs a full
re-prefill whatever loop drives it (device-resident or host).  Prints env-steps/s and the tokens forwarded per env-step in both regimes.

    python tools/bench_maze_partially_observed.py [--envs 256] [--steps 24]
"""
import argparse
import random
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import lmrl_gym_amd  # noqa: E402,F401
from lmrl_gym_amd import _lib, datasets as DS, environment as E  # noqa: E402
from lmrl_gym_amd.envs import maze as M  # noqa: E402
from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine  # noqa: E402
from lmrl_gym_amd.policies import GPT2PPOPolicy  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=256)
ap.add_argument("--steps", type=int, default=24)
a = ap.parse_args()
dev = _lib.require_gpu()
tok = DS.ByteTokenizer()
eng = GPT2Engine.random_init(GPT2Config.gpt2_small(), seed=0, device=dev)
B = a.envs
for max_steps, label in ((6, "prompt still growing (first 6 turns)"), (a.steps, f"{a.steps} turns: the prompt is left-truncated from turn ~8 on")):
    # a random-init policy never spells a move, and an unknown action makes the env drop the history (maze/env/env.py:179-180): the model still
    # generates its 10 tokens, but the text handed to the env is a random VALID move, so that the history grows as a trained policy's does
    rng = random.Random(0)
    pol = GPT2PPOPolicy(eng, tok, max_input_length=512, max_new_tokens=10, do_sample=True, seed=1, eos_token_id=tok.eos_token_id,
                        out_str_process=lambda x: rng.choice(("move left\n", "move right\n", "move up\n", "move down\n")))
    env = M.setup_maze_env("double_t_maze", "describe_observation_only_walls", "standard_reward", last_k=40, max_steps=max_steps)
    E.interact_environment(env, pol, env_seed=list(range(B)), bsize=B)          # warm-up
    pol._gen.prefilled_tokens = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    inter = E.interact_environment(env, pol, env_seed=list(range(100, 100 + B)), bsize=B)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    steps = sum(len(ep) for ep in inter)
    turns = max(len(ep) for ep in inter)
    print(f"last_k = 40, B = {B}, {label}: {steps} env steps in {dt:.2f} s -> {steps / dt:.0f} env-steps/s ({dt / turns * 1e3:.1f} ms per lock-step turn), "
          f"{pol._gen.prefilled_tokens / steps:.1f} prompt tokens forwarded per env-step", flush=True)
