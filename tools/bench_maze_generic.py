"""Maze rollouts through the generic text-policy path (GPT2PPOPolicy.act + VectorMazeEnv via interact_environment):
GPT-2-small random init, byte-level stand-in tokenizer, `describe_observation_give_position`, B envs, max_steps 20."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lmrl_gym_amd
from lmrl_gym_amd import _lib, datasets as DS, environment as E
from lmrl_gym_amd.envs import maze as M
from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine
from lmrl_gym_amd.policies import GPT2PPOPolicy
dev = _lib.require_gpu()
tok = DS.ByteTokenizer()
eng = GPT2Engine.random_init(GPT2Config.gpt2_small(), seed=0, device=dev)
for B in (32, 256):
    pol = GPT2PPOPolicy(eng, tok, max_input_length=160, max_new_tokens=12, do_sample=True, seed=1, eos_token_id=tok.eos_token_id,
                        out_str_process=lambda x: x.removesuffix("\n") + "\n")
    env = M.setup_maze_env("double_t_maze", "describe_observation_give_position", "standard_reward", last_k=1, max_steps=20)
    E.interact_environment(env, pol, env_seed=list(range(B)), bsize=B)          # warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    inter = E.interact_environment(env, pol, env_seed=list(range(100, 100 + B)), bsize=B)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    steps = sum(len(ep) for ep in inter)
    print("B=%4d: %6d env steps in %.2f s -> %.0f env-steps/s (%.1f ms per lock-step turn)" % (B, steps, dt, steps / dt, dt / max(len(ep) for ep in inter) * 1e3))
