"""cProfile of the generic text-policy Maze path (interact_environment + GPT2PPOPolicy.act + VectorMazeEnv) at 1024 envs: where the host time of a lock-step turn goes."""
import os, sys, time, cProfile, pstats
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
B = 1024
pol = GPT2PPOPolicy(eng, tok, max_input_length=160, max_new_tokens=12, do_sample=True, seed=1, eos_token_id=tok.eos_token_id, out_str_process=lambda x: x.removesuffix("\n") + "\n")
env = M.setup_maze_env("double_t_maze", "describe_observation_give_position", "standard_reward", last_k=1, max_steps=20)
E.interact_environment(env, pol, env_seed=list(range(B)), bsize=B)
torch.cuda.synchronize(); t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
inter = E.interact_environment(env, pol, env_seed=list(range(100, 100 + B)), bsize=B)
torch.cuda.synchronize(); pr.disable(); dt = time.perf_counter() - t0
steps = sum(len(ep) for ep in inter)
print("B=%d: %d env steps in %.2f s -> %.0f env-steps/s" % (B, steps, dt, steps / dt))
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
