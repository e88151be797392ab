"""Maze rollouts on the device-resident loop (MazeRolloutEngine: prompt-prefix cache + per-turn hipGraph) — same workload as
tools/bench_maze_generic.py: GPT-2-small random init, byte-level stand-in tokenizer, `describe_observation_give_position`,
B envs, max_steps 20, max_new_tokens 12."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lmrl_gym_amd
from lmrl_gym_amd import _lib, datasets as DS
from lmrl_gym_amd.envs import maze as M
from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine
from lmrl_gym_amd.maze_rollout import MazeRolloutEngine
dev = _lib.require_gpu()
tok = DS.ByteTokenizer()
eng = GPT2Engine.random_init(GPT2Config.gpt2_small(), seed=0, device=dev)
for B, cache, indexed in ((256, True, True), (1024, True, True), (1024, True, False), (1024, False, True), (4096, True, True), (4096, True, False)):
    env = M.setup_maze_env("double_t_maze", "describe_observation_give_position", "standard_reward", last_k=1, max_steps=20)
    r = MazeRolloutEngine(eng, tok, env, B, max_new_tokens=12, eos_token_id=tok.eos_token_id, max_input_length=160, prefix_cache=cache,
                          prefix_indexed=indexed)
    r.run_episode(list(range(B)), sample_seed=1, use_graph=True, sync_every=0)           # capture + warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 3
    for e in range(reps):
        r.run_episode(list(range(100 + e * B, 100 + (e + 1) * B)), sample_seed=1, episode=e + 1, use_graph=True, sync_every=0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    steps = int(r.traj["n_turns"].sum().item())
    print("B=%5d prefix_cache=%-5s indexed=%-5s rows=%d prompt<=%d tok: %6d env steps in %.3f s -> %.0f env-steps/s (%.2f ms per lock-step turn)"
          % (B, cache, indexed and cache, r.obs_tok_h.shape[0], r.max_obs_len, steps, dt, steps / dt, dt / r.T * 1e3), flush=True)
    r.close()

# text_env_eval end to end (host InteractionTransition lists included) with 1 / 2 / 3 episode batches in flight (concurrent=n: twin engines, own streams)
B = 1024
env = M.setup_maze_env("double_t_maze", "describe_observation_give_position", "standard_reward", last_k=1, max_steps=20)
r = MazeRolloutEngine(eng, tok, env, B, max_new_tokens=12, eos_token_id=tok.eos_token_id, max_input_length=160)
gen = iter(range(10 ** 6, 10 ** 9))
for lanes in (1, 2, 3):
    r.text_env_eval(lanes * B, seed_generator=gen, sample_seed=1, concurrent=lanes, sync_every=0)      # twins, graphs
    torch.cuda.synchronize(); t0 = time.perf_counter()
    inter, _ = r.text_env_eval(6 * B, seed_generator=gen, sample_seed=1, concurrent=lanes, sync_every=0)
    dt = time.perf_counter() - t0
    print("text_env_eval(6 x %d, concurrent=%d): %.0f env-steps/s incl. host lists (%.1f ms per batch)" % (B, lanes, sum(len(e) for e in inter) / dt, dt * 1e3 / 6), flush=True)
    del inter
r.close()
