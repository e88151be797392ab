"""configs[2] rollouts: ILQL value policy (pi_beta + beta*min(Q1,Q2); two GPT-2-small transformers + two MLP Q heads 768->768->V)
on the device-resident Wordle loop, 1024 envs, steered synthetic workload as in bench.py.  Prints env-steps/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lmrl_gym_amd  # noqa
from lmrl_gym_amd import _lib
from lmrl_gym_amd.envs import wordle as W
from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine
from lmrl_gym_amd.policies import heads_to_engine_layout
from lmrl_gym_amd.rollout import WordleRolloutEngine
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import scripted_guesses

dev = _lib.require_gpu()
cfg = GPT2Config.gpt2_small()
pi_beta, base = GPT2Engine.random_init(cfg, seed=0, device=dev), GPT2Engine.random_init(cfg, seed=1, device=dev)
g = torch.Generator().manual_seed(0)
d, V = cfg.d_model, cfg.vocab
mk = lambda: heads_to_engine_layout({"dense1.kernel": torch.randn(d, d, generator=g) * 0.02, "dense1.bias": torch.zeros(d),
                                     "dense2.kernel": torch.randn(d, V, generator=g) * 0.002, "dense2.bias": torch.full((V,), -4.4)}, cfg.vocab_padded, dev)
vocab = W.Vocabulary.builtin("wordle_official_400.txt")
B, steps, warm = 1024, 4, 1
ro = WordleRolloutEngine(pi_beta, vocab, B, max_new_tokens=6, bad_word_reward=-10.0, value_engine=base, q1_head=mk(), q2_head=mk(), beta=32.0)
guesses = torch.from_numpy(scripted_guesses(vocab.all_vocab, steps + warm, W.N_TRIES, B, seed=1).view(np.int32)).to(dev)
seeds = torch.arange((steps + warm) * B, dtype=torch.int64, device=dev).view(steps + warm, B)
ro.capture_episode(temperature=1.0, sample_seed=5, steer_strength=30.0 + 32.0 * 5, scripted=True)
torch.cuda.synchronize()
n = torch.zeros((), dtype=torch.int64, device=dev)
for i in range(warm):
    ro.replay_episode(seeds[i], guesses[i])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(warm, warm + steps):
    ro.replay_episode(seeds[i], guesses[i]); n += ro.traj["n_steps"].sum()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("ILQL value-policy rollouts: %.1f env-steps/s  (%.2f ms per 1024-env episode, %d env steps)" % (int(n) / dt, dt * 1e3 / steps, int(n)))
if "plain" in sys.argv[1:]:          # profiling: the plain-sampling leg alone
    sys.exit(0)

# round 6: the task script's sampler — top-k on the perturbed logits (train_ilql_gpt2.py:384-403 `policy_top_k`, generation.py:97-119) — on the FUSED
# candidate path of the three-operand head (no [B, V] logits in HBM) vs the materialised path of the same head (lmrl_sampler_set_variant(2))
for name, variant, kw_s in (("top_k=40 fused (candidates in the LM-head epilogue)", 0, dict(top_k=40)), ("top_k=40 materialised logits", 2, dict(top_k=40)),
                            ("top_k=40 + top_p=0.95 fused", 0, dict(top_k=40, top_p=0.95))):
    _lib.lib().lmrl_sampler_set_variant(variant)
    try:
        ro2 = WordleRolloutEngine(pi_beta, vocab, B, max_new_tokens=6, bad_word_reward=-10.0, value_engine=base, q1_head=ro.q1, q2_head=ro.q2, beta=32.0)
        ro2.capture_episode(temperature=1.0, sample_seed=5, steer_strength=30.0 + 32.0 * 5, scripted=True, **kw_s)
        n2 = torch.zeros((), dtype=torch.int64, device=dev)
        ro2.replay_episode(seeds[0], guesses[0])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(warm, warm + steps):
            ro2.replay_episode(seeds[i], guesses[i]); n2 += ro2.traj["n_steps"].sum()
        torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
        print("  %-52s %.1f env-steps/s  (%.2f ms per episode)" % (name, int(n2) / dt2, dt2 * 1e3 / steps))
        ro2.close()
    finally:
        _lib.lib().lmrl_sampler_set_variant(0)

# the public call, host lists included, 1 / 2 lanes (independent 1024-env batches in flight; each lane runs its two transformers on two streams)
kw = dict(scripted_guesses_fn=lambda bid: guesses[bid % (steps + warm)], steer_strength=30.0 + 32.0 * 5, temperature=1.0, sample_seed=5, use_graph=True)
gen = iter(range(10 ** 6, 10 ** 9))
for lanes in (1, 2):
    ro.text_env_eval(lanes * B, seed_generator=gen, concurrent=lanes, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    inter, _ = ro.text_env_eval(8 * B, seed_generator=gen, concurrent=lanes, **kw)
    dt = time.perf_counter() - t0
    print("text_env_eval(8 x B, concurrent=%d): %.1f env-steps/s incl. host lists (%.2f ms per batch)" % (lanes, sum(len(e) for e in inter) / dt, dt * 1e3 / 8))
    del inter
ro.close()
