"""The generic text-policy path (GPT2PPOPolicy.act + interact_environment, any env / tokenizer) with and without K/V reuse across act() calls:
Wordle through the TEXT protocol (histories grow by an action and an observation per turn), GPT-2-small random init, byte-level stand-in
tokenizer, B envs, 6 turns."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lmrl_gym_amd
from lmrl_gym_amd import _lib, datasets as DS, environment as E
from lmrl_gym_amd.envs import wordle as W
from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine
from lmrl_gym_amd.policies import GPT2PPOPolicy
dev = _lib.require_gpu()
tok = DS.ByteTokenizer()
eng = GPT2Engine.random_init(GPT2Config.gpt2_small(), seed=0, device=dev)
vocab = W.Vocabulary.builtin("wordle_official_400.txt")
for B in (64, 256):
    for reuse in (False, True):
        pol = GPT2PPOPolicy(eng, tok, max_input_length=256, max_new_tokens=12, do_sample=True, seed=1, eos_token_id=tok.eos_token_id,
                            out_str_process=lambda x: x.removesuffix("\n") + "\n", reuse_kv=reuse)
        env = W.ReformatWordleEnvironment(W.WordleEnvironment(vocab, require_words_in_vocab=False))
        E.interact_environment(env, pol, env_seed=list(range(B)), bsize=B)          # warm-up
        n0 = pol._gen.prefilled_tokens
        torch.cuda.synchronize(); t0 = time.perf_counter()
        inter = E.interact_environment(env, pol, env_seed=list(range(100, 100 + B)), bsize=B)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        steps = sum(len(ep) for ep in inter)
        print("B=%4d reuse_kv=%-5s: %6d env steps in %.2f s -> %.0f env-steps/s, %d prompt tokens forwarded" %
              (B, reuse, steps, dt, steps / dt, pol._gen.prefilled_tokens - n0), flush=True)
