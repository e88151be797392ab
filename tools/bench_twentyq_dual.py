"""configs[4] at size on ONE GPU: Twenty Questions with a GPT-2-large oracle model and a GPT-2-medium guesser, both resident on the HIP
engine, 1024 lock-step envs through the reference protocol (`interact_environment` over `BatchedTwentyQuestionsPolicyEnvironment`,
llm_rl_scripts/twenty_questions/env/env.py:66-141).  Random-init weights and a byte-level stand-in tokenizer (no GPT-2 BPE files offline):
what is measured is the dual-model loop — per turn one guesser generation (<= `--q-tokens` tokens on the whole history, KV reuse across
turns) and one oracle generation (reference prompt, <= 4 greedy tokens) — not answer quality.

    python tools/bench_twentyq_dual.py [--envs 1024] [--turns 4] [--q-tokens 16]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import lmrl_gym_amd  # noqa: E402,F401
from lmrl_gym_amd import _lib, environment as E  # noqa: E402
from lmrl_gym_amd.envs import twenty_questions as Q  # noqa: E402
from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine  # noqa: E402
from lmrl_gym_amd.policies import GPT2PPOPolicy  # noqa: E402


class ByteTok:
    """One token per byte (the models keep GPT-2's 50 257-row tables).  A random-init model samples ids all over the table: ids >= 256 decode to a
    printable character (32 + id % 95) so that a generated question has as many characters as tokens — with ids >= 256 dropped, every question was
    "?" and every oracle prompt equal to the previous one (K/V reuse then forwards 4 tokens per call: a degenerate workload)."""
    pad_token_id, eos_token_id = 50256, 10

    def encode(self, s):
        return list(s.encode("latin-1", errors="replace"))

    def decode(self, ids, skip_special_tokens=True):
        return bytes((int(i) if int(i) < 256 else 32 + int(i) % 95) for i in ids if int(i) != self.pad_token_id).decode("latin-1")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--turns", type=int, default=4)
    ap.add_argument("--q-tokens", type=int, default=16)
    a = ap.parse_args()
    dev = _lib.require_gpu()
    tok = ByteTok()
    guesser = GPT2Engine.random_init(GPT2Config.gpt2_medium(), seed=1, device=dev)
    oracle_eng = GPT2Engine.random_init(GPT2Config.gpt2_large(), seed=2, device=dev)
    Q.set_pos_tagger(Q.rule_pos_tag)
    wl = Q.get_default_word_list()
    oracle = Q.GPT2EngineOracle(oracle_eng, tok, max_input_length=160, max_new_tokens=4, eos_token_id=10)
    asker = GPT2PPOPolicy(guesser, tok, max_input_length=64 + a.turns * (a.q_tokens + 8), max_new_tokens=a.q_tokens, do_sample=True, temperature=1.0,
                          seed=3, eos_token_id=10, out_str_process=Q.asker_postproc_filter_repeats)
    env = Q.BatchedTwentyQuestionsPolicyEnvironment(oracle, wl, max_conversation_length=a.turns, bsize=a.envs)
    n_gen = {"guesser": 0, "oracle": 0}                     # tokens the engines actually generated (counted where the policies decode them)
    for name, pol in (("guesser", asker), ("oracle", oracle._policy)):
        def counted(ids, _dec=pol._decode_generation, _name=name):
            n_gen[_name] += len(ids)
            return _dec(ids)
        pol._decode_generation = counted
    for rep in range(2):                       # first pass: allocation / first-touch
        for pol in (asker, oracle._policy):
            if pol._gen is not None:
                pol._gen.prefilled_tokens = 0
        n_gen.update(guesser=0, oracle=0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        inter = E.interact_environment(env, asker, env_seed=list(range(a.envs)), env_options=[{"deterministic": True}] * a.envs, bsize=a.envs)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    n = sum(len(ep) for ep in inter)
    print(f"Twenty Questions dual-model rollout: GPT-2-medium guesser + GPT-2-large oracle, {a.envs} envs x {a.turns} turns, <= {a.q_tokens} question tokens: "
          f"{n / dt:.1f} env-steps/s ({dt * 1e3 / a.turns:.0f} ms per lock-step turn, {n} env steps in {dt:.2f} s; text protocol, host tokenisation)")
    # what the loop computes: tokens actually forwarded by the two models (prefill counters of the generators + one decode forward per generated
    # token but the last), priced at 2 x (matmul parameters) flops per token — the loop is GPU-bound (rocprofv3: kernel time == wall time,
    # profiles/r04_twentyq_kernel_stats.csv), so its rate is set by these flops, not by the host text protocol
    def mm_params(cfg):
        return cfg.n_layer * (4 * cfg.d_model ** 2 + 2 * cfg.d_model * cfg.d_ff)

    def head_params(cfg):
        return cfg.vocab_padded * cfg.d_model                                      # the tied LM head runs once per SAMPLED token only
    g_pre, o_pre = asker._gen.prefilled_tokens, oracle._policy._gen.prefilled_tokens
    g_dec, o_dec = max(n_gen["guesser"] - n, 0), max(n_gen["oracle"] - n, 0)        # decode forwards: one per generated token but the last
    fl = 2.0 * (mm_params(guesser.cfg) * (g_pre + g_dec) + mm_params(oracle_eng.cfg) * (o_pre + o_dec) +
                head_params(guesser.cfg) * n_gen["guesser"] + head_params(oracle_eng.cfg) * n_gen["oracle"])
    print(f"tokens forwarded per env-step: guesser {(g_pre + g_dec) / n:.1f} (prefill {g_pre / n:.1f}), oracle {(o_pre + o_dec) / n:.1f} (prefill {o_pre / n:.1f}); "
          f"{fl / 1e12:.1f} TFLOP in {dt:.2f} s = {fl / dt / 1e12:.0f} TFLOP/s = {fl / dt / 2.5e15:.2f} of the dense bf16 MFMA peak; "
          f"at 100 % of that peak this workload would reach {n / (fl / 2.5e15):.0f} env-steps/s (byte-level stand-in tokenizer: ~4x the tokens of GPT-2 BPE)")
    yes = sum(t.post_transition_history[-1].text == "Yes.\n" for ep in inter for t in ep)
    print(f"answers: {yes} Yes / {n - yes} No (random-init oracle)")


if __name__ == "__main__":
    main()
