"""Seeded inputs of the loss-function golden vectors (numpy only) — shared by the generator (make_loss_fixtures.py, which
feeds them to the REFERENCE's functions) and by the tests (which feed the same arrays to oracle/rl.py and to the HIP kernels).
`numpy.random.RandomState` streams are frozen across numpy versions, so only seeds/shapes are stored in the JSON."""
from __future__ import annotations

import numpy as np


def grid(rng, B, T1, p_act=0.5, pad_tail=True):
    """should_take_action [B, T1] bool, attention_mask [B, T1] int32 (right padding on some rows, one row without any action)."""
    sta = rng.rand(B, T1) < p_act
    attn = np.ones((B, T1), dtype=np.int32)
    if pad_tail:
        for b in range(1, B, 2):
            cut = rng.randint(T1 // 2, T1)
            attn[b, cut:] = 0
    if B > 2:
        sta[2, :] = False
    sta[0, 0] = True
    return sta, attn


PPO_CASES = [dict(seed=1, B=4, T1=23, cliprange_value=0.2, cliprange=0.2, value_loss_coef=1.0),
             dict(seed=2, B=3, T1=40, cliprange_value=0.1, cliprange=0.3, value_loss_coef=0.5),
             dict(seed=3, B=6, T1=9, cliprange_value=0.2, cliprange=0.2, value_loss_coef=1.0)]


def ppo_inputs(c):
    rng = np.random.RandomState(c["seed"])
    B, T1 = c["B"], c["T1"]
    sta, attn = grid(rng, B, T1)
    f = lambda s=1.0: (rng.randn(B, T1) * s).astype(np.float32)
    logprobs = f() - 2.0
    old_logprobs = (logprobs + f(0.3)).astype(np.float32)
    values, old_values, old_adv, old_ret = f(), f(), f(), f()
    old_values = (values + f(0.3)).astype(np.float32)
    return dict(attention_mask=attn, logprobs=logprobs, values=values, should_take_action=sta, old_logprobs=old_logprobs,
                old_values=old_values, old_advantages=old_adv, old_returns=old_ret)


PPO_DIFF = ("logprobs", "values")                 # differentiable inputs (the others are data)

ILQL_CASES = [dict(seed=11, B=4, T1=21, V=37, gamma=0.99, tau=0.7, cql_weight=0.01),
              dict(seed=12, B=3, T1=33, V=19, gamma=0.9, tau=0.6, cql_weight=0.1),
              dict(seed=13, B=5, T1=8, V=50, gamma=1.0, tau=0.5, cql_weight=1.0)]


def ilql_inputs(c):
    rng = np.random.RandomState(c["seed"])
    B, T1, V = c["B"], c["T1"], c["V"]
    sta, attn = grid(rng, B, T1)
    f = lambda *s: rng.randn(*s).astype(np.float32)
    d = dict(q1=f(B, T1), q2=f(B, T1), v=f(B, T1), v_final=f(B), target_q1=f(B, T1), target_q2=f(B, T1),
             q1_logits=f(B, T1, V), q2_logits=f(B, T1, V), token_ids=rng.randint(0, V, size=(B, T1)).astype(np.int32),
             attention_mask=attn, should_take_action=sta, rewards=(f(B, T1) * sta).astype(np.float32))
    return d


ILQL_DIFF = ("q1", "q2", "v", "v_final", "target_q1", "target_q2", "q1_logits", "q2_logits", "rewards")

MC_CASES = [dict(seed=21, B=4, T1=17, V=33, cql_weight=0.05), dict(seed=22, B=3, T1=30, V=12, cql_weight=1.0)]


def mc_inputs(c):
    rng = np.random.RandomState(c["seed"])
    B, T1, V = c["B"], c["T1"], c["V"]
    sta, attn = grid(rng, B, T1)
    f = lambda *s: rng.randn(*s).astype(np.float32)
    return dict(q=f(B, T1), q_logits=f(B, T1, V), token_ids=rng.randint(0, V, size=(B, T1)).astype(np.int32), attention_mask=attn,
                should_take_action=sta, returns=(f(B, T1) * sta).astype(np.float32))


MC_DIFF = ("q", "q_logits", "returns")

BC_CASES = [dict(seed=31, B=3, T=14, V=29, non_action_weight=0.3), dict(seed=32, B=4, T=20, V=11, non_action_weight=1.0),
            dict(seed=33, B=2, T=9, V=40, non_action_weight=0.0)]


def bc_inputs(c):
    rng = np.random.RandomState(c["seed"])
    B, T, V = c["B"], c["T"], c["V"]
    is_action, attn = grid(rng, B, T)
    return dict(logits=rng.randn(B, T, V).astype(np.float32), input_ids=rng.randint(0, V, size=(B, T)).astype(np.int32),
                attention_mask=attn, is_action=is_action.astype(np.int32))


WHITEN_CASES = [dict(seed=41, n=57, shift_mean=True, scale=3.0, offset=1.5), dict(seed=42, n=200, shift_mean=False, scale=0.01, offset=-4.0),
                dict(seed=43, n=2, shift_mean=True, scale=1.0, offset=0.0)]


def whiten_input(c):
    rng = np.random.RandomState(c["seed"])
    return (rng.randn(c["n"]) * c["scale"] + c["offset"]).astype(np.float32)


RTG_CASES = [dict(seed=51, n=29, gamma=1.0), dict(seed=52, n=29, gamma=0.99), dict(seed=53, n=12, gamma=0.7), dict(seed=54, n=1, gamma=0.9)]


def rtg_input(c):
    return np.random.RandomState(c["seed"]).randn(c["n"]).astype(np.float32)


LOGPROB_CASES = [dict(seed=61, B=3, T=12, V=41)]


def logprob_inputs(c):
    rng = np.random.RandomState(c["seed"])
    return dict(logits=(rng.randn(c["B"], c["T"], c["V"]) * 2).astype(np.float32),
                input_ids=rng.randint(0, c["V"], size=(c["B"], c["T"])).astype(np.int32))


def direction(seed, k, name, shape):
    """k-th random direction for input `name` (unit-scale Gaussian)."""
    h = (hash_name(name) + 1000003 * k + 7919 * seed) % (2 ** 31 - 1)
    return np.random.RandomState(h).randn(*shape)


def hash_name(name: str) -> int:
    v = 0
    for ch in name:
        v = (v * 131 + ord(ch)) % 1000000007
    return v


N_DIRECTIONS = 3
