"""Inputs of tests/golden/rl_steps.json, regenerated from seeds by the generator (make_step_fixtures.py) and by the tests: a tiny GPT-2
(2 layers, d = 128, 2 heads of 64 — a shape every engine of the package runs), MLP / linear heads, and ILQL / PPO batches with right padding,
both `dones` values, and a next-token chunk.  numpy RandomState only (bit-reproducible across machines)."""
from __future__ import annotations

import numpy as np

CFG = dict(n_layer=2, n_head=2, d_model=128, d_ff=256, vocab=97, n_pos=48)
PAD = 96
B, T, TN = 5, 20, 6


def state_dict(seed: int):
    """HF-named GPT-2 parameters as float32 numpy arrays (Conv1D kernels [in, out])."""
    c, r = CFG, np.random.RandomState(seed)
    d, f = c["d_model"], c["d_ff"]
    n = lambda *s, sc=0.08: (r.randn(*s) * sc).astype(np.float32)
    sd = {"wte.weight": n(c["vocab"], d, sc=0.3), "wpe.weight": n(c["n_pos"], d, sc=0.1),
          "ln_f.weight": (1 + n(d, sc=0.1)), "ln_f.bias": n(d, sc=0.1)}
    for l in range(c["n_layer"]):
        p = f"h.{l}."
        sd[p + "ln_1.weight"] = 1 + n(d, sc=0.1); sd[p + "ln_1.bias"] = n(d, sc=0.1)
        sd[p + "attn.c_attn.weight"] = n(d, 3 * d); sd[p + "attn.c_attn.bias"] = n(3 * d, sc=0.05)
        sd[p + "attn.c_proj.weight"] = n(d, d); sd[p + "attn.c_proj.bias"] = n(d, sc=0.05)
        sd[p + "ln_2.weight"] = 1 + n(d, sc=0.1); sd[p + "ln_2.bias"] = n(d, sc=0.1)
        sd[p + "mlp.c_fc.weight"] = n(d, f); sd[p + "mlp.c_fc.bias"] = n(f, sc=0.05)
        sd[p + "mlp.c_proj.weight"] = n(f, d); sd[p + "mlp.c_proj.bias"] = n(d, sc=0.05)
    return sd


def mlp_head(seed: int, out: int):
    r, d = np.random.RandomState(seed), CFG["d_model"]
    return {"dense1": {"kernel": (r.randn(d, d) * 0.1).astype(np.float32), "bias": (r.randn(d) * 0.1).astype(np.float32)},
            "dense2": {"kernel": (r.randn(d, out) * 0.1).astype(np.float32), "bias": np.full(out, -0.4, np.float32) + (r.randn(out) * 0.05).astype(np.float32)}}


def linear_head(seed: int, out: int = 1):
    r, d = np.random.RandomState(seed), CFG["d_model"]
    return {"dense": {"kernel": (r.randn(d, out) * 0.1).astype(np.float32), "bias": np.full(out, -0.3, np.float32)}}


def flat_head(h):
    """{'dense1': {'kernel': ..}} -> {'dense1.kernel': ..} (the package's head parameter naming)."""
    return {f"{k}.{kk}": v for k, sub in h.items() for kk, v in sub.items()}


def ilql_batch(seed: int):
    r = np.random.RandomState(seed)
    ids = r.randint(0, PAD, size=(B, T)).astype(np.int32)
    lens = np.array([T, T - 3, T, T - 7, T - 1])
    for b in range(B):
        ids[b, lens[b]:] = PAD
    t = np.arange(T - 1)
    sta = np.broadcast_to(((t >= 2) & (((t - 2) // 3) % 2 == 0))[None, :], (B, T - 1)).copy()
    sta &= t[None, :] < (lens[:, None] - 1)
    rewards = (r.randn(B, T - 1) * sta).astype(np.float32)
    dones = np.array([1, 0, 0, 1, 0], dtype=np.float32)
    nids = r.randint(0, PAD, size=(B, TN)).astype(np.int32)
    nlens = np.array([TN, 2, TN - 1, 1, 4])
    for b in range(B):
        nids[b, nlens[b]:] = PAD
    ndones = np.array([0, 1, 0, 0, 1], dtype=np.float32)
    am = (ids != PAD).astype(np.int32)
    pos = np.maximum(np.cumsum(am, axis=1) - 1, 0).astype(np.int32)
    nam = (nids != PAD).astype(np.int32)
    npos = np.maximum(np.cumsum(nam, axis=1) - 1, 0).astype(np.int32)
    return dict(input_ids=ids, attention_mask=am, position_ids=pos, should_take_action=sta, rewards=rewards, dones=dones,
                next_token_ids=nids, next_tokens_attention_mask=nam, next_tokens_position_ids=npos, next_dones=ndones)


ILQL_CASES = [
    dict(name="ilql_in_sequence_v_final", seed=1, use_next=False, target_base=True, polyak_alpha=0.005, hard_update_every=None, mini_step=None, step0=4),
    dict(name="ilql_next_token_v_final", seed=2, use_next=True, target_base=False, polyak_alpha=0.1, hard_update_every=None, mini_step=0, step0=7),
    dict(name="ilql_hard_update_on_period", seed=3, use_next=False, target_base=True, polyak_alpha=0.005, hard_update_every=4, mini_step=0, step0=7),
    dict(name="ilql_accumulating_micro_step", seed=4, use_next=True, target_base=True, polyak_alpha=0.5, hard_update_every=2, mini_step=1, step0=3),
]
LOSS_KW = dict(gamma=0.99, tau=0.7, cql_weight=0.01)


def ppo_batch(seed: int):
    r = np.random.RandomState(100 + seed)
    b = ilql_batch(seed)
    f = lambda sc: (r.randn(B, T - 1) * sc).astype(np.float32)
    return dict(input_ids=b["input_ids"], attention_mask=b["attention_mask"], position_ids=b["position_ids"], should_take_action=b["should_take_action"],
                old_logprobs=f(0.3) - 4.5, old_values=f(1.0), old_advantages=f(1.0), old_returns=f(1.0))


PPO_CASES = [dict(name="ppo_plain", seed=5, bc_weight=None), dict(name="ppo_with_bc_term", seed=6, bc_weight=0.7)]
PPO_KW = dict(cliprange_value=0.2, cliprange=0.2, value_loss_coef=1.0)


def mc_batch(seed: int):
    b = ilql_batch(seed)
    r = np.random.RandomState(200 + seed)
    return dict(input_ids=b["input_ids"], attention_mask=b["attention_mask"], position_ids=b["position_ids"], should_take_action=b["should_take_action"],
                returns=(r.randn(B, T - 1) * b["should_take_action"]).astype(np.float32))


MC_CASE = dict(name="mc_step", seed=7, cql_weight=0.05)
VALUE_RL_CASES = [dict(name="value_rl_logits_pi_beta_two_heads", seed=8, pi_beta=True, q2=True, beta=32.0),
                  dict(name="value_rl_logits_value_only_one_head", seed=9, pi_beta=False, q2=False, beta=8.0)]


class CharTok:
    """One token per character (ids < PAD), pad = PAD: the tokenizer handed to the score functions."""
    pad_token_id = PAD

    def encode(self, s):
        return [ord(c) % PAD for c in s]


def score_histories():
    """[(text, is_action), ...] per candidate: last item is the action; different prefix / action lengths, one longer than max_length."""
    return [[("maze: go\n", False), ("move left\n", True)],
            [("maze: go\n", False), ("move right\n", True)],
            [("a\n", False), ("up\n", True), ("wall\n", False), ("down\n", True)],
            [("x" * 30 + "\n", False), ("move up\n", True)],          # 39 tokens > max_length: the reference keeps the LAST max_length tokens
            [("q\n", False), ("z\n", True)]]


SCORE_MAX_LENGTH, SCORE_BSIZE = 28, 2
SCORE_CASE = dict(name="score_fns", seed=11, value_weight=1.5, logit_weight=0.25)


def ppo_chains(seed: int):
    """Token trajectory chains as plain dicts: [[{tokens, is_action, reward, done}, ...chunks], ...chains].  Later chunks start with a state
    (the reference asserts it, ppo/base_interface.py:319-327); rewards sit on action tokens; only a chain's last chunk may be done."""
    r = np.random.RandomState(300 + seed)
    spec = [[(9, True)], [(8, False), (6, True)], [(5, False), (7, False), (4, False)], [(6, False)]]
    chains = []
    for ch in spec:
        chunks = []
        for n, done in ch:
            tokens = r.randint(0, PAD, size=n).astype(np.int32)
            is_action = np.zeros(n, dtype=bool)
            is_action[2:] = (np.arange(n - 2) % 3) != 2          # runs of 2 action tokens separated by a state token, after a 2-token header
            if not is_action.any():
                is_action[-1] = True
            reward = (r.randn(n) * is_action).astype(np.float32)
            chunks.append(dict(tokens=tokens, is_action=is_action, reward=reward, done=bool(done)))
        chains.append(chunks)
    return chains


PPO_DATA_CASE = dict(name="ppo_data_pipeline", seed=12, bsize=3, gamma=0.97, lam=0.9, kl_weight=0.05)


def perturbed(sd, seed: int, eps: float = 0.03):
    """A nearby parameter set (the policy a few updates after `sd`): every tensor + eps * std * noise."""
    r = np.random.RandomState(seed)
    return {k: (v + eps * max(float(v.std()), 1e-3) * r.randn(*v.shape)).astype(np.float32) for k, v in sd.items()}


def direction(seed: int, params: dict) -> dict:
    """A seeded direction in parameter space for the directional-derivative fixtures (rl_step_grads.json): per tensor, unit-variance noise scaled by the
    tensor's RMS (0.05 where the tensor is ~0), walked in sorted key order.  `params`: FLAT {name: array}."""
    r = np.random.RandomState(seed)
    out = {}
    for k in sorted(params):
        v = np.asarray(params[k], dtype=np.float64)
        rms = float(np.sqrt((v * v).mean()))
        out[k] = r.randn(*v.shape) * (rms if rms > 1e-3 else 0.05)
    return out


GRAD_DIRECTION_SEEDS = (1001, 1002, 1003)
