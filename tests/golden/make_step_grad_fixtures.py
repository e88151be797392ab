"""Generate tests/golden/rl_step_grads.json: DIRECTIONAL DERIVATIVES of the reference's own train-step closures, by complex-step differentiation
THROUGH THE REFERENCE'S CODE ("gradients are pinned by float64 autograd of the restatement, not by the reference").

    python tests/golden/make_step_grad_fixtures.py          (build container only: reads /root/reference)

Same set-up as make_step_fixtures.py — `GPT2ILQLTrain._step`, `GPT2PPOTrain._step`, `GPT2MCTrain._step` imported from /root/reference and executed
unmodified under the numpy `jax.numpy` / `jax.lax` / `optax` / `flax.linen` stand-ins — but in the shim's DERIVATIVE MODE (`_jnp_shim.complex_step`):
every array is complex128, every stand-in is the analytic continuation of its real version (comparisons, max, relu, clip and `where` decide on the real
part), and `jax.lax.stop_gradient` DROPS THE IMAGINARY PART.  With the trainable parameters set to theta + i h v (h = 1e-30) the closure's loss comes
back as L(theta) + i h <dL/dtheta, v>: the directional derivative of the reference's own computation INCLUDING its stop_gradient placement (TD targets,
`detach_*`, the frozen target networks, whose parameters stay real), exact to rounding.  The transformer slot (JaxSeq / HF-Flax GPT-2, third party, absent)
is filled with a complex-capable numpy restatement of oracle/gpt2.py::forward, checked here against that oracle on the real parts.

Per case: the loss (== rl_steps.json) and <grad, v> for the seeded directions of step_cases.direction over ALL trainable parameters (transformer + heads).
tests/test_oracle_steps_pinned.py holds the float64 autograd of the restatement to them, tests/test_gpu_train_steps_pinned.py the device gradients.
"""
from __future__ import annotations

import json
import math
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE); sys.path.insert(0, ROOT)
import _ref_import  # noqa: E402
import _flax_shim  # noqa: E402

_ref_import.install(jnp_shim=True, extra=_flax_shim.make_modules())
import _jnp_shim as S  # noqa: E402
import step_cases as C  # noqa: E402
import jax  # noqa: E402
import jax.experimental.pjit as _pjit_mod  # noqa: E402
import JaxSeq.utils as _jsu  # noqa: E402

_pjit_mod.pjit = lambda fun=None, **kw: fun
_jsu.with_named_sharding_constraint = lambda x, mesh, ps: x
_jsu.match_partition_rules = lambda rules, tree: types.SimpleNamespace(params=None)


def _value_and_grad(fn, has_aux=False, argnums=0):
    def wrapped(*args):
        out = fn(*args)
        return out, tuple(None for _ in (argnums if isinstance(argnums, tuple) else (argnums,)))
    return wrapped


jax.value_and_grad = _value_and_grad
jax.device_get = lambda x: x

import torch  # noqa: E402
from functools import partial  # noqa: E402
from oracle import gpt2 as OG  # noqa: E402
from LLM_RL.algorithms.ilql.gpt2.interface import GPT2ILQLTrain  # noqa: E402
from LLM_RL.algorithms.ilql.base_interface import ilql_loss  # noqa: E402
from LLM_RL.heads.mlp_head import MLPHead, MLPHeadConfig  # noqa: E402
from LLM_RL.heads.linear_head import LinearHead, LinearHeadConfig  # noqa: E402
from LLM_RL.algorithms.ppo.gpt2.interface import GPT2PPOTrain  # noqa: E402
from LLM_RL.algorithms.ppo.base_interface import ppo_loss_fn  # noqa: E402
from LLM_RL.algorithms.mc_returns.gpt2.interface import GPT2MCTrain  # noqa: E402
from LLM_RL.algorithms.mc_returns.base_interface import mc_loss  # noqa: E402

H = 1e-30


def gpt2_forward_complex(sd, ids, n_head, am, pos, eps=1e-5):
    """oracle/gpt2.py::forward on complex128 numpy arrays (analytic continuation: the softmax shift uses the real part, masks are real)."""
    ids, am, pos = np.asarray(ids), np.asarray(am).astype(bool), np.asarray(pos)
    B, T = ids.shape
    x = sd["wte.weight"][ids] + sd["wpe.weight"][pos]
    d = x.shape[-1]
    hd = d // n_head

    def ln(v, g, b):
        mu = v.mean(-1, keepdims=True)
        var = ((v - mu) ** 2).mean(-1, keepdims=True)
        return (v - mu) / np.sqrt(var + eps) * g + b

    def gelu_new(v):
        return 0.5 * v * (1.0 + np.tanh(math.sqrt(2.0 / math.pi) * (v + 0.044715 * v ** 3)))
    causal = np.tril(np.ones((T, T), dtype=bool))
    n_layer = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("h."))
    for l in range(n_layer):
        p = f"h.{l}."
        h = ln(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"])
        qkv = h @ sd[p + "attn.c_attn.weight"] + sd[p + "attn.c_attn.bias"]
        q, k, v = (t.reshape(B, T, n_head, hd).transpose(0, 2, 1, 3) for t in np.split(qkv, 3, axis=-1))
        att = (q @ k.transpose(0, 1, 3, 2)) / math.sqrt(hd)
        mask = causal[None, None] & am[:, None, None, :]
        att = np.where(mask, att, -1e300)
        att = att - att.real.max(-1, keepdims=True)
        e = np.where(mask, np.exp(att), 0.0)
        att = e / e.sum(-1, keepdims=True)
        a = (att @ v).transpose(0, 2, 1, 3).reshape(B, T, d)
        x = x + a @ sd[p + "attn.c_proj.weight"] + sd[p + "attn.c_proj.bias"]
        h = ln(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"])
        x = x + gelu_new(h @ sd[p + "mlp.c_fc.weight"] + sd[p + "mlp.c_fc.bias"]) @ sd[p + "mlp.c_proj.weight"] + sd[p + "mlp.c_proj.bias"]
    hid = ln(x, sd["ln_f.weight"], sd["ln_f.bias"])
    return hid @ sd["wte.weight"].T, hid


class FakeGPT2:
    config = types.SimpleNamespace(mesh="mesh", get_partition_rules=lambda: [])

    def __call__(self, input_ids, attention_mask=None, position_ids=None, params=None, dropout_rng=None, train=True, output_hidden_states=True,
                 past_key_values=None):
        am_ = np.asarray(attention_mask).real.astype(np.int64)
        if position_ids is None:
            position_ids = np.maximum(np.cumsum(am_, axis=1) - 1, 0)
        sd = {k: np.asarray(v, dtype=np.complex128) for k, v in params.items()}
        lg, hid = gpt2_forward_complex(sd, np.asarray(input_ids).real.astype(np.int64), C.CFG["n_head"], am_, np.asarray(position_ids).real.astype(np.int64))
        return types.SimpleNamespace(hidden_states=(None, S.asarray(hid)), logits=S.asarray(lg), past_key_values=None)


class FakeTrainState:
    def __init__(self, params, step, mini_step):
        self.params, self.step = params, step
        self.opt_state = types.SimpleNamespace() if mini_step is None else types.SimpleNamespace(mini_step=mini_step)

    def apply_gradients(self, grads):
        return FakeTrainState(self.params, self.step + 1, getattr(self.opt_state, "mini_step", None))


def nest(flat):
    out = {}
    for k, v in flat.items():
        a, b = k.split(".")
        out.setdefault(a, {})[b] = v
    return out


def cplx(flat, v):
    return {k: np.asarray(flat[k], dtype=np.complex128) + 1j * H * v[k] for k in flat}


def check_forward():
    sd = C.state_dict(3)
    b = C.ilql_batch(1)
    lg, hid = gpt2_forward_complex({k: np.asarray(v, dtype=np.complex128) for k, v in sd.items()}, b["input_ids"], C.CFG["n_head"], b["attention_mask"], b["position_ids"])
    lg0, hid0 = OG.forward({k: torch.from_numpy(v) for k, v in sd.items()}, torch.from_numpy(b["input_ids"]).long(), C.CFG["n_head"],
                           attention_mask=torch.from_numpy(b["attention_mask"]), position_ids=torch.from_numpy(b["position_ids"]).long(), return_hidden=True)
    live = b["attention_mask"].astype(bool)
    assert np.abs(hid.real - hid0.numpy())[live].max() < 1e-10 and np.abs(lg.real - lg0.numpy())[live].max() < 1e-9
    print("complex GPT-2 forward == oracle/gpt2.py on the real parts")


def main():
    check_forward()
    S.complex_step(True)
    V, d = C.CFG["vocab"], C.CFG["d_model"]
    ref_losses = json.load(open(os.path.join(HERE, "rl_steps.json")))
    out = {"_meta": "directional derivatives <dL/dtheta, v> of the reference's _step closures by complex-step differentiation through the reference code "
                    "(stop_gradient drops the imaginary part); directions: step_cases.direction(seed, flat params), keys 'base.*', 'q1.*', 'q2.*', 'v.*' / 'head.*'",
           "direction_seeds": list(C.GRAD_DIRECTION_SEEDS)}
    A = S.asarray
    for case in C.ILQL_CASES:
        sd, tsd = C.state_dict(10 + case["seed"]), C.state_dict(20 + case["seed"])
        heads = {n: C.flat_head(C.mlp_head(s + case["seed"], o)) for n, s, o in (("q1", 30, V), ("q2", 40, V), ("v", 50, 1))}
        tq1, tq2 = C.mlp_head(60 + case["seed"], V), C.mlp_head(70 + case["seed"], V)
        flat = {"base." + k: v for k, v in sd.items()}
        for n, h in heads.items():
            flat.update({f"{n}.{k}": v for k, v in h.items()})
        q_model = MLPHead(MLPHeadConfig(input_dim=d, hidden_dim=d, output_dim=V, mesh="mesh"))
        v_model = MLPHead(MLPHeadConfig(input_dim=d, hidden_dim=d, output_dim=1, mesh="mesh"))
        b = C.ilql_batch(case["seed"])
        nxt = [A(b[k]) for k in ("next_token_ids", "next_tokens_attention_mask", "next_tokens_position_ids", "next_dones")] if case["use_next"] else [None] * 4
        derivs, loss0 = [], None
        for dseed in C.GRAD_DIRECTION_SEEDS:
            v = C.direction(dseed, flat)
            th = cplx(flat, v)
            part = lambda pre: {k[len(pre):]: x for k, x in th.items() if k.startswith(pre)}
            ms, st0 = case["mini_step"], case["step0"]
            train = GPT2ILQLTrain.load_train(
                base_train_state=FakeTrainState(part("base."), st0, ms), target_base_params=tsd if case["target_base"] else None,
                q1_head_train_state=FakeTrainState(nest(part("q1.")), st0, ms), q2_head_train_state=FakeTrainState(nest(part("q2.")), st0, ms),
                v_head_train_state=FakeTrainState(nest(part("v.")), st0, ms), q1_target_head_params=tq1, q2_target_head_params=tq2, base_model=FakeGPT2(),
                q_head_model=q_model, v_head_model=v_model, tokenizer=None, loss_fn=partial(ilql_loss, **C.LOSS_KW), detach_q1=False, detach_q2=False,
                detach_v=False, polyak_alpha=case["polyak_alpha"], hard_update_every=case["hard_update_every"])
            res = train._step(train.base_train_state, train.target_base_params, train.q1_head_train_state, train.q2_head_train_state, train.v_head_train_state,
                              train.q1_target_head_params, train.q2_target_head_params, A(b["input_ids"]), A(b["attention_mask"]), A(b["position_ids"]),
                              A(b["should_take_action"]), A(b["rewards"]), A(b["dones"]), *nxt, None, True)
            loss = complex(np.asarray(res[-2]))
            loss0 = loss.real
            derivs.append(loss.imag / H)
        assert abs(loss0 - ref_losses[case["name"]]["loss"]) <= 2e-5 * abs(loss0), (case["name"], loss0, ref_losses[case["name"]]["loss"])
        out[case["name"]] = dict(loss=loss0, ddir=derivs)
        print(case["name"], loss0, derivs)
    for case in C.PPO_CASES[:1]:
        sd = C.state_dict(80 + case["seed"])
        vh = C.flat_head(C.linear_head(90 + case["seed"]))
        flat = {"base." + k: v for k, v in sd.items()}
        flat.update({"head." + k: v for k, v in vh.items()})
        v_model = LinearHead(LinearHeadConfig(input_dim=d, output_dim=1, mesh="mesh"))
        b = C.ppo_batch(case["seed"])
        derivs = []
        for dseed in C.GRAD_DIRECTION_SEEDS:
            th = cplx(flat, C.direction(dseed, flat))
            part = lambda pre: {k[len(pre):]: x for k, x in th.items() if k.startswith(pre)}
            train = GPT2PPOTrain.load_train(policy_train_state=FakeTrainState(part("base."), 0, None), value_head_train_state=FakeTrainState(nest(part("head.")), 0, None),
                                            policy_model=FakeGPT2(), value_head_model=v_model, tokenizer=None, loss_fn=partial(ppo_loss_fn, **C.PPO_KW),
                                            bc_loss_fn=None, bc_loss_weight=0.0)
            res = train._step(train.policy_train_state, train.value_head_train_state, A(b["input_ids"]), A(b["attention_mask"]), A(b["position_ids"]),
                              A(b["should_take_action"]), A(b["old_logprobs"]), A(b["old_values"]), A(b["old_advantages"]), A(b["old_returns"]), None,
                              None, None, None, None, True)
            loss = complex(np.asarray(res[-2]))
            derivs.append(loss.imag / H)
        assert abs(loss.real - ref_losses[case["name"]]["loss"]) <= 2e-5 * abs(loss.real)
        out[case["name"]] = dict(loss=loss.real, ddir=derivs)
        print(case["name"], loss.real, derivs)
    case = C.MC_CASE
    sd, qh = C.state_dict(110 + case["seed"]), C.flat_head(C.mlp_head(120 + case["seed"], V))
    flat = {"base." + k: v for k, v in sd.items()}
    flat.update({"head." + k: v for k, v in qh.items()})
    q_model = MLPHead(MLPHeadConfig(input_dim=d, hidden_dim=d, output_dim=V, mesh="mesh"))
    b = C.mc_batch(case["seed"])
    derivs = []
    for dseed in C.GRAD_DIRECTION_SEEDS:
        th = cplx(flat, C.direction(dseed, flat))
        part = lambda pre: {k[len(pre):]: x for k, x in th.items() if k.startswith(pre)}
        train = GPT2MCTrain.load_train(base_train_state=FakeTrainState(part("base."), 0, None), q_head_train_state=FakeTrainState(nest(part("head.")), 0, None),
                                       base_model=FakeGPT2(), q_head_model=q_model, tokenizer=None, loss_fn=partial(mc_loss, cql_weight=case["cql_weight"]), detach_q=False)
        res = train._step(train.base_train_state, train.q_head_train_state, A(b["input_ids"]), A(b["attention_mask"]), A(b["position_ids"]),
                          A(b["should_take_action"]), A(b["returns"]), None, True)
        loss = complex(np.asarray(res[-2]))
        derivs.append(loss.imag / H)
    assert abs(loss.real - ref_losses[case["name"]]["loss"]) <= 2e-5 * abs(loss.real)
    out[case["name"]] = dict(loss=loss.real, ddir=derivs)
    print(case["name"], loss.real, derivs)
    S.complex_step(False)
    path = os.path.join(HERE, "rl_step_grads.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
