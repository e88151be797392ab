"""Generate tests/golden/rl_steps.json by EXECUTING the reference's own train-step closures and its value-RL generation call:
`GPT2ILQLTrain._step` (below), `GPT2PPOTrain._step` (ppo/gpt2/interface.py:72-211: values / log-prob wiring, the loss call, the BC combination),
`GPT2MCTrain._step` (mc_returns/gpt2/interface.py:38-160), `GPT2ValueRLGeneration.__call__` (value_rl_base/gpt2/generation.py:36-121), the score
functions (ppo/score_fn.py, ilql/gpt2/score_fn.py) and `PPOInference.get_ppo_data_from_token_trajectory_chain` (ppo/base_interface.py:464-669).

    python tests/golden/make_step_fixtures.py          (build container only: reads /root/reference)

`GPT2ILQLTrain.load_train` (LLM_RL/algorithms/ilql/gpt2/interface.py:18-367) builds `_step`: base / target-base forwards, the Q1 / Q2 / V
and target heads, the Q(s, a) gathers, `v_final` (both branches), the `loss_fn` call, `apply_gradients`, and the Polyak / periodic
target updates gated on `opt_state.mini_step`.  It is pjit-wrapped flax / optax code, so it cannot run as it is here; with
  * `jax.numpy` / `jax.lax` / `optax` backed by the numpy restatements of _jnp_shim.py (unit-tested in tests/test_jnp_shim.py),
  * `flax.linen.Module / Dense / relu` and `flax.struct.PyTreeNode` backed by _flax_shim.py — so the reference's OWN `MLPHead` modules run,
  * `pjit` -> identity decorator, `with_named_sharding_constraint` -> identity, `jax.value_and_grad` -> (value, placeholder gradients),
  * the transformer (JaxSeq / HF-Flax GPT-2, third party, not in the tree) replaced by a callable that returns the float64 oracle's hidden
    states (oracle/gpt2.py) for the given parameters, and `TrainState.apply_gradients` by a stand-in that moves every parameter by a
    fixed factor (so that the target updates have something to average),
the closure body itself — every line between the model calls and the returned tuple — is the reference's unmodified code.  The file stores
loss, the complete log dict and digests of the updated target parameters per case; inputs are regenerated from step_cases.py.
This pins `oracle/rl.py::ilql_gather_qv` / `mlp_head` and `GPT2ILQLTrain.step` / `_update_targets` to the reference's closure.
"""
from __future__ import annotations

import json
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

# stand-ins that must exist BEFORE the reference module is imported (it binds them with `from ... import`)
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

import torch  # noqa: E402
from oracle import gpt2 as OG  # noqa: E402
from LLM_RL.algorithms.ilql.gpt2.interface import GPT2ILQLTrain  # noqa: E402
from LLM_RL.algorithms.ilql.base_interface import ilql_loss  # noqa: E402
from LLM_RL.heads.mlp_head import MLPHead, MLPHeadConfig  # noqa: E402
from LLM_RL.heads.linear_head import LinearHead, LinearHeadConfig  # noqa: E402
from LLM_RL.algorithms.ppo.gpt2.interface import GPT2PPOTrain  # noqa: E402
from LLM_RL.algorithms.ppo.base_interface import ppo_loss_fn  # noqa: E402
from LLM_RL.algorithms.mc_returns.gpt2.interface import GPT2MCTrain  # noqa: E402
from LLM_RL.algorithms.mc_returns.base_interface import mc_loss  # noqa: E402
from LLM_RL.algorithms.value_rl_base.gpt2.generation import GPT2ValueRLGeneration  # noqa: E402
from LLM_RL.algorithms.ppo.score_fn import build_ppo_score_fn, build_bc_score_fn  # noqa: E402
from LLM_RL.algorithms.ilql.gpt2.score_fn import build_ilql_score_fn  # noqa: E402
from LLM_RL.environment import Text  # noqa: E402

jax.device_get = lambda x: x
from functools import partial  # noqa: E402


class FakeGPT2:
    """The transformer slot of the closure: hidden states of the float64 oracle GPT-2 for the parameters it is handed."""
    config = types.SimpleNamespace(mesh="mesh", get_partition_rules=lambda: [])

    def __call__(self, input_ids, attention_mask=None, position_ids=None, params=None, dropout_rng=None, train=True, output_hidden_states=True,
                 past_key_values=None):
        if position_ids is None:
            am_ = np.asarray(attention_mask)
            position_ids = np.maximum(np.cumsum(am_, axis=1) - 1, 0)
        sd = {k: torch.from_numpy(np.asarray(v, dtype=np.float32)) for k, v in params.items()}
        lg, hid = OG.forward(sd, torch.from_numpy(np.asarray(input_ids)).long(), C.CFG["n_head"], attention_mask=torch.from_numpy(np.asarray(attention_mask)),
                             position_ids=torch.from_numpy(np.asarray(position_ids)).long(), return_hidden=True)
        return types.SimpleNamespace(hidden_states=(None, S.asarray(hid.numpy().astype(np.float32))), logits=S.asarray(lg.numpy().astype(np.float32)),
                                     past_key_values=None)


def _scale(tree, f):
    return {k: _scale(v, f) for k, v in tree.items()} if isinstance(tree, dict) else (np.asarray(tree) * np.float32(f)).astype(np.float32)


class FakeTrainState:
    def __init__(self, params, step, mini_step):
        self.params, self.step = params, step
        self.opt_state = types.SimpleNamespace() if mini_step is None else types.SimpleNamespace(mini_step=mini_step)

    def apply_gradients(self, grads):
        ms = getattr(self.opt_state, "mini_step", None)
        return FakeTrainState(_scale(self.params, 0.9), self.step + 1, ms)      # the optimizer's stand-in: every parameter x 0.9


def digest(tree):
    """(sum, sum of squares, first 3 entries) of every leaf, flattened names."""
    out = {}

    def walk(t, prefix):
        if isinstance(t, dict):
            for k, v in t.items():
                walk(v, prefix + k + ".")
        else:
            a = np.asarray(t, dtype=np.float64).ravel()
            out[prefix[:-1]] = [float(a.sum()), float((a * a).sum())] + [float(x) for x in a[:3]]
    walk(tree, "")
    return out


def flat_logs(d, prefix=""):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(flat_logs(v, prefix + k + "."))
        else:
            out[prefix + k] = float(np.asarray(v).real)
    return out


def main():
    V, d = C.CFG["vocab"], C.CFG["d_model"]
    out = {"_meta": "reference GPT2ILQLTrain._step (ilql/gpt2/interface.py) executed under the numpy jax / flax stand-ins; transformer = float64 oracle"}
    for case in C.ILQL_CASES:
        sd, tsd = C.state_dict(10 + case["seed"]), C.state_dict(20 + case["seed"])
        q1, q2, v = C.mlp_head(30 + case["seed"], V), C.mlp_head(40 + case["seed"], V), C.mlp_head(50 + case["seed"], 1)
        tq1, tq2 = C.mlp_head(60 + case["seed"], V), C.mlp_head(70 + case["seed"], V)
        q_model = MLPHead(MLPHeadConfig(input_dim=d, hidden_dim=d, output_dim=V, mesh="mesh"))
        v_model = MLPHead(MLPHeadConfig(input_dim=d, hidden_dim=d, output_dim=1, mesh="mesh"))
        ms, st0 = case["mini_step"], case["step0"]
        train = GPT2ILQLTrain.load_train(
            base_train_state=FakeTrainState(sd, st0, ms), target_base_params=tsd if case["target_base"] else None,
            q1_head_train_state=FakeTrainState(q1, st0, ms), q2_head_train_state=FakeTrainState(q2, st0, ms), v_head_train_state=FakeTrainState(v, st0, ms),
            q1_target_head_params=tq1, q2_target_head_params=tq2, base_model=FakeGPT2(), q_head_model=q_model, v_head_model=v_model, tokenizer=None,
            loss_fn=partial(ilql_loss, **C.LOSS_KW), detach_q1=False, detach_q2=False, detach_v=False, polyak_alpha=case["polyak_alpha"],
            hard_update_every=case["hard_update_every"])
        b = C.ilql_batch(case["seed"])
        A = S.asarray
        nxt = [A(b[k]) for k in ("next_token_ids", "next_tokens_attention_mask", "next_tokens_position_ids", "next_dones")] if case["use_next"] else [None] * 4
        res = train._step(train.base_train_state, train.target_base_params, train.q1_head_train_state, train.q2_head_train_state, train.v_head_train_state,
                          train.q1_target_head_params, train.q2_target_head_params, A(b["input_ids"]), A(b["attention_mask"]), A(b["position_ids"]),
                          A(b["should_take_action"]), A(b["rewards"]), A(b["dones"]), *nxt, None, True)
        base_ts, tbase, q1_ts, q2_ts, v_ts, tq1_new, tq2_new, loss, info = res
        out[case["name"]] = dict(loss=float(np.asarray(loss)), logs=flat_logs(info), step_after=int(base_ts.step),
                                 q1_target=digest(tq1_new), q2_target=digest(tq2_new),
                                 target_base=None if tbase is None else digest({k: tbase[k] for k in ("wte.weight", "h.1.mlp.c_proj.weight", "ln_f.bias")}))
        print(case["name"], "loss", out[case["name"]]["loss"], "v_final.mean", out[case["name"]]["logs"]["v_final.mean"])
    # ---- PPO closure (ppo/gpt2/interface.py:72-211): values / logprobs wiring, the loss call, the BC combination
    for case in C.PPO_CASES:
        sd = C.state_dict(80 + case["seed"])
        vh = C.linear_head(90 + case["seed"])
        v_model = LinearHead(LinearHeadConfig(input_dim=d, output_dim=1, mesh="mesh"))
        bc_fn = None
        if case["bc_weight"] is not None:
            # (the scripts pass JaxSeq's loss_fn_mask here — third party, absent: any (loss, info) callable pins how the closure combines it)
            bc_fn = lambda model, params, ids, am, pos, tm, key, train: (S.asarray(np.float32(1.75)), {"loss": S.asarray(np.float32(1.75))})
        train = GPT2PPOTrain.load_train(policy_train_state=FakeTrainState(sd, 0, None), value_head_train_state=FakeTrainState(vh, 0, None), policy_model=FakeGPT2(),
                                        value_head_model=v_model, tokenizer=None, loss_fn=partial(ppo_loss_fn, **C.PPO_KW), bc_loss_fn=bc_fn,
                                        bc_loss_weight=case["bc_weight"] or 0.0)
        b = C.ppo_batch(case["seed"])
        A = S.asarray
        bc_args = [A(b["input_ids"]), A(b["attention_mask"]), A(b["position_ids"]), A(b["attention_mask"])] if bc_fn is not None else [None] * 4
        res = train._step(train.policy_train_state, train.value_head_train_state, A(b["input_ids"]), A(b["attention_mask"]), A(b["position_ids"]),
                          A(b["should_take_action"]), A(b["old_logprobs"]), A(b["old_values"]), A(b["old_advantages"]), A(b["old_returns"]), None, *bc_args, True)
        _, _, loss, info = res
        out[case["name"]] = dict(loss=float(np.asarray(loss)), logs=flat_logs(info))
        print(case["name"], "loss", out[case["name"]]["loss"])
    # ---- MC-returns closure (mc_returns/gpt2/interface.py:38-160): Q head, Q(s, a) gather, the mc_loss call
    case = C.MC_CASE
    sd, qh = C.state_dict(110 + case["seed"]), C.mlp_head(120 + case["seed"], V)
    q_model = MLPHead(MLPHeadConfig(input_dim=d, hidden_dim=d, output_dim=V, mesh="mesh"))
    train = GPT2MCTrain.load_train(base_train_state=FakeTrainState(sd, 0, None), q_head_train_state=FakeTrainState(qh, 0, None), base_model=FakeGPT2(),
                                   q_head_model=q_model, tokenizer=None, loss_fn=partial(mc_loss, cql_weight=case["cql_weight"]), detach_q=False)
    b = C.mc_batch(case["seed"])
    A = S.asarray
    res = train._step(train.base_train_state, train.q_head_train_state, A(b["input_ids"]), A(b["attention_mask"]), A(b["position_ids"]),
                      A(b["should_take_action"]), A(b["returns"]), None, True)
    out[case["name"]] = dict(loss=float(np.asarray(res[-2])), logs=flat_logs(res[-1]))
    print(case["name"], "loss", out[case["name"]]["loss"])
    # ---- GPT2ValueRLGeneration.__call__ (value_rl_base/gpt2/generation.py:36-121): logits = pi_beta + beta * min(q1, q2) at every position
    for case in C.VALUE_RL_CASES:
        pi_sd, base_sd = C.state_dict(130 + case["seed"]), C.state_dict(140 + case["seed"])
        q1, q2 = C.mlp_head(150 + case["seed"], V), C.mlp_head(160 + case["seed"], V)
        gen = GPT2ValueRLGeneration(types.SimpleNamespace(), FakeGPT2() if case["pi_beta"] else None, FakeGPT2(), q_model, case["beta"])
        b = C.ilql_batch(case["seed"])
        o = gen(A(b["input_ids"]), attention_mask=A(b["attention_mask"]), params=(pi_sd if case["pi_beta"] else None, base_sd, q1, q2 if case["q2"] else None),
                position_ids=A(b["position_ids"]))
        lg = np.asarray(o.logits, dtype=np.float64)
        last = b["attention_mask"].sum(1) - 1
        out[case["name"]] = dict(last_logits=[[float(x) for x in lg[i, last[i]]] for i in range(lg.shape[0])], all_sum=float(lg.sum()), all_sq=float((lg * lg).sum()))
        print(case["name"], "sum", out[case["name"]]["all_sum"])
    # ---- score functions (ppo/score_fn.py:10-126, ilql/gpt2/score_fn.py:11-68): token windows, prefix lengths, masked sums over the last action
    case = C.SCORE_CASE
    tok = C.CharTok()
    hists = [tuple(Text(t, a) for t, a in h) for h in C.score_histories()]
    pol_sd, base_sd = C.state_dict(170 + case["seed"]), C.state_dict(180 + case["seed"])
    q1, q2, vh = C.mlp_head(190 + case["seed"], V), C.mlp_head(200 + case["seed"], V), C.mlp_head(210 + case["seed"], 1)
    fake = FakeGPT2()

    def fwd(sd, tokens, attention_mask=None):
        am = np.asarray(attention_mask if attention_mask is not None else (np.asarray(tokens) != tok.pad_token_id)).astype(np.int64)
        return fake(tokens, attention_mask=am, params=sd)
    ppo_inf = types.SimpleNamespace(forward=lambda t_, attention_mask=None, train=False, prng_key=None: types.SimpleNamespace(policy_raw_output=fwd(pol_sd, t_, attention_mask)))
    bc_inf = types.SimpleNamespace(forward=lambda t_, attention_mask=None, train=False, prng_key=None: fwd(pol_sd, t_, attention_mask))

    def ilql_forward(batch):
        o = fwd(base_sd, batch)
        h = o.hidden_states[-1]
        return types.SimpleNamespace(q1=q_model.apply({"params": q1}, h, train=False), q2=q_model.apply({"params": q2}, h, train=False),
                                     v=S.squeeze(v_model_mlp.apply({"params": vh}, h, train=False), axis=2))
    v_model_mlp = MLPHead(MLPHeadConfig(input_dim=d, hidden_dim=d, output_dim=1, mesh="mesh"))
    ilql_inf = types.SimpleNamespace(forward=ilql_forward)
    pib_inf = types.SimpleNamespace(get_logits_from_tokens=lambda batch: fwd(pol_sd, batch).logits)
    L, bs = C.SCORE_MAX_LENGTH, C.SCORE_BSIZE
    out[case["name"]] = dict(
        ppo=build_ppo_score_fn(ppo_inf, tok, L, bs)(hists), bc=build_bc_score_fn(bc_inf, tok, L, bs)(hists),
        ilql=build_ilql_score_fn(ilql_inf, None, tok, L, case["value_weight"], None, bs)(hists),
        ilql_with_logits=build_ilql_score_fn(ilql_inf, pib_inf, tok, L, case["value_weight"], case["logit_weight"], bs)(hists))
    print(case["name"], [round(x, 3) for x in out[case["name"]]["ppo"]], [round(x, 3) for x in out[case["name"]]["ilql_with_logits"]])
    # ---- PPOInference.get_ppo_data_from_token_trajectory_chain (ppo/base_interface.py:464-669), the whole function: forwards in batches,
    # un-padding, bootstrap value x (1 - done), KL penalty on the rewards, GAE per chain, whitening over the batch, chunk unrolling
    import enum
    from typing import NamedTuple
    import LLM_RL.algorithms.ppo.base_interface as PB
    from LLM_RL.environment import TokenTrajectory, TokenTrajectoryChain

    class Padding(enum.Enum):
        LEFT = "left"; RIGHT = "right"

    class Truncation(enum.Enum):
        LEFT = "left"; RIGHT = "right"

    class BlockingStrategy(NamedTuple):
        padding: Padding
        truncation: Truncation
        max_length: object

    def block_sequences(sequences, pad_value, dtype, blocking_strategy):
        # JaxSeq.utils.block_sequences (third party, absent): pad / truncate to max_length (default: the longest sequence)
        L = blocking_strategy.max_length or max(len(x) for x in sequences)
        o = np.full((len(sequences), L), pad_value, dtype=dtype)
        for i, x in enumerate(sequences):
            x = list(x)[:L] if blocking_strategy.truncation == Truncation.RIGHT else list(x)[-L:]
            if blocking_strategy.padding == Padding.RIGHT:
                o[i, :len(x)] = x
            else:
                o[i, L - len(x):] = x
        return o
    PB.Padding, PB.Truncation, PB.BlockingStrategy, PB.block_sequences = Padding, Truncation, BlockingStrategy, block_sequences
    PB.multihost_device_get = lambda x, mesh=None: x
    case = C.PPO_DATA_CASE
    init_sd, vh = C.state_dict(230 + case["seed"]), C.linear_head(240 + case["seed"])
    pol_sd = C.perturbed(init_sd, 220 + case["seed"])
    lin_model = LinearHead(LinearHeadConfig(input_dim=d, output_dim=1, mesh="mesh"))
    fake = FakeGPT2()

    def ppo_forward(tokens_batch, train=False, prng_key=None):
        am = (np.asarray(tokens_batch) != C.PAD).astype(np.int64)
        po, io = fake(tokens_batch, attention_mask=am, params=pol_sd), fake(tokens_batch, attention_mask=am, params=init_sd)
        values = S.squeeze(lin_model.apply({"params": vh}, po.hidden_states[-1], train=False), axis=2)
        return types.SimpleNamespace(initial_policy_raw_output=io, policy_raw_output=po, values=values)
    fake_self = types.SimpleNamespace(initial_policy_model=types.SimpleNamespace(config=types.SimpleNamespace(mesh="mesh")), initial_policy_params=init_sd,
                                      policy_model=types.SimpleNamespace(config=types.SimpleNamespace(mesh="mesh")),
                                      tokenizer=types.SimpleNamespace(pad_token_id=C.PAD), forward=ppo_forward,
                                      token_logprobs_from_logits=PB.PPOInference.token_logprobs_from_logits)
    chains = []
    for ch in C.ppo_chains(case["seed"]):
        node = None
        for tt in reversed(ch):
            node = TokenTrajectoryChain(TokenTrajectory(tt["tokens"], tt["is_action"], tt["reward"], np.asarray(tt["done"])), node)
        chains.append(node)
    datas, kls = PB.PPOInference.get_ppo_data_from_token_trajectory_chain(fake_self, chains, case["bsize"], None, verbose=False, gamma=case["gamma"],
                                                                          lam=case["lam"], kl_weight=case["kl_weight"])
    tl = lambda a: [float(x) for x in np.asarray(a).ravel()]
    out[case["name"]] = dict(kls=tl(kls), datas=[dict(input_ids=[int(x) for x in dd.input_ids], should_take_action=[bool(x) for x in dd.should_take_action],
                                                      old_logprobs=tl(dd.old_logprobs), old_values=tl(dd.old_values), old_advantages=tl(dd.old_advantages),
                                                      old_returns=tl(dd.old_returns)) for dd in datas])
    print(case["name"], len(datas), "chunks,", len(kls), "action tokens, mean kl", float(np.mean(kls)))
    path = os.path.join(HERE, "rl_steps.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(f"wrote rl_steps.json: {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
