"""CPU tier: `oracle/rl.py`'s restatement of the ILQL train-step closure (heads, Q(s, a) gathers, v / v_final in both branches, the loss
call) against tests/golden/rl_steps.json — outputs of the reference's OWN `GPT2ILQLTrain._step` (ilql/gpt2/interface.py:88-367) executed
under numpy stand-ins for jax / flax (tests/golden/make_step_fixtures.py).  The transformer slot of the closure was filled with this
oracle's GPT-2, so the comparison isolates exactly the code between the model calls and the returned loss.  Tolerance: the reference ran
in float32 (numpy), the oracle in float64 — 2e-5 relative."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import step_cases as C  # noqa: E402
from conftest import load_golden  # noqa: E402
from oracle import gpt2 as O, rl  # noqa: E402


def _flat(d, prefix=""):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(_flat(v, prefix + k + "."))
        else:
            out[prefix + k] = float(v)
    return out


def oracle_step(case):
    t = lambda a: torch.from_numpy(np.asarray(a))
    V = C.CFG["vocab"]
    sd = {k: t(v) for k, v in C.state_dict(10 + case["seed"]).items()}
    tsd = {k: t(v) for k, v in C.state_dict(20 + case["seed"]).items()} if case["target_base"] else sd
    heads = [C.flat_head(C.mlp_head(s + case["seed"], o)) for s, o in ((30, V), (40, V), (50, 1), (60, V), (70, V))]
    mh = lambda x, h: rl.mlp_head(x, t(h["dense1.kernel"]), t(h["dense1.bias"]), t(h["dense2.kernel"]), t(h["dense2.bias"]))
    b = C.ilql_batch(case["seed"])
    ids, am, pos = t(b["input_ids"]).long(), t(b["attention_mask"]), t(b["position_ids"]).long()
    _, hid = O.forward(sd, ids, C.CFG["n_head"], attention_mask=am, position_ids=pos, return_hidden=True)
    _, thid = O.forward(tsd, ids, C.CFG["n_head"], attention_mask=am, position_ids=pos, return_hidden=True)
    q1o, q2o, vo, tq1o, tq2o = mh(hid, heads[0]), mh(hid, heads[1]), mh(hid, heads[2]), mh(thid, heads[3]), mh(thid, heads[4])
    nxt = {}
    if case["use_next"]:
        nam = t(b["next_tokens_attention_mask"])
        _, nhid = O.forward(sd, t(b["next_token_ids"]).long(), C.CFG["n_head"], attention_mask=nam, position_ids=t(b["next_tokens_position_ids"]).long(),
                            return_hidden=True)
        nxt = dict(next_v_head_out=mh(nhid, heads[2]), next_attention_mask=nam, next_dones=t(b["next_dones"]))
    sta = t(b["should_take_action"])
    q1, q2, v, v_final, tq1, tq2 = rl.ilql_gather_qv(q1o, q2o, vo, tq1o, tq2o, ids, am, sta, t(b["dones"]), **nxt)
    loss, logs = rl.ilql_loss(q1, q2, v, v_final, tq1, tq2, q1o[:, :-1], q2o[:, :-1], ids[:, 1:], am[:, 1:].double(), sta, t(b["rewards"]).double(), **C.LOSS_KW)
    return float(loss), _flat(logs)


@pytest.mark.parametrize("case", C.ILQL_CASES, ids=[c["name"] for c in C.ILQL_CASES])
def test_oracle_ilql_closure_equals_reference_step(case):
    fx = load_golden("rl_steps.json")[case["name"]]
    loss, logs = oracle_step(case)
    assert abs(loss - fx["loss"]) <= 2e-5 * abs(fx["loss"]), (loss, fx["loss"])
    assert set(logs) == set(fx["logs"])
    for k, e in fx["logs"].items():
        assert abs(logs[k] - e) <= 2e-5 * max(1.0, abs(e)), (k, logs[k], e)


def test_reference_target_updates_follow_the_stated_rule():
    """The fixture's target parameters after the step: the closure's stand-in optimizer scaled every online parameter by 0.9; targets then are
    Polyak-averaged (`optax.incremental_update`), hard-copied when `TrainState.step` (AFTER its increment) hits `hard_update_every`, and left
    alone on an accumulating micro-step (`opt_state.mini_step != 0`) — interface.py:327-365.  (The package's device implementation is compared
    with the same digests in tests/test_gpu_train_steps_pinned.py.)"""
    V = C.CFG["vocab"]
    fxs = load_golden("rl_steps.json")
    for case in C.ILQL_CASES:
        fx = fxs[case["name"]]
        assert fx["step_after"] == case["step0"] + 1
        online, target = C.flat_head(C.mlp_head(30 + case["seed"], V)), C.flat_head(C.mlp_head(60 + case["seed"], V))
        a = case["polyak_alpha"]
        hard = case["hard_update_every"] is not None and fx["step_after"] % case["hard_update_every"] == 0
        for name, old in target.items():
            new = (online[name] * np.float32(0.9)).astype(np.float32)
            if case["mini_step"] not in (None, 0):
                exp = old
            elif hard:
                exp = new
            else:
                exp = (np.float32(a) * new + np.float32(1.0 - a) * old).astype(np.float32)
            got = fx["q1_target"][name]
            assert abs(got[0] - float(exp.astype(np.float64).sum())) <= 1e-4 * max(1.0, abs(got[0])), (case["name"], name)
            np.testing.assert_allclose(got[2:], exp.ravel()[:3].astype(np.float64), rtol=1e-5, atol=1e-7)
        assert (fx["target_base"] is None) == (not case["target_base"])


@pytest.mark.parametrize("case", C.PPO_CASES, ids=[c["name"] for c in C.PPO_CASES])
def test_oracle_ppo_closure_equals_reference_step(case):
    """ppo/gpt2/interface.py:111-133: values = LinearHead(hidden)[:, :-1], logprobs = -CE(logits[:, :-1], ids[:, 1:]), the loss call, and
    (:180-203) info = {'ppo', 'bc', 'total_loss'}, loss = ppo + bc_loss_weight * bc (the fixture's bc callable returned 1.75)."""
    fx = load_golden("rl_steps.json")[case["name"]]
    t = lambda a: torch.from_numpy(np.asarray(a))
    sd = {k: t(v) for k, v in C.state_dict(80 + case["seed"]).items()}
    vh = C.flat_head(C.linear_head(90 + case["seed"]))
    b = C.ppo_batch(case["seed"])
    ids, am = t(b["input_ids"]).long(), t(b["attention_mask"])
    logits, hid = O.forward(sd, ids, C.CFG["n_head"], attention_mask=am, position_ids=t(b["position_ids"]).long(), return_hidden=True)
    values = rl.linear_head(hid, t(vh["dense.kernel"]), t(vh["dense.bias"]))[:, :-1, 0]
    logprobs = rl.token_logprobs_from_logits(logits, ids)
    td = lambda k: t(b[k]).double()
    loss, logs = rl.ppo_loss(am[:, 1:].double(), logprobs, values, t(b["should_take_action"]), td("old_logprobs"), td("old_values"), td("old_advantages"),
                             td("old_returns"), **C.PPO_KW)
    got = _flat(logs)
    if case["bc_weight"] is not None:
        total = float(loss) + 1.75 * case["bc_weight"]
        got = {**{"ppo." + k: v for k, v in got.items()}, "bc.loss": 1.75, "total_loss": total}
        loss = total
    assert abs(float(loss) - fx["loss"]) <= 2e-5 * abs(fx["loss"]), (float(loss), fx["loss"])
    assert set(got) == set(fx["logs"])
    for k, e in fx["logs"].items():
        assert abs(got[k] - e) <= 2e-5 * max(1.0, abs(e)), (k, got[k], e)


def test_oracle_mc_closure_equals_reference_step():
    """mc_returns/gpt2/interface.py:96-121: Q head on the base hidden states, Q(s, a) gather on the shifted ids, mc_loss."""
    case = C.MC_CASE
    fx = load_golden("rl_steps.json")[case["name"]]
    t = lambda a: torch.from_numpy(np.asarray(a))
    sd = {k: t(v) for k, v in C.state_dict(110 + case["seed"]).items()}
    qh = C.flat_head(C.mlp_head(120 + case["seed"], C.CFG["vocab"]))
    b = C.mc_batch(case["seed"])
    ids, am = t(b["input_ids"]).long(), t(b["attention_mask"])
    _, hid = O.forward(sd, ids, C.CFG["n_head"], attention_mask=am, position_ids=t(b["position_ids"]).long(), return_hidden=True)
    qo = rl.mlp_head(hid, t(qh["dense1.kernel"]), t(qh["dense1.bias"]), t(qh["dense2.kernel"]), t(qh["dense2.bias"]))
    q = qo[:, :-1].gather(2, ids[:, 1:].unsqueeze(-1)).squeeze(2)
    loss, logs = rl.mc_loss(q, qo[:, :-1], ids[:, 1:], am[:, 1:].double(), t(b["should_take_action"]), t(b["returns"]).double(), cql_weight=case["cql_weight"])
    assert abs(float(loss) - fx["loss"]) <= 2e-5 * abs(fx["loss"])
    got = _flat(logs)
    assert set(got) == set(fx["logs"])
    for k, e in fx["logs"].items():
        assert abs(got[k] - e) <= 2e-5 * max(1.0, abs(e)), (k, got[k], e)


@pytest.mark.parametrize("case", C.VALUE_RL_CASES, ids=[c["name"] for c in C.VALUE_RL_CASES])
def test_oracle_value_rl_logits_equal_reference_generation_call(case):
    """value_rl_base/gpt2/generation.py:36-121 executed as it is (pi_beta optional, q2 optional): logits = pi_beta + beta * min(q1, q2)."""
    fx = load_golden("rl_steps.json")[case["name"]]
    t = lambda a: torch.from_numpy(np.asarray(a))
    V = C.CFG["vocab"]
    pi_sd = {k: t(v) for k, v in C.state_dict(130 + case["seed"]).items()}
    base_sd = {k: t(v) for k, v in C.state_dict(140 + case["seed"]).items()}
    h1, h2 = C.flat_head(C.mlp_head(150 + case["seed"], V)), C.flat_head(C.mlp_head(160 + case["seed"], V))
    mh = lambda x, h: rl.mlp_head(x, t(h["dense1.kernel"]), t(h["dense1.bias"]), t(h["dense2.kernel"]), t(h["dense2.bias"]))
    b = C.ilql_batch(case["seed"])
    ids, am, pos = t(b["input_ids"]).long(), t(b["attention_mask"]), t(b["position_ids"]).long()
    pi_logits = O.forward(pi_sd, ids, C.CFG["n_head"], attention_mask=am, position_ids=pos) if case["pi_beta"] else None
    _, hid = O.forward(base_sd, ids, C.CFG["n_head"], attention_mask=am, position_ids=pos, return_hidden=True)
    lg = rl.value_rl_logits(pi_logits, mh(hid, h1), mh(hid, h2) if case["q2"] else None, case["beta"]).numpy()
    last = b["attention_mask"].sum(1) - 1
    exp = np.asarray(fx["last_logits"])
    scale = np.abs(exp).max()
    np.testing.assert_allclose(np.stack([lg[i, last[i]] for i in range(len(last))]), exp, rtol=0, atol=3e-5 * scale)
    assert abs(float(lg.sum()) - fx["all_sum"]) <= 1e-5 * abs(fx["all_sum"]) and abs(float((lg * lg).sum()) - fx["all_sq"]) <= 1e-5 * fx["all_sq"]


def test_oracle_ppo_data_pipeline_equals_reference_function():
    """`rl.ppo_data_from_chains` (+ the oracle forwards) against the reference's WHOLE `get_ppo_data_from_token_trajectory_chain`
    (ppo/base_interface.py:464-669) executed under the stand-ins: per-chunk PPOData fields and the KL vector."""
    case = C.PPO_DATA_CASE
    fx = load_golden("rl_steps.json")[case["name"]]
    t = lambda a: torch.from_numpy(np.asarray(a))
    init_np = C.state_dict(230 + case["seed"])
    init_sd = {k: t(v) for k, v in init_np.items()}
    pol_sd = {k: t(v) for k, v in C.perturbed(init_np, 220 + case["seed"]).items()}
    vh = C.flat_head(C.linear_head(240 + case["seed"]))
    chains = C.ppo_chains(case["seed"])
    lp_c, ilp_c, v_c, chain_dicts = [], [], [], []
    for ch in chains:
        lps, ilps, vs = [], [], []
        for tt in ch:
            ids = t(tt["tokens"]).long()[None]
            lg, hid = O.forward(pol_sd, ids, C.CFG["n_head"], return_hidden=True)
            ilg = O.forward(init_sd, ids, C.CFG["n_head"])
            lps.append(rl.token_logprobs_from_logits(lg, ids)[0].numpy()); ilps.append(rl.token_logprobs_from_logits(ilg, ids)[0].numpy())
            v = rl.linear_head(hid, t(vh["dense.kernel"]), t(vh["dense.bias"]))[0, :, 0].numpy()
            vs.append(v[:-1]); last = v[-1]
        lp_c.append(np.concatenate(lps)); ilp_c.append(np.concatenate(ilps))
        v_c.append(np.concatenate(vs + [np.array([last * (1.0 - float(ch[-1]["done"]))])]))
        chain_dicts.append([dict(tokens=tt["tokens"].tolist(), is_action=tt["is_action"].tolist(), reward=tt["reward"].tolist(), done=tt["done"]) for tt in ch])
    ref, kls = rl.ppo_data_from_chains(chain_dicts, lp_c, ilp_c, v_c, gamma=case["gamma"], lam=case["lam"], kl_weight=case["kl_weight"])
    np.testing.assert_allclose(kls, fx["kls"], rtol=2e-4, atol=2e-6)
    k = 0
    for ch, r in zip(chains, ref):
        offs = np.cumsum([0] + r["chunk_lens"])
        for j, tt in enumerate(ch):
            d = fx["datas"][k]; k += 1
            sl = slice(offs[j], offs[j + 1])
            assert d["input_ids"] == tt["tokens"].tolist() and d["should_take_action"] == [bool(x) for x in r["should_take_action"][sl]]
            for name in ("old_logprobs", "old_values", "old_advantages", "old_returns"):
                np.testing.assert_allclose(r[name][sl], d[name], rtol=3e-5, atol=3e-5, err_msg=name)
    assert k == len(fx["datas"])


# ------------------------------------------------------------------------------------------------------------------------------------------
# GRADIENTS pinned to the reference: tests/golden/rl_step_grads.json holds <dL/dtheta, v> of the reference's own `_step`
# closures, obtained by complex-step differentiation THROUGH THE REFERENCE'S CODE (stop_gradient drops the imaginary part; generator
# tests/golden/make_step_grad_fixtures.py).  Here: float64 autograd of the restatement, contracted with the same seeded directions.
def _dot(grads: dict, v: dict) -> float:
    return float(sum((np.asarray(grads[k], dtype=np.float64) * v[k]).sum() for k in v))


def _oracle_ilql_grads(case):
    t = lambda a: torch.from_numpy(np.asarray(a)).double()
    V = C.CFG["vocab"]
    sd = {k: t(v).requires_grad_(True) for k, v in C.state_dict(10 + case["seed"]).items()}
    tsd = {k: t(v) for k, v in C.state_dict(20 + case["seed"]).items()} if case["target_base"] else {k: v.detach() for k, v in sd.items()}
    names = (("q1", 30, V), ("q2", 40, V), ("v", 50, 1))
    heads = {n: {k: t(x).requires_grad_(True) for k, x in C.flat_head(C.mlp_head(s + case["seed"], o)).items()} for n, s, o in names}
    tq = [{k: t(x) for k, x in C.flat_head(C.mlp_head(s + case["seed"], V)).items()} for s in (60, 70)]
    mh = lambda x, h: rl.mlp_head(x, h["dense1.kernel"], h["dense1.bias"], h["dense2.kernel"], h["dense2.bias"])
    b = C.ilql_batch(case["seed"])
    ti = lambda a: torch.from_numpy(np.asarray(a))
    ids, am, pos = ti(b["input_ids"]).long(), ti(b["attention_mask"]), ti(b["position_ids"]).long()
    _, hid = O.forward(sd, ids, C.CFG["n_head"], attention_mask=am, position_ids=pos, return_hidden=True)
    with torch.no_grad():
        _, thid = O.forward(tsd, ids, C.CFG["n_head"], attention_mask=am, position_ids=pos, return_hidden=True)
        tq1o, tq2o = mh(thid, tq[0]), mh(thid, tq[1])
    q1o, q2o, vo = mh(hid, heads["q1"]), mh(hid, heads["q2"]), mh(hid, heads["v"])
    nxt = {}
    if case["use_next"]:
        nam = ti(b["next_tokens_attention_mask"])
        _, nhid = O.forward(sd, ti(b["next_token_ids"]).long(), C.CFG["n_head"], attention_mask=nam, position_ids=ti(b["next_tokens_position_ids"]).long(),
                            return_hidden=True)
        nxt = dict(next_v_head_out=mh(nhid, heads["v"]), next_attention_mask=nam, next_dones=ti(b["next_dones"]))
    sta = ti(b["should_take_action"])
    q1, q2, v, v_final, tq1, tq2 = rl.ilql_gather_qv(q1o, q2o, vo, tq1o, tq2o, ids, am, sta, ti(b["dones"]), **nxt)
    loss, _ = rl.ilql_loss(q1, q2, v, v_final, tq1, tq2, q1o[:, :-1], q2o[:, :-1], ids[:, 1:], am[:, 1:].double(), sta, ti(b["rewards"]).double(), **C.LOSS_KW)
    loss.backward()
    g = {"base." + k: p.grad.numpy() for k, p in sd.items()}
    flat = {"base." + k: p.detach().numpy() for k, p in sd.items()}
    for n, h in heads.items():
        g.update({f"{n}.{k}": p.grad.numpy() for k, p in h.items()})
        flat.update({f"{n}.{k}": p.detach().numpy() for k, p in h.items()})
    return float(loss), g, flat


@pytest.mark.parametrize("case", C.ILQL_CASES, ids=[c["name"] for c in C.ILQL_CASES])
def test_oracle_ilql_gradients_equal_reference_complex_step(case):
    fx = load_golden("rl_step_grads.json")
    loss, g, flat = _oracle_ilql_grads(case)
    assert abs(loss - fx[case["name"]]["loss"]) <= 1e-9 * abs(loss)
    for dseed, ref in zip(fx["direction_seeds"], fx[case["name"]]["ddir"]):
        v = C.direction(dseed, flat)
        got = _dot(g, v)
        scale = sum(float(np.abs(np.asarray(g[k]) * v[k]).sum()) for k in v)          # no cancellation hiding: relative to sum |g_i v_i|
        assert abs(got - ref) <= 1e-9 * scale, (case["name"], dseed, got, ref)
