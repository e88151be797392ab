"""GPU tier: `GPT2ILQLTrain.step` (heads, Q gathers, both `v_final` branches, the loss, the target updates) against tests/golden/rl_steps.json —
outputs of the reference's OWN `_step` closure (ilql/gpt2/interface.py:88-367) run under numpy stand-ins for jax / flax with the oracle GPT-2 in
the transformer slot (tests/golden/make_step_fixtures.py).  Loss and every log entry within 1e-4 relative (fp32 device vs the fixture's fp32 numpy
on float64 hidden states); the Polyak / periodic / micro-step target-update rule against the fixture's parameter digests."""
import os
import sys

import numpy as np
import pytest

import lmrl_gym_amd  # noqa: F401

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import step_cases as C  # noqa: E402
from conftest import load_golden  # noqa: E402


def _flat(d, prefix=""):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(_flat(v, prefix + k + "."))
        else:
            out[prefix + k] = float(v)
    return out


@pytest.mark.parametrize("case", C.ILQL_CASES, ids=[c["name"] for c in C.ILQL_CASES])
def test_ilql_step_equals_reference_closure(case):
    from lmrl_gym_amd import _lib
    from lmrl_gym_amd.algorithms import ilql
    from lmrl_gym_amd.train.gpt2_f32 import GPT2F32, MLPHeadF32
    dev = _lib.require_gpu()
    fx = load_golden("rl_steps.json")[case["name"]]
    V = C.CFG["vocab"]
    t = lambda a: torch.from_numpy(np.asarray(a))
    sd = {k: t(v) for k, v in C.state_dict(10 + case["seed"]).items()}
    tsd = {k: t(v) for k, v in C.state_dict(20 + case["seed"]).items()}
    hp = [{k: t(v) for k, v in C.flat_head(C.mlp_head(s + case["seed"], o)).items()} for s, o in ((30, V), (40, V), (50, 1), (60, V), (70, V))]
    base = GPT2F32(sd, C.CFG["n_head"], device=dev)
    tbase = GPT2F32(tsd, C.CFG["n_head"], device=dev) if case["target_base"] else None
    tr = ilql.GPT2ILQLTrain(base, MLPHeadF32(hp[0], dev), MLPHeadF32(hp[1], dev), MLPHeadF32(hp[2], dev), C.PAD, C.LOSS_KW, target_base=tbase, lr=1e-4,
                            polyak_alpha=case["polyak_alpha"], hard_update_every=case["hard_update_every"])
    tr.q1_target, tr.q2_target = MLPHeadF32(hp[3], dev), MLPHeadF32(hp[4], dev)      # the fixture's target heads differ from the online ones
    b = C.ilql_batch(case["seed"])
    kw = dict(next_token_ids=b["next_token_ids"], next_dones=b["next_dones"]) if case["use_next"] else {}
    loss, logs = ilql.GPT2ILQLInference(base, tr.q1, tr.q2, tr.v, C.PAD, loss_kwargs=C.LOSS_KW, target_base=tbase, q1_target_head=tr.q1_target,
                                        q2_target_head=tr.q2_target).eval_loss(b["input_ids"], b["should_take_action"], b["rewards"], b["dones"], **kw)
    assert abs(loss - fx["loss"]) <= 1e-4 * abs(fx["loss"]), (loss, fx["loss"])
    got = _flat(logs)
    assert set(got) == set(fx["logs"])
    for k, e in fx["logs"].items():
        assert abs(got[k] - e) <= 1e-4 * max(1.0, abs(e)), (k, got[k], e)
    # the train step itself gives the same loss / logs (same forward) ...
    _, loss2, logs2 = tr.step(b["input_ids"], b["should_take_action"], b["rewards"], b["dones"], **kw)
    assert abs(loss2 - loss) <= 1e-6 * abs(loss)
    # ... and the target-update rule, applied to the fixture's (online x 0.9, target) pair at the fixture's step counter
    online = {k: (v * 0.9).to(dev) for k, v in hp[0].items()}
    target = {k: v.clone().to(dev) for k, v in hp[3].items()}
    if case["mini_step"] in (None, 0):            # (an accumulating micro-step leaves the targets alone: the trainer does not call the update)
        tr._update_targets(online, target, fx["step_after"])
    torch.cuda.synchronize()
    for name, d in fx["q1_target"].items():
        a = target[name].double().cpu().numpy().ravel()
        assert abs(float(a.sum()) - d[0]) <= 1e-4 * max(1.0, abs(d[0])), (case["name"], name)
        np.testing.assert_allclose(a[:3], d[2:], rtol=1e-5, atol=1e-7)


def test_ppo_step_equals_reference_closure():
    """`GPT2PPOTrain.step` loss / logs == the reference's PPO `_step` closure (ppo/gpt2/interface.py:72-211) on the plain case."""
    from lmrl_gym_amd import _lib
    from lmrl_gym_amd.algorithms import ppo
    from lmrl_gym_amd.train.gpt2_f32 import GPT2F32, LinearHeadF32
    dev = _lib.require_gpu()
    case = C.PPO_CASES[0]
    fx = load_golden("rl_steps.json")[case["name"]]
    t = lambda a: torch.from_numpy(np.asarray(a))
    sd = {k: t(v) for k, v in C.state_dict(80 + case["seed"]).items()}
    vh = C.flat_head(C.linear_head(90 + case["seed"]))
    pol = GPT2F32(sd, C.CFG["n_head"], device=dev)
    head = LinearHeadF32(dict(kernel=t(vh["dense.kernel"]), bias=t(vh["dense.bias"])), dev)
    tr = ppo.GPT2PPOTrain(pol, head, C.PAD, C.PPO_KW, lr=1e-5)
    b = C.ppo_batch(case["seed"])
    _, loss, logs = tr.step(b["input_ids"], b["should_take_action"], b["old_logprobs"], b["old_values"], b["old_advantages"], b["old_returns"])
    assert abs(loss - fx["loss"]) <= 1e-4 * abs(fx["loss"]), (loss, fx["loss"])
    got = _flat(logs)
    assert set(got) == set(fx["logs"])
    for k, e in fx["logs"].items():
        assert abs(got[k] - e) <= 1e-4 * max(1.0, abs(e)), (k, got[k], e)


def test_mc_step_equals_reference_closure():
    from lmrl_gym_amd import _lib
    from lmrl_gym_amd.algorithms import mc_returns as mc
    from lmrl_gym_amd.train.gpt2_f32 import GPT2F32, MLPHeadF32
    dev = _lib.require_gpu()
    case = C.MC_CASE
    fx = load_golden("rl_steps.json")[case["name"]]
    t = lambda a: torch.from_numpy(np.asarray(a))
    base = GPT2F32({k: t(v) for k, v in C.state_dict(110 + case["seed"]).items()}, C.CFG["n_head"], device=dev)
    head = MLPHeadF32({k: t(v) for k, v in C.flat_head(C.mlp_head(120 + case["seed"], C.CFG["vocab"])).items()}, dev)
    tr = mc.GPT2MCTrain(base, head, C.PAD, dict(cql_weight=case["cql_weight"]), lr=1e-4)
    b = C.mc_batch(case["seed"])
    _, loss, logs = tr.step(b["input_ids"], b["should_take_action"], b["returns"])
    assert abs(loss - fx["loss"]) <= 1e-4 * abs(fx["loss"]), (loss, fx["loss"])
    got = _flat(logs)
    assert set(got) == set(fx["logs"])
    for k, e in fx["logs"].items():
        assert abs(got[k] - e) <= 1e-4 * max(1.0, abs(e)), (k, got[k], e)


@pytest.mark.parametrize("case", C.VALUE_RL_CASES, ids=[c["name"] for c in C.VALUE_RL_CASES])
def test_value_rl_logits_equal_reference_generation_call(case):
    """`GPT2ILQLInference.forward` (fp32 kernels) -> pi_beta + beta * min(q1, q2) == the reference's `GPT2ValueRLGeneration.__call__` output."""
    from lmrl_gym_amd import _lib
    from lmrl_gym_amd.algorithms import ilql
    from lmrl_gym_amd.train.gpt2_f32 import GPT2F32, MLPHeadF32
    dev = _lib.require_gpu()
    fx = load_golden("rl_steps.json")[case["name"]]
    t = lambda a: torch.from_numpy(np.asarray(a))
    V = C.CFG["vocab"]
    pi = GPT2F32({k: t(v) for k, v in C.state_dict(130 + case["seed"]).items()}, C.CFG["n_head"], device=dev)
    base = GPT2F32({k: t(v) for k, v in C.state_dict(140 + case["seed"]).items()}, C.CFG["n_head"], device=dev)
    q1 = MLPHeadF32({k: t(v) for k, v in C.flat_head(C.mlp_head(150 + case["seed"], V)).items()}, dev)
    q2 = MLPHeadF32({k: t(v) for k, v in C.flat_head(C.mlp_head(160 + case["seed"], V)).items()}, dev)
    b = C.ilql_batch(case["seed"])
    out = ilql.GPT2ILQLInference(base, q1, q2 if case["q2"] else None, None, C.PAD).forward(b["input_ids"])
    q = np.minimum(out.q1, out.q2) if case["q2"] else out.q1
    lg = np.float32(case["beta"]) * q
    if case["pi_beta"]:
        lg = ilql.GPT2ILQLInference(pi, q1, None, None, C.PAD).forward(b["input_ids"]).base_logits + lg
    last = b["attention_mask"].sum(1) - 1
    exp = np.asarray(fx["last_logits"])
    np.testing.assert_allclose(np.stack([lg[i, last[i]] for i in range(len(last))]), exp, rtol=0, atol=1e-4 * np.abs(exp).max())


def test_score_functions_equal_reference_score_fns():
    """`build_ppo_score_fn` / `build_bc_score_fn` / `build_ilql_score_fn` (token window = last max_length tokens right-padded, prefix length of
    the UNtruncated history, masked sum over the last action — incl. the reference's empty-slice result for an over-long history) against the
    reference's own functions executed under the stand-ins (ppo/score_fn.py:10-126, ilql/gpt2/score_fn.py:11-68)."""
    from lmrl_gym_amd import _lib
    from lmrl_gym_amd.algorithms import reranker
    from lmrl_gym_amd.environment import Text
    from lmrl_gym_amd.train.gpt2_f32 import GPT2F32, MLPHeadF32
    dev = _lib.require_gpu()
    case = C.SCORE_CASE
    fx = load_golden("rl_steps.json")[case["name"]]
    t = lambda a: torch.from_numpy(np.asarray(a))
    V = C.CFG["vocab"]
    pol = GPT2F32({k: t(v) for k, v in C.state_dict(170 + case["seed"]).items()}, C.CFG["n_head"], device=dev)
    base = GPT2F32({k: t(v) for k, v in C.state_dict(180 + case["seed"]).items()}, C.CFG["n_head"], device=dev)
    mk = lambda s, o: MLPHeadF32({k: t(v) for k, v in C.flat_head(C.mlp_head(s + case["seed"], o)).items()}, dev)
    q1, q2, vh = mk(190, V), mk(200, V), mk(210, 1)
    tok = C.CharTok()
    hists = [tuple(Text(x, a) for x, a in h) for h in C.score_histories()]
    L, bs = C.SCORE_MAX_LENGTH, C.SCORE_BSIZE
    close = lambda got, exp: np.testing.assert_allclose(got, exp, rtol=1e-4, atol=2e-4)
    close(reranker.build_ppo_score_fn(pol, tok, L, bs)(hists), fx["ppo"])
    close(reranker.build_bc_score_fn(pol, tok, L, bs)(hists), fx["bc"])
    close(reranker.build_ilql_score_fn(base, q1, q2, vh, tok, L, bs, value_weight=case["value_weight"])(hists), fx["ilql"])
    close(reranker.build_ilql_score_fn(base, q1, q2, vh, tok, L, bs, value_weight=case["value_weight"], pi_beta=pol, logit_weight=case["logit_weight"])(hists),
          fx["ilql_with_logits"])
    assert fx["ppo"][3] == 0.0 and fx["ppo"][0] < -50          # the over-long history scores exactly 0 in the reference (empty slice); others are real sums


def test_ppo_data_pipeline_equals_reference_function():
    """`GPT2PPOInference.get_ppo_data_from_token_trajectory_chain` on the device == the reference's whole function (ppo/base_interface.py:464-669)."""
    from lmrl_gym_amd import _lib, environment as E
    from lmrl_gym_amd.algorithms.ppo_inference import GPT2PPOInference
    from lmrl_gym_amd.train.gpt2_f32 import GPT2F32, LinearHeadF32
    dev = _lib.require_gpu()
    case = C.PPO_DATA_CASE
    fx = load_golden("rl_steps.json")[case["name"]]
    t = lambda a: torch.from_numpy(np.asarray(a))
    init_np = C.state_dict(230 + case["seed"])
    vh = C.flat_head(C.linear_head(240 + case["seed"]))
    pol = GPT2F32({k: t(v) for k, v in C.perturbed(init_np, 220 + case["seed"]).items()}, C.CFG["n_head"], device=dev)
    init = GPT2F32({k: t(v) for k, v in init_np.items()}, C.CFG["n_head"], device=dev)
    inf = GPT2PPOInference(pol, LinearHeadF32(dict(kernel=t(vh["dense.kernel"]), bias=t(vh["dense.bias"])), dev), C.PAD, initial_policy=init)
    chains = []
    for ch in C.ppo_chains(case["seed"]):
        node = None
        for tt in reversed(ch):
            node = E.TokenTrajectoryChain(E.TokenTrajectory(tt["tokens"], tt["is_action"], tt["reward"], np.asarray(tt["done"])), node)
        chains.append(node)
    datas, kls = inf.get_ppo_data_from_token_trajectory_chain(chains, bsize=case["bsize"], max_length=None, gamma=case["gamma"], lam=case["lam"],
                                                              kl_weight=case["kl_weight"])
    np.testing.assert_allclose(kls, fx["kls"], rtol=2e-3, atol=2e-5)
    assert len(datas) == len(fx["datas"])
    for d, e in zip(datas, fx["datas"]):
        assert d.input_ids.tolist() == e["input_ids"] and [bool(x) for x in d.should_take_action] == e["should_take_action"]
        np.testing.assert_allclose(d.old_logprobs, e["old_logprobs"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(d.old_values, e["old_values"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(d.old_returns, e["old_returns"], rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(d.old_advantages, e["old_advantages"], rtol=2e-3, atol=2e-3)


# ------------------------------------------------------------------------------------------------------------------------------------------
# DEVICE GRADIENTS against the reference's own closures: tests/golden/rl_step_grads.json = <dL/dtheta, v> by complex-step
# differentiation through the reference's `_step` code (tests/golden/make_step_grad_fixtures.py).  Tolerance: 3e-4 of sum |g_i v_i| (fp32 device;
# the float64 restatement agrees with the fixture to 1e-9: tests/test_oracle_steps_pinned.py).
def _check_directions(case_name, grads: dict, params: dict):
    fx = load_golden("rl_step_grads.json")
    for dseed, ref in zip(fx["direction_seeds"], fx[case_name]["ddir"]):
        v = C.direction(dseed, params)
        got = sum(float((grads[k].double().cpu().numpy() * v[k]).sum()) for k in v)
        scale = sum(float(np.abs(grads[k].double().cpu().numpy() * v[k]).sum()) for k in v)
        assert abs(got - ref) <= 3e-4 * scale, (case_name, dseed, got, ref, scale)


@pytest.mark.parametrize("case", C.ILQL_CASES, ids=[c["name"] for c in C.ILQL_CASES])
def test_ilql_step_gradients_equal_reference_complex_step(case):
    from lmrl_gym_amd import _lib
    from lmrl_gym_amd.algorithms import ilql
    from lmrl_gym_amd.train.gpt2_f32 import GPT2F32, MLPHeadF32
    dev = _lib.require_gpu()
    V = C.CFG["vocab"]
    t = lambda a: torch.from_numpy(np.asarray(a))
    sd_np = C.state_dict(10 + case["seed"])
    heads_np = [C.flat_head(C.mlp_head(s + case["seed"], o)) for s, o in ((30, V), (40, V), (50, 1), (60, V), (70, V))]
    base = GPT2F32({k: t(v) for k, v in sd_np.items()}, C.CFG["n_head"], device=dev)
    tbase = GPT2F32({k: t(v) for k, v in C.state_dict(20 + case["seed"]).items()}, C.CFG["n_head"], device=dev) if case["target_base"] else None
    hp = [{k: t(v) for k, v in h.items()} for h in heads_np]
    tr = ilql.GPT2ILQLTrain(base, MLPHeadF32(hp[0], dev), MLPHeadF32(hp[1], dev), MLPHeadF32(hp[2], dev), C.PAD, C.LOSS_KW, target_base=tbase, lr=1e-4,
                            polyak_alpha=case["polyak_alpha"], hard_update_every=case["hard_update_every"])
    tr.q1_target, tr.q2_target = MLPHeadF32(hp[3], dev), MLPHeadF32(hp[4], dev)
    b = C.ilql_batch(case["seed"])
    kw = dict(next_token_ids=b["next_token_ids"], next_dones=b["next_dones"]) if case["use_next"] else {}
    tr.step(b["input_ids"], b["should_take_action"], b["rewards"], b["dones"], **kw)
    bg, g1, g2, gv = tr.last_grads
    grads = {"base." + k: bg[k] for k in sd_np}
    params = {"base." + k: v for k, v in sd_np.items()}
    for n, g, h in (("q1", g1, heads_np[0]), ("q2", g2, heads_np[1]), ("v", gv, heads_np[2])):
        grads.update({f"{n}.{k}": g[k] for k in h})
        params.update({f"{n}.{k}": v for k, v in h.items()})
    _check_directions(case["name"], grads, params)


def test_ppo_and_mc_step_gradients_equal_reference_complex_step():
    from lmrl_gym_amd import _lib
    from lmrl_gym_amd.algorithms import mc_returns as mc, ppo
    from lmrl_gym_amd.train.gpt2_f32 import GPT2F32, LinearHeadF32, MLPHeadF32
    dev = _lib.require_gpu()
    t = lambda a: torch.from_numpy(np.asarray(a))
    case = C.PPO_CASES[0]
    sd_np, vh = C.state_dict(80 + case["seed"]), C.flat_head(C.linear_head(90 + case["seed"]))
    pol = GPT2F32({k: t(v) for k, v in sd_np.items()}, C.CFG["n_head"], device=dev)
    head = LinearHeadF32(dict(kernel=t(vh["dense.kernel"]), bias=t(vh["dense.bias"])), dev)
    tr = ppo.GPT2PPOTrain(pol, head, C.PAD, C.PPO_KW, lr=1e-5)
    b = C.ppo_batch(case["seed"])
    tr.step(b["input_ids"], b["should_take_action"], b["old_logprobs"], b["old_values"], b["old_advantages"], b["old_returns"])
    pg, hg = tr.last_grads
    grads = {"base." + k: pg[k] for k in sd_np}
    grads.update({"head.dense.kernel": hg["kernel"], "head.dense.bias": hg["bias"]})
    params = {"base." + k: v for k, v in sd_np.items()}
    params.update({"head." + k: v for k, v in vh.items()})
    _check_directions(case["name"], grads, params)
    case = C.MC_CASE
    sd_np, qh = C.state_dict(110 + case["seed"]), C.flat_head(C.mlp_head(120 + case["seed"], C.CFG["vocab"]))
    base = GPT2F32({k: t(v) for k, v in sd_np.items()}, C.CFG["n_head"], device=dev)
    trm = mc.GPT2MCTrain(base, MLPHeadF32({k: t(v) for k, v in qh.items()}, dev), C.PAD, dict(cql_weight=case["cql_weight"]), lr=1e-4)
    b = C.mc_batch(case["seed"])
    trm.step(b["input_ids"], b["should_take_action"], b["returns"])
    bg, qg = trm.last_grads
    grads = {"base." + k: bg[k] for k in sd_np}
    grads.update({"head." + k: qg[k] for k in qh})
    params = {"base." + k: v for k, v in sd_np.items()}
    params.update({"head." + k: v for k, v in qh.items()})
    _check_directions(case["name"], grads, params)
