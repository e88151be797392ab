"""Pins oracle/rl.py's loss restatements to tests/golden/rl_losses.json — outputs of the REFERENCE's own unmodified
`ppo_loss_fn`, `ilql_loss`, `get_query_indicators`, `mc_loss`, `get_rtg`, `whiten`, `bc_loss`, `token_logprobs_from_logits`,
`get_tensor_stats` executed under the numpy `jax.numpy`/`optax` shim (tests/golden/make_loss_fixtures.py).

Tolerances: the golden values are float32 arithmetic (as the reference computes), the oracle is float64 -> loss / logs 2e-6
relative (5e-6 for std entries: float32 cancellation in the reference itself); directional derivatives (complex-step through the
reference code incl. its stop_gradient calls) vs float64 autograd of the oracle: 1e-9 relative.
"""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden
from oracle import rl

sys.path.insert(0, GOLDEN)
import loss_cases as LC  # noqa: E402

G = load_golden("rl_losses.json")
F64 = torch.float64


def _flat(d, prefix=""):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(_flat(v, prefix + k + "."))
        else:
            out[prefix + k] = float(v.detach()) if hasattr(v, "detach") else float(v)
    return out


def _tensors(inputs, diff):
    t = {}
    for k, v in inputs.items():
        if v.dtype.kind == "f":
            t[k] = torch.from_numpy(v.astype(np.float64)).requires_grad_(k in diff)
        else:
            t[k] = torch.from_numpy(v)
    return t


def _check(entry, loss, logs, tensors, diff):
    c = entry["case"]
    lv = float(loss.detach())
    assert abs(lv - entry["loss"]) <= 2e-6 * max(1.0, abs(entry["loss"])), (c, lv, entry["loss"])
    got = _flat(logs)
    assert set(got) == set(entry["logs"]), (sorted(got), sorted(entry["logs"]))
    for k, ref in entry["logs"].items():
        tol = (5e-6 if k.endswith("std") else 2e-6) * max(1.0, abs(ref))
        assert abs(got[k] - ref) <= tol, (c, k, got[k], ref)
    loss.backward()
    for kdir, ref in enumerate(entry["dloss"]):
        tot = 0.0
        for name in diff:
            g = tensors[name].grad
            if g is not None:
                tot += float((g.numpy() * LC.direction(c["seed"], kdir, name, tuple(g.shape))).sum())
        assert abs(tot - ref) <= 1e-9 * max(1.0, abs(ref)), (c, kdir, tot, ref)


def test_ppo_loss_pinned_to_reference():
    assert len(G["ppo_loss_fn"]) == len(LC.PPO_CASES)
    for e in G["ppo_loss_fn"]:
        c = e["case"]
        t = _tensors(LC.ppo_inputs(c), LC.PPO_DIFF)
        loss, logs = rl.ppo_loss(t["attention_mask"], t["logprobs"], t["values"], t["should_take_action"], t["old_logprobs"], t["old_values"],
                                 t["old_advantages"], t["old_returns"], cliprange_value=c["cliprange_value"], cliprange=c["cliprange"],
                                 value_loss_coef=c["value_loss_coef"])
        _check(e, loss, logs, t, LC.PPO_DIFF)


def test_ilql_loss_pinned_to_reference():
    assert len(G["ilql_loss"]) == len(LC.ILQL_CASES)
    for e in G["ilql_loss"]:
        c = e["case"]
        t = _tensors(LC.ilql_inputs(c), LC.ILQL_DIFF)
        loss, logs = rl.ilql_loss(t["q1"], t["q2"], t["v"], t["v_final"], t["target_q1"], t["target_q2"], t["q1_logits"], t["q2_logits"],
                                  t["token_ids"], t["attention_mask"], t["should_take_action"], t["rewards"], gamma=c["gamma"], tau=c["tau"],
                                  cql_weight=c["cql_weight"])
        _check(e, loss, logs, t, LC.ILQL_DIFF)
        # what the reference's stop_gradient placement implies: no gradient reaches the targets, v_final or the rewards
        for name in ("target_q1", "target_q2", "v_final", "rewards"):
            assert t[name].grad is None or float(t[name].grad.abs().max()) == 0.0, name


def test_query_indicators_pinned_to_reference():
    for e, c in zip(G["get_query_indicators"], LC.ILQL_CASES):
        m = LC.ilql_inputs(c)["should_take_action"].reshape(-1)
        ind = rl.get_query_indicators(torch.from_numpy(m)).numpy()
        cols = [int(r.argmax()) if r.sum() else -1 for r in ind]
        assert cols == e["cols"] and set(np.unique(ind)) <= {0.0, 1.0}


def test_mc_loss_pinned_to_reference():
    for e in G["mc_loss"]:
        c = e["case"]
        t = _tensors(LC.mc_inputs(c), LC.MC_DIFF)
        loss, logs = rl.mc_loss(t["q"], t["q_logits"], t["token_ids"], t["attention_mask"], t["should_take_action"], t["returns"],
                                cql_weight=c["cql_weight"])
        _check(e, loss, logs, t, LC.MC_DIFF)


def test_bc_loss_pinned_to_reference():
    for e in G["bc_loss"]:
        c = e["case"]
        t = _tensors(LC.bc_inputs(c), ("logits",))
        loss = rl.bc_loss(t["logits"], t["input_ids"], t["attention_mask"], t["is_action"], non_action_weight=c["non_action_weight"])
        _check(e, loss, {"loss": loss}, t, ("logits",))


def test_whiten_rtg_logprobs_stats_pinned_to_reference():
    for e in G["whiten"]:
        c = e["case"]
        np.testing.assert_allclose(rl.whiten(LC.whiten_input(c), shift_mean=c["shift_mean"]), np.array(e["out"]), rtol=3e-5, atol=3e-5)
    for e in G["get_rtg"]:
        c = e["case"]
        ref = np.array(e["out"])
        np.testing.assert_allclose(rl.get_rtg(LC.rtg_input(c), c["gamma"]), ref, rtol=3e-5, atol=3e-5 * max(1.0, float(np.abs(ref).max())))
        # the float32 path the reference takes (cumprod ratio in float32): closer still
        np.testing.assert_allclose(rl.get_rtg(LC.rtg_input(c), c["gamma"], dtype=np.float32), ref, rtol=2e-6, atol=2e-6 * max(1.0, float(np.abs(ref).max())))
    for e in G["token_logprobs_from_logits"]:
        inp = LC.logprob_inputs(e["case"])
        got = rl.token_logprobs_from_logits(torch.from_numpy(inp["logits"]), torch.from_numpy(inp["input_ids"])).numpy()
        np.testing.assert_allclose(got, np.array(e["out"]), rtol=2e-6, atol=2e-6)
    rng = np.random.RandomState(71)
    xs = rng.randn(4, 9).astype(np.float32); mk = rng.rand(4, 9) < 0.5
    st = rl.tensor_stats(torch.from_numpy(xs), torch.from_numpy(mk), float(mk.sum()))
    for k, ref in G["get_tensor_stats"][0]["out"].items():
        assert abs(float(st[k]) - ref) <= 3e-6 * max(1.0, abs(ref)), k
    st0 = rl.tensor_stats(torch.from_numpy(xs), torch.zeros(4, 9), 1.0)
    ref0 = G["get_tensor_stats"][1]["out"]
    assert float(st0["mean"]) == ref0["mean"] == 0.0 and float(st0["min"]) == ref0["min"] == float("inf")
    assert float(st0["max"]) == ref0["max"] == float("-inf") and ref0["std"] is None and np.isnan(float(st0["std"]))
    assert [u["n_kept"] for u in G["unpad_array"]] == [3, 3]
