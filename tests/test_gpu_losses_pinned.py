"""GPU tier: the HIP loss kernels against tests/golden/rl_losses.json — outputs (loss, every log entry, directional derivatives)
of the REFERENCE's own unmodified loss functions run under the numpy jax/optax shim (tests/golden/make_loss_fixtures.py).
Tolerances: loss / logs 3e-5 relative (fp32 kernels with fp64 partial sums vs the reference's float32 arithmetic), derivatives 1e-4."""
import sys

import numpy as np
import pytest

import lmrl_gym_amd  # noqa: F401
from conftest import GOLDEN, load_golden

sys.path.insert(0, GOLDEN)
import loss_cases as LC  # noqa: E402

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
G = load_golden("rl_losses.json")


@pytest.fixture(scope="module")
def dev():
    from lmrl_gym_amd import _lib
    return _lib.require_gpu()


def _flat(d, prefix=""):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(_flat(v, prefix + k + "."))
        else:
            out[prefix + k] = float(v)
    return out


def _check_values(e, loss, logs):
    assert abs(loss - e["loss"]) <= 3e-5 * max(1.0, abs(e["loss"])), (e["case"], loss, e["loss"])
    got = _flat(logs)
    assert set(got) == set(e["logs"])
    for k, ref in e["logs"].items():
        assert abs(got[k] - ref) <= 3e-5 * max(1.0, abs(ref)), (e["case"], k, got[k], ref)


def _ddot(e, grads):
    for kdir, ref in enumerate(e["dloss"]):
        tot = sum(float((g.double().cpu().numpy() * LC.direction(e["case"]["seed"], kdir, name, tuple(g.shape))).sum()) for name, g in grads.items())
        assert abs(tot - ref) <= 1e-4 * max(1.0, abs(ref)), (e["case"], kdir, tot, ref)


def test_ppo_loss_kernel_vs_reference_golden(dev):
    from lmrl_gym_amd.algorithms import ppo
    f = lambda x, dt=np.float32: torch.from_numpy(np.ascontiguousarray(x.astype(dt))).to(dev)
    for e in G["ppo_loss_fn"]:
        c, i = e["case"], LC.ppo_inputs(e["case"])
        loss, logs, dlp, dv = ppo.ppo_loss_device(f(i["attention_mask"]), f(i["logprobs"]), f(i["values"]), f(i["should_take_action"], np.uint8),
                                                  f(i["old_logprobs"]), f(i["old_values"]), f(i["old_advantages"]), f(i["old_returns"]),
                                                  cliprange_value=c["cliprange_value"], cliprange=c["cliprange"], value_loss_coef=c["value_loss_coef"])
        _check_values(e, loss, logs)
        _ddot(e, dict(logprobs=dlp, values=dv))


def test_ilql_and_mc_loss_kernels_vs_reference_golden(dev):
    from lmrl_gym_amd.algorithms import ilql, mc_returns as mc
    from lmrl_gym_amd.train import ops
    f = lambda x, dt=np.float32: torch.from_numpy(np.ascontiguousarray(x.astype(dt))).to(dev)

    def ce_of(logits, tok):
        B, T1, V = logits.shape
        lgd = f(logits.reshape(B * T1, V)); lp = torch.empty(B * T1, device=dev); lse = torch.empty(B * T1, device=dev)
        ops.lse_gather(lgd, V, V, f(tok.reshape(-1), np.int32), B * T1, logprob=lp, lse=lse)
        return lgd, lse, (-lp).view(B, T1).contiguous()

    for e in G["ilql_loss"]:
        c, i = e["case"], LC.ilql_inputs(e["case"])
        B, T1, V = i["q1_logits"].shape
        lg1, lse1, ce1 = ce_of(i["q1_logits"], i["token_ids"]); lg2, lse2, ce2 = ce_of(i["q2_logits"], i["token_ids"])
        loss, logs, dq1, dq2, dv, coef = ilql.ilql_loss_device(f(i["q1"]), f(i["q2"]), f(i["v"]), f(i["v_final"]), f(i["target_q1"]), f(i["target_q2"]),
                                                               ce1, ce2, f(i["attention_mask"]), f(i["should_take_action"], np.uint8), f(i["rewards"]),
                                                               gamma=c["gamma"], tau=c["tau"], cql_weight=c["cql_weight"])
        _check_values(e, loss, logs)
        # d loss / d logits = coef * d CE / d logits (ce_bwd overwrites the logits buffer with the gradient)
        tok = f(i["token_ids"].reshape(-1), np.int32); zero = torch.zeros(B * T1, device=dev)
        ops.ce_bwd(lg1, V, V, lse1, tok, coef.reshape(-1).contiguous(), zero, B * T1)
        ops.ce_bwd(lg2, V, V, lse2, tok, coef.reshape(-1).contiguous(), zero, B * T1)
        _ddot(e, dict(q1=dq1, q2=dq2, v=dv, q1_logits=lg1.view(B, T1, V), q2_logits=lg2.view(B, T1, V)))   # targets / v_final / rewards: no gradient
        # the numpy face with the reference signature
        loss2, logs2 = ilql.ilql_loss(i["q1"], i["q2"], i["v"], i["v_final"], i["target_q1"], i["target_q2"], i["q1_logits"], i["q2_logits"], i["token_ids"],
                                      i["attention_mask"], i["should_take_action"], i["rewards"], gamma=c["gamma"], tau=c["tau"], cql_weight=c["cql_weight"])
        _check_values(e, loss2, logs2)
    for e in G["mc_loss"]:
        c, i = e["case"], LC.mc_inputs(e["case"])
        B, T1, V = i["q_logits"].shape
        lg, lse, ce = ce_of(i["q_logits"], i["token_ids"])
        loss, logs, dq, coef = mc.mc_loss_device(f(i["q"]), ce, f(i["attention_mask"]), f(i["should_take_action"], np.uint8), f(i["returns"]), cql_weight=c["cql_weight"])
        _check_values(e, loss, logs)
        ops.ce_bwd(lg, V, V, lse, f(i["token_ids"].reshape(-1), np.int32), coef.reshape(-1).contiguous(), torch.zeros(B * T1, device=dev), B * T1)
        _ddot(e, dict(q=dq, q_logits=lg.view(B, T1, V)))


def test_whiten_rtg_logprobs_vs_reference_golden(dev):
    from lmrl_gym_amd.algorithms import mc_returns as mc, ppo
    from lmrl_gym_amd.train import ops
    for e in G["whiten"]:
        c = e["case"]
        np.testing.assert_allclose(ppo.whiten(LC.whiten_input(c), shift_mean=c["shift_mean"]), np.array(e["out"]), rtol=3e-5, atol=3e-5)
    for e in G["get_rtg"]:
        c = e["case"]
        ref = np.array(e["out"])
        np.testing.assert_allclose(mc.get_rtg(LC.rtg_input(c), c["gamma"]), ref, rtol=3e-5, atol=3e-5 * max(1.0, float(np.abs(ref).max())))
    for e in G["token_logprobs_from_logits"]:
        inp = LC.logprob_inputs(e["case"])
        B, T, V = inp["logits"].shape
        lg = torch.from_numpy(inp["logits"][:, :-1].reshape(-1, V).copy()).to(dev)
        tgt = torch.from_numpy(inp["input_ids"][:, 1:].reshape(-1).copy()).to(dev)
        lp = torch.empty(B * (T - 1), device=dev)
        ops.lse_gather(lg, V, V, tgt, B * (T - 1), logprob=lp)
        np.testing.assert_allclose(lp.cpu().numpy().reshape(B, T - 1), np.array(e["out"]), rtol=1e-5, atol=1e-5)
