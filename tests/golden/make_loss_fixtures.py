"""Generate tests/golden/rl_losses.json by EXECUTING the reference's own, unmodified loss functions.

    python tests/golden/make_loss_fixtures.py          (build container only: reads /root/reference)

jax / optax are not installable here, so `jax.numpy`, `jax.nn`, `jax.lax` and `optax` are backed by the numpy restatements in
_jnp_shim.py (unit-tested in tests/test_jnp_shim.py); the functions below are imported from /root/reference and run as they are:
  ppo_loss_fn, whiten, PPOInference.token_logprobs_from_logits   LLM_RL/algorithms/ppo/base_interface.py:72-142,245-251,396-403
  get_query_indicators, ilql_loss                                LLM_RL/algorithms/ilql/base_interface.py:22-119
  mc_loss                                                        LLM_RL/algorithms/mc_returns/base_interface.py:19-60
  get_rtg                                                        LLM_RL/algorithms/mc_returns/data.py:10-14
  bc_loss                                                        LLM_RL/algorithms/bc/interface.py:28-43
  get_tensor_stats, unpad_array                                  LLM_RL/utils.py:12-38
Per case the file stores the loss, the complete log dict, and `N_DIRECTIONS` directional derivatives d loss / d inputs . direction
obtained by complex-step differentiation THROUGH the reference code (its jax.lax.stop_gradient calls drop the imaginary part), which
pins where the reference stops gradients — something forward values alone cannot.
Inputs are regenerated from the seeds in loss_cases.py; only outputs are stored.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

_ref_import.install(jnp_shim=True)
import _jnp_shim as S  # noqa: E402
import loss_cases as LC  # noqa: E402

from LLM_RL.algorithms.ppo.base_interface import ppo_loss_fn, whiten, PPOInference  # noqa: E402
from LLM_RL.algorithms.ilql.base_interface import ilql_loss, get_query_indicators  # noqa: E402
from LLM_RL.algorithms.mc_returns.base_interface import mc_loss  # noqa: E402
from LLM_RL.algorithms.mc_returns.data import get_rtg  # noqa: E402
from LLM_RL.algorithms.bc.interface import bc_loss  # noqa: E402
from LLM_RL.utils import get_tensor_stats, unpad_array  # noqa: E402

H = 1e-30


def flat(d, prefix=""):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(flat(v, prefix + k + "."))
        else:
            out[prefix + k] = float(np.asarray(v).real)
    return out


def run_with_derivatives(fn, inputs, diff_names, seed, **hyper):
    """fn(**inputs, **hyper) -> (loss, logs) in float32 (the reference's arithmetic), then directional derivatives."""
    S.complex_step(False)
    loss, logs = fn(**{k: S.asarray(v) for k, v in inputs.items()}, **hyper)
    out = dict(loss=float(loss), logs=flat(logs), dloss=[])
    S.complex_step(True)
    try:
        for k in range(LC.N_DIRECTIONS):
            pert = {}
            for name, v in inputs.items():
                if name in diff_names:
                    pert[name] = S.asarray(v.astype(np.float64) + 1j * H * LC.direction(seed, k, name, v.shape))
                else:
                    pert[name] = S.asarray(v)
            l, _ = fn(**pert, **hyper)
            out["dloss"].append(float(np.asarray(l).imag / H))
    finally:
        S.complex_step(False)
    return out


class _FakeModel:
    """bc_loss calls `model(input_ids=, attention_mask=, params=, dropout_rng=, train=).logits`: hand back the case's logits."""

    def __init__(self, logits):
        self._logits = logits

    def __call__(self, **kw):
        class O:
            pass
        o = O()
        o.logits = self._logits
        return o


def main():
    out = {"_meta": dict(generator="tests/golden/make_loss_fixtures.py", arithmetic="float32 numpy under the jax x64-off promotion rules",
                         derivative="complex step h=1e-30 through the reference code (stop_gradient -> real part)")}
    out["ppo_loss_fn"] = []
    for c in LC.PPO_CASES:
        hyper = {k: c[k] for k in ("cliprange_value", "cliprange", "value_loss_coef")}
        out["ppo_loss_fn"].append(dict(case=c, **run_with_derivatives(ppo_loss_fn, LC.ppo_inputs(c), LC.PPO_DIFF, c["seed"], **hyper)))
    out["ilql_loss"] = []
    for c in LC.ILQL_CASES:
        hyper = {k: c[k] for k in ("gamma", "tau", "cql_weight")}
        out["ilql_loss"].append(dict(case=c, **run_with_derivatives(ilql_loss, LC.ilql_inputs(c), LC.ILQL_DIFF, c["seed"], **hyper)))
    out["mc_loss"] = []
    for c in LC.MC_CASES:
        out["mc_loss"].append(dict(case=c, **run_with_derivatives(mc_loss, LC.mc_inputs(c), LC.MC_DIFF, c["seed"], cql_weight=c["cql_weight"])))
    out["bc_loss"] = []
    for c in LC.BC_CASES:
        inp = LC.bc_inputs(c)

        def fn(logits, input_ids, attention_mask, is_action, non_action_weight):
            return bc_loss(_FakeModel(logits), input_ids, attention_mask, is_action, None, None, False, non_action_weight=non_action_weight)
        out["bc_loss"].append(dict(case=c, **run_with_derivatives(fn, inp, ("logits",), c["seed"], non_action_weight=c["non_action_weight"])))
    out["whiten"] = [dict(case=c, out=[float(x) for x in whiten(S.asarray(LC.whiten_input(c)), shift_mean=c["shift_mean"])]) for c in LC.WHITEN_CASES]
    out["get_rtg"] = [dict(case=c, out=[float(x) for x in get_rtg(S.asarray(LC.rtg_input(c)), c["gamma"])]) for c in LC.RTG_CASES]
    out["token_logprobs_from_logits"] = []
    for c in LC.LOGPROB_CASES:
        inp = LC.logprob_inputs(c)
        lp = PPOInference.token_logprobs_from_logits(S.asarray(inp["logits"]), S.asarray(inp["input_ids"]))
        out["token_logprobs_from_logits"].append(dict(case=c, out=np.asarray(lp).astype(float).tolist()))
    # get_query_indicators on the masks of the ILQL cases: stored sparsely as the column of the 1 in each row (-1 = zero row)
    out["get_query_indicators"] = []
    for c in LC.ILQL_CASES:
        m = LC.ilql_inputs(c)["should_take_action"].reshape(-1)
        ind = np.asarray(get_query_indicators(S.asarray(m)))
        assert ind.shape == (m.size, m.size) and set(np.unique(ind)) <= {0.0, 1.0} and (ind.sum(1) <= 1).all()
        out["get_query_indicators"].append(dict(case=dict(seed=c["seed"]), cols=[int(r.argmax()) if r.sum() else -1 for r in ind]))
    # get_tensor_stats incl. the all-masked corner, unpad_array
    rng = np.random.RandomState(71)
    xs = rng.randn(4, 9).astype(np.float32); mk = (rng.rand(4, 9) < 0.5)
    st = get_tensor_stats(S.asarray(xs), S.asarray(mk.astype(np.float32)), float(mk.sum()))
    out["get_tensor_stats"] = [dict(case=dict(seed=71), out={k: float(v) for k, v in st.items()})]
    with np.errstate(all="ignore"):
        st0 = get_tensor_stats(S.asarray(xs), S.asarray(np.zeros_like(xs)), 1.0)
    out["get_tensor_stats"].append(dict(case=dict(seed=71, all_masked=True), out={k: (None if np.isnan(float(v)) else float(v)) for k, v in st0.items()}))
    m1 = np.array([1, 1, 1, 0, 0, 1], dtype=np.int32)
    out["unpad_array"] = [dict(mask=m1.tolist(), n_kept=int(len(unpad_array(np.arange(6), m1)))),
                          dict(mask=[1, 1, 1], n_kept=int(len(unpad_array(np.arange(3), np.ones(3, dtype=np.int32)))))]
    path = os.path.join(HERE, "rl_losses.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(f"wrote rl_losses.json: {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
