"""Pins oracle/rl.py against the reference-produced vectors in tests/golden/rl_helpers.json and checks the
JAX-only restatements for internal consistency (they are 'parity unpinned', see oracle/rl.py header)."""
import numpy as np
import torch

from conftest import load_golden
from oracle import rl

G = load_golden("rl_helpers.json")


def test_gae_matches_reference_numpy():
    for c in G["gae"]:
        dt = np.float32 if c["out_dtype"] == "float32" else np.float64
        adv, ret = rl.gae(np.array(c["values"], dtype=dt), np.array(c["next_values"], dtype=dt),
                          np.array(c["rewards"], dtype=dt), c["gamma"], c["lam"], dtype=dt)
        assert adv.dtype == dt
        np.testing.assert_array_equal(adv, np.array(c["advantages"], dtype=dt))
        np.testing.assert_array_equal(ret, np.array(c["returns"], dtype=dt))


def test_idxs():
    for c in G["idxs"]:
        a, s, n = rl.get_action_state_next_state_idxs(np.array(c["mask"], dtype=bool))
        assert a.tolist() == c["action"] and s.tolist() == c["state"] and n.tolist() == c["next_state"]


def test_kl_controller():
    for c in G["kl"]["adaptive"]:
        k = rl.AdaptiveKLController(c["init"], c["target"], c["horizon"])
        for st in c["seq"]:
            k.update(st["current"], st["n_steps"])
            assert float(k.value) == st["value"]


def test_chain_shaping():
    for ch in G["chains"]:
        d = rl.ilql_data_from_chain(ch["token_chain"])
        assert d == ch["ilql_data"]
        for ml, ref in ch["combined"].items():
            ml = None if ml == "None" else int(ml)
            if "error" in ref:
                try:
                    rl.combined_chain(ch["token_chain"], ml)
                    assert False
                except AssertionError as e:
                    assert str(e) == ref["error"]
            else:
                assert rl.combined_chain(ch["token_chain"], ml) == ref


def test_rtg_closed_form():
    r = np.random.RandomState(0).randn(17)
    for g in [1.0, 0.99, 0.5]:
        ref = np.array([sum(g ** (j - i) * r[j] for j in range(i, 17)) for i in range(17)])
        np.testing.assert_allclose(rl.get_rtg(r, g), ref, rtol=1e-12)


def test_whiten():
    x = np.random.RandomState(1).randn(100) * 3 + 2
    w = rl.whiten(x)
    assert abs(w.mean()) < 1e-12 and abs(w.var() - 1) < 1e-6
    np.testing.assert_allclose(rl.whiten(x, shift_mean=False), w + x.mean())


def test_ilql_loss_selection_equals_per_row_next_action():
    # the flat k-th-True pairing (ilql/base_interface.py:55-74) == per-row "next action position" pairing
    torch.manual_seed(0)
    B, T, V = 3, 9, 11
    sta = torch.tensor([[0, 1, 1, 0, 0, 1, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0], [1, 0, 1, 1, 0, 0, 0, 1]], dtype=torch.bool)
    am = torch.ones(B, T - 1)
    q1, q2, v, tq1, tq2, r = [torch.randn(B, T - 1, dtype=torch.float64) for _ in range(6)]
    vf = torch.randn(B, dtype=torch.float64)
    ql1, ql2 = torch.randn(B, T - 1, V, dtype=torch.float64), torch.randn(B, T - 1, V, dtype=torch.float64)
    ids = torch.randint(0, V, (B, T - 1))
    loss, logs = rl.ilql_loss(q1, q2, v, vf, tq1, tq2, ql1, ql2, ids, am, sta, r, gamma=0.99, tau=0.7, cql_weight=0.01)
    n = sta.sum().double()
    tot = 0.0
    for b in range(B):
        pos = torch.nonzero(sta[b])[:, 0].tolist()
        for j, p in enumerate(pos):
            vns = v[b, pos[j + 1]] if j + 1 < len(pos) else vf[b]
            tot += 0.5 * (q1[b, p] - (r[b, p] + 0.99 * vns)) ** 2
    torch.testing.assert_close(logs["losses"]["q1_loss"], tot / n)


def _episode_from_record(rec):
    """A fixture record (character tokenizer: id = ord) back into the interaction list a rollout returns."""
    from lmrl_gym_amd.environment import InteractionTransition, Text
    ia = rec["is_action"]
    cuts = [0] + [t for t in range(1, len(ia)) if ia[t] != ia[t - 1]] + [len(ia)]
    texts = [Text("".join(chr(c) for c in rec["tokens"][a:b]), bool(ia[a])) for a, b in zip(cuts[:-1], cuts[1:])]
    out = []
    for k in range(1, len(texts), 2):
        out.append(InteractionTransition(tuple(texts[:k]), tuple(texts[:k + 1]), tuple(texts[:k + 2]), float(rec["reward"][cuts[k + 1] - 1]),
                                         rec["done"] and k + 2 >= len(texts)))
    return out


def test_length_rule_equals_the_reference_loop():
    """tests/golden/ppo_truncation.json = outputs of the reference script's own drop-the-last-turns loop (train_ppo_gpt2.py:317-341, executed):
    the oracle's token-record form and the package's host function (`text_trajectory_chains_from_interactions`) reproduce them exactly."""
    import lmrl_gym_amd  # noqa: F401
    from lmrl_gym_amd.algorithms.ppo_inference import text_trajectory_chains_from_interactions
    from lmrl_gym_amd.environment import TokenTrajectory

    class CharTok:
        def encode(self, s):
            return [ord(c) for c in s]
    fx = load_golden("ppo_truncation.json")
    n_short = n_skip = 0
    for case in fx["cases"]:
        kept = [rl.truncate_turns_record(r, case["max_length"], case["gamma"]) for r in case["records"]]
        n_skip += sum(k is None for k in kept)
        n_short += sum(k is not None and len(k["tokens"]) < len(r["tokens"]) for k, r in zip(kept, case["records"]))
        kept = [k for k in kept if k is not None]
        assert len(kept) == len(case["kept"])
        for got, exp in zip(kept, case["kept"]):
            assert got["tokens"] == exp["tokens"] and got["is_action"] == exp["is_action"] and got["done"] == exp["done"]
            assert got["reward"] == exp["reward"]                                        # float32-exact: the fold runs in double, rounded once
        chains = text_trajectory_chains_from_interactions([_episode_from_record(r) for r in case["records"]], CharTok(), case["max_length"], case["gamma"])
        assert len(chains) == len(case["kept"])
        for ch, exp in zip(chains, case["kept"]):
            tt = TokenTrajectory.from_text_trajectory(ch.text_trajectory, CharTok())
            assert tt.tokens.tolist() == exp["tokens"] and [float(x) for x in tt.reward] == exp["reward"] and bool(tt.done) == exp["done"]
    assert n_short >= 20 and n_skip >= 20


def test_partially_observed_chains_equal_the_reference_loop():
    """tests/golden/ppo_po_chains.json = outputs of the partially observed Maze script's own rollout -> chain loop (llm_rl_scripts/maze/ppo/
    partially_observed_ppo_online.py:372-398, executed on MazeEnv(last_k)-shaped interaction lists): the package's host function
    (`text_trajectory_chains_partially_observed`, what `MazeRolloutEngine.ppo_records` builds on the device for last_k > 1) reproduces every token
    trajectory of every chain — the window's item texts joined by single spaces as one non-action text, then the action, reward on its last token."""
    import lmrl_gym_amd  # noqa: F401
    from lmrl_gym_amd.algorithms.ppo_inference import text_trajectory_chains_partially_observed
    from lmrl_gym_amd.environment import InteractionTransition, Text, TokenTrajectoryChain

    class CharTok:
        def encode(self, s):
            return [ord(c) for c in s]
    fx = load_golden("ppo_po_chains.json")
    n_tt = longest = 0
    for case in fx["cases"]:
        raw = []
        for ep in case["episodes"]:
            trs = []
            for tr in ep:
                pa = tuple(Text(t, bool(a)) for t, a in tr["post_action_history"])
                trs.append(InteractionTransition(pa[:-1], pa, (), tr["reward"], tr["done"]))
            raw.append(trs)
        chains = text_trajectory_chains_partially_observed(raw)
        assert len(chains) == len(case["chains"])
        for ch, exp in zip(chains, case["chains"]):
            tts = TokenTrajectoryChain.from_text_trajectory_chain(ch, CharTok()).to_list()
            assert len(tts) == len(exp)
            for tt, e in zip(tts, exp):
                assert tt.tokens.tolist() == e["tokens"] and tt.is_action.astype(int).tolist() == e["is_action"]
                assert [float(x) for x in tt.reward] == e["reward"] and bool(tt.done) == e["done"]
                n_tt += 1
                longest = max(longest, len(e["tokens"]))
    assert n_tt == 160 and longest > 200
