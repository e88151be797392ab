"""GPU tier: the device-resident online-PPO data path (VERDICT r04 "next" #1) — rollout records -> PPO data -> PPO batches -> train step with no
host text, re-tokenisation or materialised logits in between (lmrl-gym_amd/algorithms/ppo_device.py, csrc/ppo_data.hip).

  * the device pipeline on the reference-function fixture's chains (multi-trajectory chains, bootstrap x (1 - done), KL list, whitening, chunk
    unrolling) == tests/golden/rl_steps.json = the output of the reference's OWN `get_ppo_data_from_token_trajectory_chain`
    (ppo/base_interface.py:464-669), same tolerances as the host-array form's test;
  * on real Wordle episodes of the device engine: device path == the host-array form (`GPT2PPOInference.get_ppo_data_from_token_trajectory_chain`
    + `PPODataset.from_ppo_data_list`) on the same episodes: identical ids / masks / KL-list length, log-probs and values 1e-5, returns 1e-5,
    whitened advantages 2e-5, KL terms 5e-6 (float32 cancellation in exp(lr) - 1 - lr);
  * a `DevicePPODataset.batch(...)` fed to `GPT2PPOTrain.step` == the same rows as numpy arrays: loss, every log, every gradient, the updated
    parameters — bit for bit (the same kernels on the same operands);
  * the small kernels against their numpy definitions; `GPT2Engine.load_params` == a freshly built engine; a whole round (`ppo_rollouts`).
"""
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


@pytest.fixture(scope="module")
def dev():
    from lmrl_gym_amd import _lib
    return _lib.require_gpu()


def _rows(ds_host, n_tok):
    """DevicePPODataset.to_host() -> per-trajectory PPOData-like dicts (cut at the trajectory's length)."""
    out = []
    for k, n in enumerate(n_tok):
        n = int(n)
        out.append(dict(input_ids=ds_host.input_ids[k, :n], should_take_action=ds_host.should_take_action[k, :n - 1],
                        old_logprobs=ds_host.old_logprobs[k, :n - 1], old_values=ds_host.old_values[k, :n - 1],
                        old_advantages=ds_host.old_advantages[k, :n - 1], old_returns=ds_host.old_returns[k, :n - 1]))
    return out


def test_device_pipeline_equals_reference_function_fixture(dev):
    from lmrl_gym_amd import environment as E
    from lmrl_gym_amd.algorithms.ppo_device import PPORecords, ppo_data_from_records
    from lmrl_gym_amd.algorithms.ppo_inference import GPT2PPOInference
    from lmrl_gym_amd.train.gpt2_f32 import GPT2F32, LinearHeadF32
    case = C.PPO_DATA_CASE
    fx = load_golden("rl_steps.json")[case["name"]]
    t = lambda a: torch.from_numpy(np.asarray(a))
    init_np = C.state_dict(230 + case["seed"])
    vh = C.flat_head(C.linear_head(240 + case["seed"]))
    pol = GPT2F32({k: t(v) for k, v in C.perturbed(init_np, 220 + case["seed"]).items()}, C.CFG["n_head"], device=dev)
    init = GPT2F32({k: t(v) for k, v in init_np.items()}, C.CFG["n_head"], device=dev)
    inf = GPT2PPOInference(pol, LinearHeadF32(dict(kernel=t(vh["dense.kernel"]), bias=t(vh["dense.bias"])), dev), C.PAD, initial_policy=init)
    chains, n_tok = [], []
    for ch in C.ppo_chains(case["seed"]):
        node = None
        for tt in reversed(ch):
            node = E.TokenTrajectoryChain(E.TokenTrajectory(tt["tokens"], tt["is_action"], tt["reward"], np.asarray(tt["done"])), node)
        chains.append(node)
        n_tok += [len(tt["tokens"]) for tt in ch]
    rec = PPORecords.from_token_trajectory_chains(chains, device=dev)
    assert rec.n == len(n_tok) == 7 and rec.n_chains == 4                      # chains of 1, 2, 3, 1 trajectories
    for bsize in (case["bsize"], 64):                                          # the reference's forward batching and one forward for all
        ds, kls = ppo_data_from_records(inf, rec, gamma=case["gamma"], lam=case["lam"], kl_weight=case["kl_weight"], bsize=bsize)
        np.testing.assert_allclose(kls.cpu().numpy(), fx["kls"], rtol=2e-3, atol=2e-5)
        got = _rows(ds.to_host(), n_tok)
        assert len(got) == len(fx["datas"])
        for d, e in zip(got, fx["datas"]):
            assert d["input_ids"].tolist() == e["input_ids"] and [bool(x) for x in d["should_take_action"]] == e["should_take_action"]
            np.testing.assert_allclose(d["old_logprobs"], e["old_logprobs"], rtol=1e-4, atol=1e-4)
            np.testing.assert_allclose(d["old_values"], e["old_values"], rtol=1e-4, atol=1e-4)
            np.testing.assert_allclose(d["old_returns"], e["old_returns"], rtol=2e-4, atol=2e-4)
            np.testing.assert_allclose(d["old_advantages"], e["old_advantages"], rtol=2e-3, atol=2e-3)
    # everything past a trajectory's length is padding: pad ids, zeros, False
    h = ds.to_host()
    for k, n in enumerate(n_tok):
        assert (h.input_ids[k, n:] == C.PAD).all() and not h.should_take_action[k, n - 1:].any()
        for name in ("old_logprobs", "old_values", "old_advantages", "old_returns"):
            assert (getattr(h, name)[k, n - 1:] == 0).all()


@pytest.fixture(scope="module")
def wordle_setup(dev):
    """A small policy (2 layers, d = 128, full GPT-2 vocabulary) with its fp32 trainer twin, an initial policy a few updates away, one lock-step
    episode of 96 envs on the device engine (moderately steered: real sampling, valid and invalid words, early and late finishes)."""
    from lmrl_gym_amd.algorithms.ppo_inference import GPT2PPOInference
    from lmrl_gym_amd.envs import wordle as W
    from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine, init_hf_style_state_dict
    from lmrl_gym_amd.rollout import WordleRolloutEngine
    from lmrl_gym_amd.train.gpt2_f32 import GPT2F32, LinearHeadF32
    cfg = GPT2Config(2, 2, 128, 512, 50257, 128)
    sd = init_hf_style_state_dict(cfg, seed=5)
    sd["wte.weight"] = sd["wte.weight"] * 8
    g = torch.Generator().manual_seed(11)
    sd_pol = {k: v + 0.02 * v.abs().mean().clamp_min(1e-3) * torch.randn(v.shape, generator=g) for k, v in sd.items()}
    eng = GPT2Engine(cfg, sd_pol, dev)
    vocab = W.Vocabulary.builtin("wordle_official_400.txt")
    B = 96
    ro = WordleRolloutEngine(eng, vocab, B, max_new_tokens=6, bad_word_reward=-10.0)
    rng = np.random.RandomState(5)
    packed = np.array([W.pack_guess(w) for w in vocab.all_vocab], dtype=np.uint32)
    guesses = torch.from_numpy(packed[rng.randint(0, len(packed), size=(6, B))].view(np.int32)).to(dev)
    # the tied embedding (x 8) makes this random model repeat its last token with a large logit; +24 on the scripted token wins most but not all
    # draws against that: 41 .. 76-token episodes, valid and invalid words, 1 .. 7-token actions
    ro.run_episode(np.arange(B, dtype=np.uint64) + 40, temperature=1.0, sample_seed=3, scripted_guesses=guesses, steer_strength=24.0)
    torch.cuda.synchronize()
    # the engine's pad id is the first id AFTER the policy's vocabulary (the reference's added `<|pad|>`, whose logit the model forces to -inf,
    # ppo/gpt2/interface.py:330): it cannot be sampled, so the record needs no clean-up before the reference-shaped host path reads it
    assert ro.tokens.pad == cfg.vocab and not bool((ro.traj["tokens"] == ro.tokens.pad).any())
    mk = lambda matmul: (GPT2F32(sd_pol, cfg.n_head, device=dev, matmul=matmul), GPT2F32(sd, cfg.n_head, device=dev, matmul=matmul))
    head = lambda: LinearHeadF32(dict(kernel=torch.randn(cfg.d_model, 1, generator=torch.Generator().manual_seed(2)) * 0.05, bias=torch.tensor([-0.3])), dev)
    pol, init = mk("f32")
    inf = GPT2PPOInference(pol, head(), ro.tokens.pad, initial_policy=init)
    yield dict(cfg=cfg, sd=sd, sd_pol=sd_pol, eng=eng, ro=ro, inf=inf, mk=mk, head=head, B=B, vocab=vocab)
    ro.close()


def _host_form(ro, inf, max_length, **kw):
    """The host-array form on the same episodes: TokenTrajectoryChains from the record -> the reference-shaped function -> blocked dataset."""
    from lmrl_gym_amd import environment as E
    from lmrl_gym_amd.algorithms import ppo
    from lmrl_gym_amd.algorithms.common import BlockingStrategy, Padding, Truncation
    chains = [E.TokenTrajectoryChain(E.TokenTrajectory(tok, ia, rw, np.asarray(dn)), None) for tok, ia, rw, dn in ro.token_trajectories()]
    datas, kls = inf.get_ppo_data_from_token_trajectory_chain(chains, bsize=32, max_length=max_length, **kw)

    class _Tok:
        pad_token_id = inf.pad
    return ppo.PPODataset.from_ppo_data_list(datas, _Tok, BlockingStrategy(Padding.RIGHT, Truncation.RIGHT, max_length)), kls, chains


def test_device_path_equals_host_form_on_engine_episodes(wordle_setup):
    s = wordle_setup
    ro, inf = s["ro"], s["inf"]
    kw = dict(gamma=0.97, lam=0.9, kl_weight=0.05)
    max_length = ro.cap + 1
    host, kls_h, chains = _host_form(ro, inf, max_length, **kw)
    ds, kls_d = ro.ppo_data(inf, max_length=max_length, bsize=40, **kw)
    dev_h = ds.to_host()
    n_tok = ro.traj["n_tok"].cpu().numpy()
    assert n_tok.min() >= 10 and len(set(n_tok.tolist())) > 3                       # ragged episodes
    assert dev_h.input_ids.shape == host.input_ids.shape == (s["B"], max_length)
    assert (dev_h.input_ids == host.input_ids).all() and (dev_h.should_take_action == host.should_take_action).all()
    assert host.should_take_action.sum() > 12 * s["B"]
    np.testing.assert_allclose(dev_h.old_logprobs, host.old_logprobs, rtol=0, atol=1e-5)
    np.testing.assert_allclose(dev_h.old_values, host.old_values, rtol=0, atol=1e-5)
    np.testing.assert_allclose(dev_h.old_returns, host.old_returns, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(dev_h.old_advantages, host.old_advantages, rtol=2e-5, atol=2e-5)
    kd = kls_d.cpu().numpy()
    assert kd.shape == kls_h.shape == (int(host.should_take_action.sum()),)
    np.testing.assert_allclose(kd, kls_h, rtol=1e-5, atol=5e-6)          # exp(lr) - 1 - lr in float32: the cancellation leaves a few 1e-7 x |1 + lr| absolute
    assert kd.min() >= 0 and kd.mean() > 1e-5                                        # a real KL between two different policies
    # whitening off, and a wider dataset than the blocking width: the same numbers in a different frame
    ds2, _ = ro.ppo_data(inf, max_length=max_length, pad_to=160, use_advantage_whitening=False, **kw)
    host2, _, _ = _host_form(ro, inf, max_length, use_advantage_whitening=False, **kw)
    h2 = ds2.to_host()
    assert h2.input_ids.shape == (s["B"], 160) and (h2.input_ids[:, max_length:] == inf.pad).all()
    np.testing.assert_allclose(h2.old_advantages[:, :max_length - 1], host2.old_advantages, rtol=1e-5, atol=1e-5)
    # max_length within reach of an episode: the script's drop-the-last-turns rule (train_ppo_gpt2.py:323-341) applies ON THE DEVICE — against the
    # oracle's form of the rule on the same records (itself pinned to the reference's executed loop, tests/test_oracle_rl.py) + the host-array form
    from lmrl_gym_amd import environment as E
    from lmrl_gym_amd.algorithms import ppo
    from lmrl_gym_amd.algorithms.common import BlockingStrategy, Padding, Truncation
    from oracle import rl as ORL
    ml = 60
    assert (n_tok >= ml).sum() > 10 and (n_tok < ml).sum() > 10
    tm = {}
    ds3, kls3 = ro.ppo_data(inf, max_length=ml, timings=tm, **kw)
    kept = [ORL.truncate_turns_record(dict(tokens=tok, is_action=ia, reward=rw, done=dn), ml, kw["gamma"]) for tok, ia, rw, dn in ro.token_trajectories()]
    assert tm["episodes_shortened"] == sum(k is not None and len(k["tokens"]) < n for k, n in zip(kept, n_tok)) > 10
    assert tm["episodes_skipped"] == sum(k is None for k in kept)
    kept = [k for k in kept if k is not None]
    chains3 = [E.TokenTrajectoryChain(E.TokenTrajectory(np.asarray(k["tokens"], np.int32), np.asarray(k["is_action"], bool), np.asarray(k["reward"], np.float32),
                                                        np.asarray(k["done"])), None) for k in kept]
    datas3, kls3_h = inf.get_ppo_data_from_token_trajectory_chain(chains3, bsize=32, max_length=ml, **kw)

    class _Tok:
        pad_token_id = inf.pad
    host3 = ppo.PPODataset.from_ppo_data_list(datas3, _Tok, BlockingStrategy(Padding.RIGHT, Truncation.RIGHT, ml))
    h3 = ds3.to_host()
    assert h3.input_ids.shape == host3.input_ids.shape == (len(kept), ml)
    assert (h3.input_ids == host3.input_ids).all() and (h3.should_take_action == host3.should_take_action).all()
    np.testing.assert_allclose(h3.old_returns, host3.old_returns, rtol=1e-5, atol=1e-5)      # folded, discounted rewards + bootstrap values (done cleared)
    np.testing.assert_allclose(h3.old_advantages, host3.old_advantages, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(kls3.cpu().numpy(), kls3_h, rtol=1e-5, atol=5e-6)
    # the engine's own record is untouched by the rule (reward / n_tok / done are copied)
    assert np.array_equal(ro.traj["n_tok"].cpu().numpy(), n_tok)
    # batches cut to the longest episode: the same rows, fewer padded columns
    w = ds.trimmed_width()
    assert ds.longest == int(n_tok.max()) and w == 128 and ds2.trimmed_width() == 128
    full, cut = ds2.batch(np.arange(8)), ds2.batch(np.arange(8), width=w)
    for k in full:
        assert torch.equal(full[k][:, :cut[k].shape[1]], cut[k]) and cut[k].shape[1] == (w if k in ("input_ids", "attention_mask", "position_ids") else w - 1) and cut[k].is_contiguous()


def test_device_path_in_the_bf16_matmul_mode(wordle_setup):
    """`bf16_activations` (train_ppo_gpt2.py:70): log-sum-exp and the target logit straight from the LM-head GEMM's accumulators, no logits stored.
    Against the fp32 path on the same weights: log-probs within bf16 product accuracy, same masks, KL terms of the same size."""
    from lmrl_gym_amd.algorithms.ppo_inference import GPT2PPOInference
    s = wordle_setup
    ro = s["ro"]
    pol, init = s["mk"]("bf16")
    inf_b = GPT2PPOInference(pol, s["head"](), ro.tokens.pad, initial_policy=init)
    kw = dict(gamma=1.0, lam=0.95, kl_weight=0.001, max_length=ro.cap + 1)
    a, kls_a = ro.ppo_data(s["inf"], **kw)
    b, kls_b = ro.ppo_data(inf_b, lm_head_rows=1000, **kw)                            # several row chunks, the last one ragged
    ha, hb = a.to_host(), b.to_host()
    assert (ha.input_ids == hb.input_ids).all() and (ha.should_take_action == hb.should_take_action).all()
    # (this model's logits reach |x| ~ 30: bf16 products and the log-sum-exp of bf16-rounded logits move a log-prob by up to |x| 2^-8)
    np.testing.assert_allclose(hb.old_logprobs, ha.old_logprobs, rtol=0, atol=0.3)
    np.testing.assert_allclose(hb.old_values, ha.old_values, rtol=0, atol=5e-2)
    assert np.abs(hb.old_logprobs - ha.old_logprobs)[ha.should_take_action].mean() < 0.1
    assert abs(float(kls_b.mean()) - float(kls_a.mean())) < 0.3 * float(kls_a.mean()) + 1e-3
    # what matters to PPO: old_logprobs come from the SAME arithmetic as the train step's log-probs (CE out of the GEMM accumulators in both), so
    # the first step after a data build sees ratio == 1 / approx_kl == 0 up to the forwards' different padding widths — in either matmul mode
    from lmrl_gym_amd.algorithms import ppo
    lk = dict(cliprange_value=0.2, cliprange=0.2, value_loss_coef=1.0)
    for ds, inf, tol in ((a, s["inf"], 1e-5), (b, inf_b, 2e-3)):
        tr = ppo.GPT2PPOTrain(inf.policy, inf.value_head, ro.tokens.pad, lk, lr=1e-5)
        _, _, logs = tr.step(**ds.batch(np.arange(32)), train=False)
        assert abs(float(logs["ratio"]) - 1.0) < tol and abs(float(logs["policy"]["approx_kl"])) < tol and float(logs["policy"]["clipfrac"]) == 0.0, logs
        assert float(logs["values"]["values_error"]) >= 0 and float(logs["values"]["clipfrac"]) == 0.0


def test_train_step_on_a_device_batch_equals_the_numpy_batch(wordle_setup):
    from lmrl_gym_amd.algorithms import ppo
    s = wordle_setup
    ro, inf = s["ro"], s["inf"]
    ds, _ = ro.ppo_data(inf, gamma=1.0, lam=0.95, kl_weight=0.001, max_length=ro.cap + 1)
    index = np.random.RandomState(0).permutation(len(ds))[:24]
    batch_d = ds.batch(index)
    host = ds.to_host()
    batch_h = host[index]
    for k in batch_h:
        assert np.array_equal(batch_d[k].cpu().numpy().astype(batch_h[k].dtype), batch_h[k]), k
    kw = dict(cliprange_value=0.2, cliprange=0.2, value_loss_coef=1.0)
    outs = []
    for batch in (batch_h, batch_d):
        pol, _ = s["mk"]("f32")
        tr = ppo.GPT2PPOTrain(pol, s["head"](), ro.tokens.pad, kw, lr=1e-3)
        _, loss, logs = tr.step(**batch)
        torch.cuda.synchronize()
        outs.append((loss, logs, tr.last_grads[0].flat.clone(), tr.last_grads[1].flat.clone(), pol.p.flat.clone()))
    (l0, g0, pg0, hg0, p0), (l1, g1, pg1, hg1, p1) = outs
    assert l0 == l1 and np.isfinite(l0)
    def flat(d, pre=""):
        out = {}
        for k, x in d.items():
            out.update(flat(x, pre + k + ".") if isinstance(x, dict) else {pre + k: float(x)})
        return out
    f0, f1 = flat(g0), flat(g1)
    assert f0.keys() == f1.keys() and len(f0) > 15
    for k in f0:
        assert f0[k] == f1[k] or (np.isnan(f0[k]) and np.isnan(f1[k])), k
    assert torch.equal(pg0, pg1) and torch.equal(hg0, hg1) and torch.equal(p0, p1) and float(pg0.abs().sum()) > 0


def test_small_kernels_against_their_numpy_definitions(dev):
    from lmrl_gym_amd import _lib
    from lmrl_gym_amd.algorithms.common import initialize_attn_mask_pos_ids, masked_rows
    from lmrl_gym_amd.algorithms.ppo_device import mask_pos_device, masked_rows_device
    rng = np.random.RandomState(1)
    for B, T in ((7, 5), (33, 64), (5, 200), (1100, 129)):
        ids = rng.randint(0, 6, size=(B, T)).astype(np.int32)                       # pad = 3 appears anywhere, also inside
        ids[0, :] = 3
        am, pos = initialize_attn_mask_pos_ids(ids, 3)
        am_d, pos_d, nxt = mask_pos_device(torch.from_numpy(ids).to(dev), 3, shifted=True)
        assert np.array_equal(am_d.cpu().numpy(), am.astype(np.uint8)) and np.array_equal(pos_d.cpu().numpy(), pos)
        assert np.array_equal(nxt.cpu().numpy(), am[:, 1:].astype(np.float32))
        sta = rng.rand(B, T - 1) < 0.4
        sta[B // 2] = False
        idx, tgt, ra = masked_rows_device(torch.from_numpy(sta.astype(np.uint8)).to(dev), am_d, torch.from_numpy(ids).to(dev), T)
        m = sta & (am[:, 1:] != 0)
        assert ra == int(m.sum()) and np.array_equal(idx.cpu().numpy(), masked_rows(m, T)) and np.array_equal(tgt.cpu().numpy(), ids[:, 1:][m])
    # row gather of odd-sized byte rows and of 4-byte-multiple rows
    L = _lib.lib()
    for row_bytes in (1, 7, 128, 516):
        src = torch.from_numpy(rng.randint(0, 256, size=(50, row_bytes)).astype(np.uint8)).to(dev)
        index = torch.from_numpy(rng.randint(0, 50, size=31).astype(np.int32)).to(dev)
        dst = torch.zeros(31, row_bytes, dtype=torch.uint8, device=dev)
        _lib.check(L.lmrl_gather_rows_bytes(src.data_ptr(), index.data_ptr(), dst.data_ptr(), 31, row_bytes, _lib.stream_ptr()))
        assert torch.equal(dst, src[index.long()])


def test_engine_load_params_equals_a_fresh_engine(wordle_setup, dev):
    from lmrl_gym_amd.gpt2 import GPT2Engine, init_hf_style_state_dict
    s = wordle_setup
    cfg = s["cfg"]
    new_sd = init_hf_style_state_dict(cfg, seed=77)
    for k in new_sd:
        if "ln_" in k:
            new_sd[k] = new_sd[k] + 0.1 * torch.randn(new_sd[k].shape, generator=torch.Generator().manual_seed(len(k)))
    eng = GPT2Engine(cfg, s["sd"], dev)
    ses = eng.session(8, 32)
    eng.load_params({k: v.to(dev) for k, v in new_sd.items()})                        # device fp32 masters, as a trainer holds them
    fresh = GPT2Engine(cfg, new_sd, dev)
    ses_f = fresh.session(8, 32)
    toks = torch.from_numpy(np.random.RandomState(2).randint(0, cfg.vocab, size=8 * 8).astype(np.int32)).to(dev)
    cnt = torch.full((8,), 8, dtype=torch.int32, device=dev)
    for x in (ses, ses_f):
        x.reset()
        x.forward(toks, cnt, 8)
    torch.cuda.synchronize()
    assert torch.equal(ses.last_hidden, ses_f.last_hidden) and float(ses.last_hidden.float().abs().sum()) > 0


def test_ppo_rollouts_round_and_online_iteration(wordle_setup):
    """One whole data-collection round over several episode batches (graph replays), advantages whitened over ALL rollouts of the round, then
    train steps on device batches and the weights pushed back into the rollout engine in place."""
    from lmrl_gym_amd.algorithms import ppo
    s = wordle_setup
    ro, inf = s["ro"], s["inf"]
    n = 2 * s["B"] + 17
    ds, kls, summary = ro.ppo_rollouts(inf, n, seed_generator=iter(range(500, 10 ** 6)), gamma=1.0, lam=0.95, kl_weight=0.001, max_length=ro.cap + 1,
                                       temperature=1.0, sample_seed=9, use_graph=True)
    assert len(ds) == n and set(summary) == {"reward", "done", "length"} and summary["length"]["max"] <= 6 and summary["done"]["mean"] == 1.0
    h = ds.to_host()
    a = h.old_advantages[h.should_take_action]
    assert kls.numel() == a.size and abs(float(a.mean())) < 1e-4 and abs(float(a.std()) - 1.0) < 1e-3
    assert (h.old_advantages[~h.should_take_action] == 0).all()
    pol = inf.policy
    tr = ppo.GPT2PPOTrain(pol, inf.value_head, ro.tokens.pad, dict(cliprange_value=0.2, cliprange=0.2, value_loss_coef=1.0), lr=1e-4)
    before = ro.eng.layers[0][2].clone()
    for step in range(2):
        _, loss, logs = tr.step(**ds.batch(np.arange(step * 32, step * 32 + 32)))
        assert np.isfinite(loss)
    ro.eng.load_params(pol.p)
    torch.cuda.synchronize()
    assert not torch.equal(before, ro.eng.layers[0][2])
    assert torch.equal(ro.eng.layers[0][2], pol.p["h.0.attn.c_attn.weight"].t().to(torch.bfloat16))
    ds2, _, _ = ro.ppo_rollouts(inf, s["B"], seed_generator=iter(range(9000, 10 ** 6)), gamma=1.0, lam=0.95, kl_weight=0.001, max_length=ro.cap + 1,
                                temperature=1.0, sample_seed=9, use_graph=True)          # the captured graph replays on the new weights
    assert len(ds2) == s["B"]


def _records_from_dicts(recs, dev, cap=None):
    from lmrl_gym_amd.algorithms.ppo_device import PPORecords
    cap = cap or max(len(r["tokens"]) for r in recs)
    n = len(recs)
    tok, ia, rw = np.zeros((n, cap), np.int32), np.zeros((n, cap), np.uint8), np.zeros((n, cap), np.float32)
    for k, r in enumerate(recs):
        m = len(r["tokens"])
        tok[k, :m], ia[k, :m], rw[k, :m] = r["tokens"], r["is_action"], r["reward"]
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return PPORecords(up(tok), up(ia), up(rw), up(np.array([len(r["tokens"]) for r in recs], np.int32)), up(np.array([r["done"] for r in recs], np.uint8)))


def _dicts_from_records(rec):
    tok, ia, rw = rec.tokens.cpu().numpy(), rec.is_action.cpu().numpy(), rec.reward.cpu().numpy()
    n, dn = rec.n_tok.cpu().numpy(), rec.done.cpu().numpy()
    return [dict(tokens=tok[k, :n[k]].tolist(), is_action=ia[k, :n[k]].astype(int).tolist(), reward=[float(x) for x in rw[k, :n[k]]], done=bool(dn[k]))
            for k in range(rec.n)]


def test_length_rule_on_the_device_equals_the_reference_loop(dev):
    """`lmrl_ppo_truncate_turns` + row compaction against tests/golden/ppo_truncation.json — the reference script's own loop
    (llm_rl_scripts/wordle/ppo/train_ppo_gpt2.py:317-341) EXECUTED on synthetic rollouts (make_truncation_fixture.py): kept episodes, their
    tokens / flags, the folded float32 rewards and the cleared `done`, bit for bit; then larger random records against the oracle's form."""
    from lmrl_gym_amd.algorithms.ppo_device import truncate_turns
    from oracle import rl as ORL
    fx = load_golden("ppo_truncation.json")
    for case in fx["cases"]:
        rec = _records_from_dicts(case["records"], dev)
        before = rec.reward.clone()
        out, info = truncate_turns(rec, case["max_length"], case["gamma"])
        got = _dicts_from_records(out)
        assert got == case["kept"], case["max_length"]
        assert info["skipped"] == len(case["records"]) - len(case["kept"])
        assert info["shortened"] >= sum(1 for g in got if not any(g["tokens"] == r["tokens"] for r in case["records"]))
        assert torch.equal(rec.reward, before)                                   # the caller's record is not modified
    # 3000 random records (up to 12 turns, cap 160, assorted rewards and gammas): device == oracle
    rng = np.random.default_rng(3)
    recs = []
    for _ in range(3000):
        toks, ia, rw = [1, 2, 3], [0, 0, 0], [0.0, 0.0, 0.0]
        for turn in range(int(rng.integers(0, 13))):
            na, no = int(rng.integers(1, 8)), int(rng.integers(1, 8))
            toks += rng.integers(4, 90, size=na + no).tolist()
            ia += [1] * na + [0] * no
            rw += [0.0] * (na - 1) + [float(rng.choice([-1.0, 0.0, -10.0, 0.3, 2.5]))] + [0.0] * no
        recs.append(dict(tokens=toks, is_action=ia, reward=rw, done=bool(rng.integers(0, 2))))
    rec = _records_from_dicts(recs, dev, cap=3 + 12 * 14)
    for ml, gamma in ((40, 0.9), (90, 1.0), (12, 0.37)):
        exp = [k for k in (ORL.truncate_turns_record(r, ml, gamma) for r in recs) if k is not None]
        out, info = truncate_turns(rec, ml, gamma)
        got = _dicts_from_records(out)
        assert len(got) == len(exp) == len(recs) - info["skipped"] and 0 < info["skipped"] < len(recs)
        assert got == exp


def test_reference_truncation_asserts_are_raised(wordle_setup, dev):
    """CombinedTokenTrajectoryChain.from_token_trajectory_chain asserts 'trajectory truncation error' (ppo/base_interface.py:318-327) when max_length
    cuts action tokens or a chain continues with an action token; the device path refuses the same inputs (it used to clip silently), and a
    caller's chain_len_bound that is too short for its chains is an error instead of dropped GAE slots."""
    from lmrl_gym_amd import environment as E
    from lmrl_gym_amd.algorithms.ppo_device import PPORecords, ppo_data_from_records
    s = wordle_setup
    inf, ro = s["inf"], s["ro"]
    kw = dict(gamma=1.0, lam=0.95, kl_weight=0.001)
    long = _records_from_dicts([dict(tokens=list(range(5, 45)), is_action=[0] * 4 + [1, 1, 0, 0] * 9, reward=[0.0] * 40, done=True),
                                dict(tokens=list(range(5, 25)), is_action=[0] * 4 + [1, 1, 0, 0] * 4, reward=[0.0] * 20, done=True)], dev)
    ds, _ = ppo_data_from_records(inf, long, max_length=48, **kw)                   # nothing is cut
    assert len(ds) == 2
    ds, _ = ppo_data_from_records(inf, long, max_length=39, **kw)                   # one observation token is cut: allowed (ends_with_state)
    assert ds.longest == 39
    with pytest.raises(ValueError, match="trajectory truncation error"):
        ppo_data_from_records(inf, long, max_length=30, **kw)                        # action tokens beyond 30 (the engines' wrappers apply the length rule first)
    mk = lambda toks, ia, rw, dn: E.TokenTrajectory(np.asarray(toks, np.int32), np.asarray(ia, bool), np.asarray(rw, np.float32), np.asarray(dn))
    a = mk([5, 6, 7, 8], [0, 0, 1, 1], [0, 0, 0, 1.0], False)
    b_ok = mk([9, 10, 11], [0, 1, 1], [0, 0, -1.0], True)
    b_bad = mk([9, 10, 11], [1, 1, 0], [0, 0.5, 0], True)
    ds, _ = ppo_data_from_records(inf, PPORecords.from_token_trajectory_chains([E.TokenTrajectoryChain(a, E.TokenTrajectoryChain(b_ok, None))], device=dev), **kw)
    assert len(ds) == 2
    with pytest.raises(ValueError, match="trajectory truncation error"):
        ppo_data_from_records(inf, PPORecords.from_token_trajectory_chains([E.TokenTrajectoryChain(a, E.TokenTrajectoryChain(b_bad, None))], device=dev), **kw)
    rec = PPORecords.from_token_trajectory_chains([E.TokenTrajectoryChain(a, E.TokenTrajectoryChain(b_ok, None))], device=dev)
    rec.chain_len_bound = 3                                                        # the chain has 3 + 2 slots
    with pytest.raises(ValueError, match="chain_len_bound"):
        ppo_data_from_records(inf, rec, **kw)


def test_batches_carry_the_data_builds_own_masks(wordle_setup):
    """A tokenizer whose pad id the policy CAN sample (not this package's default): a pad id inside a trajectory is an attended token of the data
    build (lengths come from the records); the dataset hands those lengths to the train step as attention_mask / position_ids, so the first step
    after a build still sees ratio == 1 — and the build warns."""
    from lmrl_gym_amd.algorithms import ppo
    from lmrl_gym_amd.algorithms.ppo_inference import GPT2PPOInference
    s = wordle_setup
    ro = s["ro"]
    inside = int(ro.traj["tokens"][0, 6])                                           # an id that occurs inside trajectories: play "pad"
    inf = GPT2PPOInference(s["inf"].policy, s["inf"].value_head, inside, initial_policy=s["inf"].initial_policy)
    with pytest.warns(RuntimeWarning, match="INSIDE trajectories"):
        ds, _ = ro.ppo_data(inf, gamma=1.0, lam=0.95, kl_weight=0.001, max_length=ro.cap + 1)
    batch = ds.batch(np.arange(32))
    n_tok = ro.traj["n_tok"][:32].cpu().numpy()
    am = batch["attention_mask"].cpu().numpy()
    assert (am.sum(1) == n_tok).all() and (am[:, :-1] >= am[:, 1:]).all()
    assert ((batch["input_ids"].cpu().numpy() == inside) & (am != 0)).any()         # attended "pad" ids exist
    pos = batch["position_ids"].cpu().numpy()
    assert all((pos[b, :n] == np.arange(n)).all() and (pos[b, n:] == n - 1).all() for b, n in enumerate(n_tok))
    tr = ppo.GPT2PPOTrain(inf.policy, inf.value_head, inside, dict(cliprange_value=0.2, cliprange=0.2, value_loss_coef=1.0), lr=1e-5)
    _, _, logs = tr.step(**batch, train=False)
    assert abs(float(logs["ratio"]) - 1.0) < 1e-5 and abs(float(logs["policy"]["approx_kl"])) < 1e-5
    masks_from_ids = {k: v for k, v in batch.items() if k not in ("attention_mask", "position_ids")}
    _, _, logs2 = tr.step(**masks_from_ids, train=False)                            # the reference's `ids != pad` masks: a different sequence
    r2 = float(logs2["ratio"])
    assert np.isnan(r2) or abs(r2 - 1.0) > 1e-4                                      # (nan: every action token of the batch equals the would-be pad id)


def test_embedding_rows_beyond_the_vocabulary_and_live_rows(dev):
    """lmrl_embed_fwd: ids outside [0, vocab) (the pad id = first id after the vocabulary) embed as a zero row; lmrl_embed_bwd: they own no wte
    row, and with t_row a row whose flag is 0 but whose next position is attended keeps its gradient (left padding / holes), == numpy."""
    from lmrl_gym_amd.train import ops
    rng = np.random.RandomState(0)
    V, P, d, B, T = 37, 16, 128, 5, 9
    R = B * T
    wte, wpe = rng.randn(V, d).astype(np.float32), rng.randn(P, d).astype(np.float32)
    ids = rng.randint(0, V + 2, size=R).astype(np.int32)
    ids[::7] = V                                                                     # pad id
    pos = rng.randint(0, P, size=R).astype(np.int32)
    t = lambda a: torch.from_numpy(a).to(dev)
    x = torch.empty(R, d, dtype=torch.float32, device=dev)
    ops.embed_fwd(t(wte), t(wpe), t(ids), t(pos), x, R, d, vocab=V)
    inb = (ids >= 0) & (ids < V)
    exp = np.where(inb[:, None], wte[np.clip(ids, 0, V - 1)], 0.0) + wpe[pos]
    assert np.array_equal(x.cpu().numpy(), exp.astype(np.float32))
    am = (rng.rand(B, T) < 0.6).astype(np.uint8)
    am[0] = [0, 0, 0, 1, 1, 1, 1, 1, 1]                                              # left padding
    am[1] = [1, 1, 1, 1, 0, 0, 0, 0, 0]                                              # right padding
    dx = rng.randn(R, d).astype(np.float32)
    for t_row, live in ((0, am.reshape(-1) != 0), (T, (am != 0) | np.concatenate([am[:, 1:] != 0, np.zeros((B, 1), bool)], 1))):
        live = np.asarray(live).reshape(-1)
        dwte, dwpe = torch.zeros(V, d, device=dev), torch.zeros(P, d, device=dev)
        ops.embed_bwd(t(dx), t(ids), t(pos), dwte, dwpe, R, d, live=t(am.reshape(-1).copy()), vocab=V, t_row=t_row)
        e_wte, e_wpe = np.zeros((V, d), np.float64), np.zeros((P, d), np.float64)
        for r in range(R):
            if live[r]:
                if inb[r]:
                    e_wte[ids[r]] += dx[r]
                e_wpe[pos[r]] += dx[r]
        np.testing.assert_allclose(dwte.cpu().numpy(), e_wte, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(dwpe.cpu().numpy(), e_wpe, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("matmul", ["f32", "bf16"])
def test_device_ppo_data_at_gpt2_small_size_equals_the_host_form(dev, matmul):
    """The size `bench.py`'s ppo_iteration leg times, minus a factor 4 in envs: GPT-2-small (bench weights, a policy a few updates away from the
    initial policy), 256 lock-step envs, real steered episodes of the device engine, the script's blocking width.  Device path
    (ragged row list, forward width = ceil8(longest), per-chunk log-probs, bf16: CE out of the LM-head accumulators at V = 50 257 on compacted
    rows) == the host-array form (`get_ppo_data_from_token_trajectory_chain`: [32, max_length] forwards, full [32, T, V] logits): ids / masks
    identical, log-probs / values 1e-5 (f32), returns 1e-5, whitened advantages 2e-5, the KL list; in the bf16-matmul mode the two forms run
    the same bf16 products on different paddings (2e-3) and sit within the bf16 bound of the f32 numbers."""
    from lmrl_gym_amd.algorithms.ppo_inference import GPT2PPOInference
    from lmrl_gym_amd.envs import wordle as W
    from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine, init_hf_style_state_dict
    from lmrl_gym_amd.rollout import WordleRolloutEngine
    from lmrl_gym_amd.train.gpt2_f32 import GPT2F32, LinearHeadF32
    cfg = GPT2Config.gpt2_small()
    sd = init_hf_style_state_dict(cfg, seed=0)
    g = torch.Generator().manual_seed(21)
    sd_pol = {k: v + 0.05 * v.abs().mean().clamp_min(1e-3) * torch.randn(v.shape, generator=g) for k, v in sd.items()}
    eng = GPT2Engine(cfg, sd_pol, dev)
    vocab = W.Vocabulary.builtin("wordle_official_400.txt")
    B = 256
    ro = WordleRolloutEngine(eng, vocab, B, max_new_tokens=6, bad_word_reward=-10.0)
    rng = np.random.RandomState(8)
    packed = np.array([W.pack_guess(w) for w in vocab.all_vocab], dtype=np.uint32)
    guesses = torch.from_numpy(packed[rng.randint(0, len(packed), size=(6, B))].view(np.int32)).to(dev)
    ro.run_episode(np.arange(B, dtype=np.uint64) + 900, temperature=1.0, sample_seed=6, scripted_guesses=guesses, steer_strength=13.0)
    torch.cuda.synchronize()
    n_tok = ro.traj["n_tok"].cpu().numpy()
    assert ro.tokens.pad == cfg.vocab and len(set(n_tok.tolist())) > 3              # ragged episodes (valid words, junk actions), unsampleable pad
    pol, init = GPT2F32(sd_pol, cfg.n_head, device=dev, matmul=matmul), GPT2F32(sd, cfg.n_head, device=dev, matmul=matmul)
    head = LinearHeadF32(dict(kernel=torch.randn(cfg.d_model, 1, generator=torch.Generator().manual_seed(2)) * 0.05, bias=torch.tensor([-0.3])), dev)
    inf = GPT2PPOInference(pol, head, ro.tokens.pad, initial_policy=init)
    kw = dict(gamma=1.0, lam=0.95, kl_weight=0.001)
    max_length = ro.cap + 1
    host, kls_h, _ = _host_form(ro, inf, max_length, **kw)
    ds, kls_d = ro.ppo_data(inf, max_length=max_length, bsize=96, lm_head_rows=7000, **kw)      # ragged sequence chunks and ragged row chunks
    dh = ds.to_host()
    assert (dh.input_ids == host.input_ids).all() and (dh.should_take_action == host.should_take_action).all()
    assert host.should_take_action.sum() > 10 * B
    tol = 1e-5 if matmul == "f32" else 2e-3
    np.testing.assert_allclose(dh.old_logprobs, host.old_logprobs, rtol=0, atol=tol)
    np.testing.assert_allclose(dh.old_values, host.old_values, rtol=0, atol=tol)
    np.testing.assert_allclose(dh.old_returns, host.old_returns, rtol=tol, atol=tol)
    np.testing.assert_allclose(dh.old_advantages, host.old_advantages, rtol=2 * tol, atol=2 * tol)
    kd = kls_d.cpu().numpy()
    assert kd.shape == kls_h.shape == (int(host.should_take_action.sum()),)
    np.testing.assert_allclose(kd, kls_h, rtol=1e-5 if matmul == "f32" else 5e-2, atol=5e-6 if matmul == "f32" else 2e-4)
    assert kd.mean() > 1e-6
    if matmul == "bf16":                                                             # ... and within the bf16 bound of the f32 arithmetic
        pol32, init32 = GPT2F32(sd_pol, cfg.n_head, device=dev), GPT2F32(sd, cfg.n_head, device=dev)
        ds32, _ = ro.ppo_data(GPT2PPOInference(pol32, head, ro.tokens.pad, initial_policy=init32), max_length=max_length, **kw)
        h32 = ds32.to_host()
        m = h32.should_take_action
        assert np.abs(dh.old_logprobs - h32.old_logprobs)[m].max() < 0.05 and np.abs(dh.old_logprobs - h32.old_logprobs)[m].mean() < 5e-3
        assert np.abs(dh.old_values - h32.old_values).max() < 0.1 and np.abs(dh.old_values - h32.old_values)[m].mean() < 2e-2
    ro.close()


@pytest.mark.parametrize("matmul", ["f32", "bf16"])
def test_train_step_on_trimmed_batches_equals_the_full_width_step(wordle_setup, matmul):
    """`DevicePPODataset.batch(width=trimmed_width())` — the columns beyond the round's longest episode are dropped; the reference blocks every batch
    to max_input_length + max_output_length only because XLA wants one static shape (train_ppo_gpt2.py:344-353).  Right padding of a causal model
    never reaches a kept position: same loss, same logs (but padding_percentage), same gradients as the full-width batch — what lets
    `bench.py`'s ppo_iteration and the harness train on 128 instead of 1024 columns."""
    from lmrl_gym_amd.algorithms import ppo
    from lmrl_gym_amd.algorithms.ppo_inference import GPT2PPOInference
    s = wordle_setup
    ro = s["ro"]
    pol0, init0 = s["mk"](matmul)
    inf = GPT2PPOInference(pol0, s["head"](), ro.tokens.pad, initial_policy=init0)
    ds, _ = ro.ppo_data(inf, gamma=1.0, lam=0.95, kl_weight=0.001, max_length=ro.cap + 1, pad_to=512)
    w = ds.trimmed_width()
    assert w in (64, 128) and ds.input_ids.shape[1] == 512        # (the module's engine may hold a later test's shorter episodes)
    index = np.arange(32)
    kw = dict(cliprange_value=0.2, cliprange=0.2, value_loss_coef=1.0)
    outs = []
    for width in (None, w):
        pol, _ = s["mk"](matmul)
        tr = ppo.GPT2PPOTrain(pol, s["head"](), ro.tokens.pad, kw, lr=1e-3)
        _, loss, logs = tr.step(**ds.batch(index, width=width))
        torch.cuda.synchronize()
        outs.append((loss, logs, tr.last_grads[0].flat.clone(), tr.last_grads[1].flat.clone()))
    (l0, g0, pg0, hg0), (l1, g1, pg1, hg1) = outs
    tol = 1e-5 if matmul == "f32" else 2e-3
    assert abs(l0 - l1) <= tol * max(1.0, abs(l0))
    for k in ("ratio",):
        assert abs(float(g0[k]) - float(g1[k])) < tol
    for grp in ("policy", "values"):
        for k in g0[grp]:
            a, b = float(np.asarray(g0[grp][k]).reshape(-1)[0]), float(np.asarray(g1[grp][k]).reshape(-1)[0])
            assert abs(a - b) <= tol * max(1.0, abs(a)), (grp, k, a, b)
    assert float(g0["padding_percentage"]) != float(g1["padding_percentage"])
    scale = float(pg0.abs().max())
    assert scale > 0 and float((pg0 - pg1).abs().max()) <= (1e-5 if matmul == "f32" else 2e-2) * scale
    assert float((hg0 - hg1).abs().max()) <= (1e-5 if matmul == "f32" else 2e-2) * float(hg0.abs().max())
