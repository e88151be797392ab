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
    # the reference cuts a sequence at its first pad id (`unpad_array`); a random-init policy can sample the table's pad id as an ordinary token:
    # such draws (~1e-5 of the tokens) are rewritten in the record so that both paths see legal input
    tok = ro.traj["tokens"]
    tok[tok == ro.tokens.pad] = 0
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
    with pytest.raises(ValueError):
        ro.ppo_data(inf, max_length=76, **kw)                                        # <= 4 + 6 x (7 + 6) tokens: the script's drop-the-last-turns rule could apply
    # batches cut to the longest episode: the same rows, fewer padded columns
    w = ds.trimmed_width()
    assert ds.longest == int(n_tok.max()) and w == 128 and ds2.trimmed_width() == 128
    full, cut = ds2.batch(np.arange(8)), ds2.batch(np.arange(8), width=w)
    for k in full:
        assert torch.equal(full[k][:, :cut[k].shape[1]], cut[k]) and cut[k].shape[1] == (w if k == "input_ids" else w - 1) and cut[k].is_contiguous()


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
