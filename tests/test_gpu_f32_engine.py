"""GPU tier: the fp32 rollout mode (`GPT2EngineF32`, VERDICT r02 item 6 / "what's weak" #1).

The reference's default rollout arithmetic is float32 (llm_rl_scripts/wordle/bc/eval_bc_gpt2.py:34,69) and BASELINE.json asks for sampled
actions within fp32 tolerance.  Here, with fp32 weights / activations / K-V cache and exact-fp32 MFMA products:
  * hidden states and logits against the float64 oracle at fp32 tolerance (2e-5 of the largest entry; the bf16 engine's bound is 5e-2);
  * a whole Wordle rollout (GPT-2-small, 12 layers) whose EVERY sampled token — not only the decisive draws the bf16 engine is held to — is
    re-derived from the float64 oracle's logits and the reference's own random stream (`jax.random.categorical` under the per-turn /
    per-token key splits, oracle/jax_random.py), together with the env replay on the oracle env.
Tolerance on a draw: the device's perturbed scores carry fp32 rounding (logits ~1e-5, float log a few ulp); a draw is compared unless the
oracle's top-2 perturbed scores are closer than 2e-4 — such draws must be rare (< 0.5 %) and every other draw must match exactly.
The same two tests run on the engine's "bf16x3" matmul mode (three-term bf16 splits: ~16 mantissa bits per product, 3x the bf16 MFMA cost
instead of 16x): hidden states within 3e-4, every draw whose top-2 gap exceeds 3e-3 identical (near-ties < 3 % of the draws)."""
import numpy as np
import pytest

import lmrl_gym_amd  # noqa: F401

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    from lmrl_gym_amd import _lib
    return _lib.require_gpu()


@pytest.mark.parametrize("matmul,tol", [("f32", 2e-5), ("bf16x3", 3e-4)])
def test_f32_engine_hidden_states_and_cache_vs_float64(dev, matmul, tol):
    """Prefill in 16-token chunks with ragged per-env lengths, then single-token decode steps: last hidden state / logits of every env after
    every forward == the float64 oracle on the env's full token prefix, at fp32 tolerance."""
    from lmrl_gym_amd.gpt2 import GPT2Config, init_hf_style_state_dict
    from lmrl_gym_amd.gpt2_f32_engine import GPT2EngineF32
    from oracle import gpt2 as O
    cfg = GPT2Config(3, 4, 256, 1024, 1000, 128)
    sd = init_hf_style_state_dict(cfg, seed=3)
    g = torch.Generator().manual_seed(4)
    for k in sd:
        sd[k] = sd[k] * 3 + (0.1 * torch.randn(sd[k].shape, generator=g) if sd[k].dim() == 1 else 0)
    eng = GPT2EngineF32(cfg, sd, dev, matmul=matmul)       # "bf16x3": three-term bf16 splits of the fp32 operands, ~16 mantissa bits per product
    B = 9
    ses = eng.session(B, 96)
    ses.reset()
    rng = np.random.RandomState(0)
    seqs = [[] for _ in range(B)]

    def step(C, cnts):
        toks = np.zeros((B, C), dtype=np.int32)
        for b in range(B):
            new = rng.randint(0, cfg.vocab, size=cnts[b])
            toks[b, : cnts[b]] = new
            seqs[b] += new.tolist()
        ses.forward(torch.from_numpy(toks.reshape(-1)).to(dev), torch.from_numpy(np.asarray(cnts, dtype=np.int32)).to(dev), C)
        torch.cuda.synchronize()
        assert ses.len.cpu().numpy().tolist() == [len(s) for s in seqs]
        hid = ses.last_hidden.cpu().double().numpy()
        lg = ses.lm_logits().cpu().double().numpy()[:, : cfg.vocab]
        for b in range(B):
            if cnts[b] == 0:
                continue
            ref_lg, ref_h = O.forward(sd, torch.tensor([seqs[b]]), cfg.n_head, return_hidden=True)
            ref_h, ref_lg = ref_h[0, -1].numpy(), ref_lg[0, -1].numpy()
            assert np.abs(hid[b] - ref_h).max() <= tol * max(1.0, np.abs(ref_h).max()), (b, np.abs(hid[b] - ref_h).max())
            assert np.abs(lg[b] - ref_lg).max() <= tol * max(1.0, np.abs(ref_lg).max()), (b, np.abs(lg[b] - ref_lg).max())
    step(16, [16, 16, 5, 16, 1, 16, 9, 16, 16])
    step(16, [16, 3, 0, 16, 0, 7, 16, 1, 16])
    step(8, [8, 8, 8, 2, 8, 0, 8, 8, 8])
    for _ in range(3):
        step(1, [1] * B)
    step(1, [1, 0, 1, 0, 1, 1, 0, 1, 1])
    # broadcast of a 1-env prefix == prefilling it per env
    s1 = eng.session(1, 32); s1.reset()
    hdr = rng.randint(0, cfg.vocab, size=5).astype(np.int32)
    pad8 = np.zeros(8, dtype=np.int32); pad8[:5] = hdr
    s1.forward(torch.from_numpy(pad8).to(dev), torch.tensor([5], dtype=torch.int32, device=dev), 8)
    sa, sb = eng.session(4, 32), eng.session(4, 32)
    sa.reset(); sb.reset()
    sa.broadcast_prefix_from(s1, 5)
    sb.forward(torch.from_numpy(np.tile(pad8, 4)).to(dev), torch.full((4,), 5, dtype=torch.int32, device=dev), 8)
    assert torch.equal(sa.len, sb.len) and torch.equal(sa.last_hidden, sb.last_hidden) and torch.equal(sa.kv[:, :, :, :5], sb.kv[:, :, :, :5])


@pytest.mark.parametrize("matmul,tie", [("f32", 2e-4), ("bf16x3", 3e-3)])
def test_f32_rollout_every_sampled_token_equals_float64_oracle_with_jax_stream(dev, matmul, tie):
    from lmrl_gym_amd import jax_prng as JP
    from lmrl_gym_amd.envs import wordle as W
    from lmrl_gym_amd.gpt2 import GPT2Config, init_hf_style_state_dict
    from lmrl_gym_amd.gpt2_f32_engine import GPT2EngineF32
    from lmrl_gym_amd.rollout import WordleRolloutEngine
    from oracle import gpt2 as O, jax_random as JR
    from oracle.wordle import OracleWordleEnv
    cfg = GPT2Config.gpt2_small()
    sd = init_hf_style_state_dict(cfg, seed=0)               # the bench's weights, NOT rounded to bf16: this engine keeps fp32
    eng = GPT2EngineF32(cfg, sd, dev, matmul=matmul)
    vocab = W.Vocabulary.builtin("wordle_official_400.txt")
    words = vocab.all_vocab
    B, STEER, SEED = 64, 12.5, 2024      # +12.5 on the scripted token: it wins ~3 draws in 4 (e^12.5 against ~50 k logits of unit scale)
    ro = WordleRolloutEngine(eng, vocab, B, max_new_tokens=6, bad_word_reward=-10.0)
    tab = ro.tokens
    rng = np.random.RandomState(5)
    packed = np.array([W.pack_guess(w) for w in words], dtype=np.uint32)
    g_np = packed[rng.randint(0, len(packed), size=(6, B))]
    guesses = torch.from_numpy(g_np.view(np.int32)).to(dev)
    seeds = np.arange(B, dtype=np.uint64) + 7
    ro.run_episode(seeds, temperature=1.0, sample_seed=SEED, scripted_guesses=guesses, steer_strength=STEER, sampler="jax")
    torch.cuda.synchronize()
    trajs = ro.token_trajectories()
    nl = tab.newline
    # ---- env replay on the oracle env + the action runs of every env
    runs_of, n_valid, n_bad = [], 0, 0
    for b in range(B):
        tok, ia, rw, dn = trajs[b]
        hdr = len(tab.header)
        o = OracleWordleEnv(words, True, -10.0)
        hist = o.reset(int(seeds[b]))
        pos, turn, done, runs = hdr, 0, False, []
        while pos < len(tok):
            start = pos
            while pos < len(tok) and ia[pos]:
                pos += 1
            run = tok[start:pos]
            runs.append((start, pos, turn))
            text = "".join(tab.strings.get(int(t), "¿") for t in run)
            hist, r, done = o.step(hist + ((text, True),))
            obs_ids = tab.encode_text(hist[-1][0])
            assert tok[pos:pos + len(obs_ids)].tolist() == obs_ids and rw[pos - 1] == float(r)
            n_valid += r != -10.0; n_bad += r == -10.0
            pos += len(obs_ids); turn += 1
            if done:
                break
        assert pos == len(tok) and done and dn
        runs_of.append(runs)
    assert n_valid > 20 and n_bad > 20                       # a moderately steered policy: real sampling, both env branches
    # ---- every sampled token from the float64 oracle + the reference's stream
    V = cfg.vocab
    maxlen = max(len(t[0]) for t in trajs)
    ids = torch.zeros(B, maxlen, dtype=torch.int64)
    for b in range(B):
        ids[b, : len(trajs[b][0])] = torch.from_numpy(trajs[b][0].astype(np.int64))
    logits = O.forward(sd, ids, cfg.n_head)[:, :, :V].numpy()                # float64; right padding + causal mask: prefixes unaffected
    pol_key = JR.prng_key(SEED)
    n_turns = max(len(r) for r in runs_of)
    checked = near_tie = 0
    for turn in range(6):
        pol_key, new_key = JR.split(pol_key)
        keys = JR.hf_flax_sample_keys(new_key, 6)
        assert [tuple(int(x) for x in k) for k in keys] == (lambda sk: [sk.next() for _ in range(6)])(JP.SampleKeys(tuple(int(x) for x in new_key)))
        if turn >= n_turns:
            continue
        for k in range(6):
            noise = JR.gumbel(keys[k], (B, V)).astype(np.float64)            # ONE [B, V] noise array per sampled position, as in JAX
            for b in range(B):
                run = next((r for r in runs_of[b] if r[2] == turn), None)
                if run is None or k >= min(run[1] - run[0], 6):
                    continue                                                 # env finished earlier / its action ended before token k
                start = run[0]
                z = logits[b, start + k - 1].copy()
                c = (int(g_np[turn, b]) >> (5 * k)) & 31
                st = nl if k >= 5 else (tab.letter_first[c % 26] if k == 0 else tab.letter_sp[c % 26])
                z[st] += STEER
                score = z + noise[b]
                top2 = np.partition(score, -2)[-2:]
                if top2[1] - top2[0] < tie:
                    near_tie += 1
                    continue
                checked += 1
                assert int(score.argmax()) == int(trajs[b][0][start + k]), (turn, k, b)
    assert checked > 1200 and near_tie <= (0.005 if matmul == "f32" else 0.03) * (checked + near_tie), (checked, near_tie)
    ro.close()


@pytest.mark.parametrize("matmul", ["f32", "bf16x3"])
def test_f32_engines_graph_replay_is_bit_identical_to_eager(dev, matmul):
    """The fp32-accurate rollout modes under hipGraph replay (what bench.py times as `fp32_mode` / `bf16x3_mode`): capture_episode + replay == the
    eager episode with the same seeds / guesses / epoch word, bit for bit — tokens, action flags, rewards, counters — for two replays with fresh
    noise.  Covers the deterministic split-K products (fixed-order reduces) and, in bf16x3, the fused LM-head sampler on the split operands."""
    from lmrl_gym_amd.envs import wordle as W
    from lmrl_gym_amd.gpt2 import GPT2Config, init_hf_style_state_dict
    from lmrl_gym_amd.gpt2_f32_engine import GPT2EngineF32
    from lmrl_gym_amd.rollout import WordleRolloutEngine
    cfg = GPT2Config(2, 4, 256, 1024, 50257, 128)
    eng = GPT2EngineF32(cfg, init_hf_style_state_dict(cfg, seed=1), dev, matmul=matmul)
    vocab = W.Vocabulary.builtin("wordle_official_400.txt")
    B = 192
    ro = WordleRolloutEngine(eng, vocab, B, max_new_tokens=6, bad_word_reward=-10.0)
    rng = np.random.RandomState(3)
    packed = np.array([W.pack_guess(w) for w in vocab.all_vocab], dtype=np.uint32)
    guesses = torch.from_numpy(packed[rng.randint(0, len(packed), size=(6, B))].view(np.int32)).to(dev)
    seeds = (torch.arange(B, dtype=torch.int64) + 900).to(dev)
    kw = dict(temperature=1.0, sample_seed=11, steer_strength=12.5)
    ro.capture_episode(scripted=True, **kw)
    snap = lambda: {k: ro.traj[k].clone() for k in ("tokens", "is_action", "reward", "n_tok", "n_steps", "ep_reward", "env_done")}
    outs = []
    for rep in range(2):
        ro.replay_episode(seeds + rep, guesses)
        g = snap()
        epoch = ro.g_epoch.clone()
        ro.sample_step = 0                                  # the captured graph baked steps 0..35
        ro.run_episode(seeds + rep, scripted_guesses=guesses, epoch=epoch, **kw)
        e = snap()
        for k in g:
            assert torch.equal(g[k], e[k]), (matmul, rep, k)
        outs.append(g)
    assert int(outs[0]["n_steps"].sum()) > 3 * B and not torch.equal(outs[0]["tokens"], outs[1]["tokens"])
    ro.close()


def test_bf16x3_lanes_own_their_split_k_workspace(dev):
    """the split-K partial sums of the bf16x3 c_proj products used to live on the ENGINE, which `text_env_eval(concurrent=n)`
    lanes share across HIP streams — one lane's reduce could read the other's partials.  They belong to the session now (sized once per chunk
    size, before any capture).  Greedy episodes are noise-independent, so the same batches must come back bit-identical from one lane and from
    two lanes in flight, on a width whose c_proj takes the split-K path at decode AND chunk size."""
    from lmrl_gym_amd.envs import wordle as W
    from lmrl_gym_amd.gpt2 import GPT2Config, init_hf_style_state_dict
    from lmrl_gym_amd.gpt2_f32_engine import GPT2EngineF32
    from lmrl_gym_amd.rollout import WordleRolloutEngine
    cfg = GPT2Config(2, 4, 256, 2048, 50257, 128)
    eng = GPT2EngineF32(cfg, init_hf_style_state_dict(cfg, seed=2), dev, matmul="bf16x3")
    B = 256
    assert eng.splitk_ws_bytes(B) > 0 and eng.splitk_ws_bytes(8 * B) > 0 and not hasattr(eng, "_splitk_ws")
    vocab = W.Vocabulary.builtin("wordle_official_400.txt")
    ro = WordleRolloutEngine(eng, vocab, B, max_new_tokens=6, bad_word_reward=-10.0)
    kw = dict(temperature=0.0, sample_seed=3, use_graph=True)
    one, s1 = ro.text_env_eval(6 * B, seed_generator=iter(range(100, 10 ** 6)), **kw)
    two, s2 = ro.text_env_eval(6 * B, seed_generator=iter(range(100, 10 ** 6)), concurrent=2, **kw)
    three, _ = ro.text_env_eval(6 * B, seed_generator=iter(range(100, 10 ** 6)), concurrent=3, **kw)
    assert len(ro._lanes) == 3 and one == two == three and s1 == s2
    ws = [e.ses._ws[c]["splitk"].data_ptr() for e, _ in ro._lanes for c in sorted(e.ses._ws)]
    assert len(set(ws)) == len(ws)                        # every lane / chunk size has its own partial-sum buffer
    ro.close()
