"""GPU tier: the device-resident Maze rollout loop (lmrl_gym_amd.maze_rollout.MazeRolloutEngine, csrc/maze_tokens.hip,
lmrl_gpt2_kv_gather) against (a) the CPU oracle env replaying the recorded actions, (b) the host text path for the action decoding,
(c) per-turn prefill instead of the prompt-prefix cache, (d) eager launches instead of the per-turn hipGraph, and (e) the generic
host-tokenising path `interact_environment(env, GPT2ValuePolicy)` on greedy decoding — transition by transition."""
import numpy as np
import pytest

import lmrl_gym_amd  # noqa: F401

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

_MERGED = ["move left", "move right", "move up", "move down", "move", " left", " right", " up", " down", "\n\n", "mo", "ve left"]


class MergedByteTokenizer:
    """Bytes 0..255 as ids 0..255, a few multi-byte tokens above (so that an action can be spelled in several ways), pad = special."""
    eos_token_id = 10                      # '\n'

    def __init__(self):
        self.extra = list(_MERGED)
        self.pad_token_id = 256 + len(self.extra)
        self.all_special_ids = [self.pad_token_id]

    def __len__(self):
        return self.pad_token_id + 1

    def encode(self, s):
        return list(s.encode("utf-8"))

    def decode(self, ids):
        out = b""
        for i in ids:
            i = int(i)
            if i < 256:
                out += bytes([i])
            elif i < self.pad_token_id:
                out += self.extra[i - 256].encode()
        return out.decode("utf-8", errors="replace")


def _setup(dev, B, max_new, max_steps, describe="describe_observation_give_position", prefix_cache=True, seed=0, boost=12.0,
           prefix_indexed=True, width="tiny", last_k=1, max_input_length=256, whole_actions_only=False):
    from lmrl_gym_amd.envs import maze as M
    from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine
    from lmrl_gym_amd.maze_rollout import MazeRolloutEngine
    from lmrl_gym_amd.policies import heads_to_engine_layout
    tok = MergedByteTokenizer()
    # model vocabulary a little larger than the tokenizer's.  width="small": GPT-2-small's 12 layers x 12 heads x d = 768 (configs[0] names this
    # model): the prefix-indexed decode attention (XCD re-mapping branch), lmrl_gpt2_kv_gather and the per-turn graph at the size
    # profiles/r04_maze_device.txt times, with the real ~129-token prompt rows of the byte tokenizer
    cfg = GPT2Config(2, 2, 128, 256, len(tok) + 3, 256) if width == "tiny" else GPT2Config(12, 12, 768, 3072, len(tok) + 3, 256)
    pi = GPT2Engine.random_init(cfg, seed=seed, device=dev)
    vb = GPT2Engine.random_init(cfg, seed=seed + 1, device=dev)
    d, V = cfg.d_model, cfg.vocab
    g = torch.Generator().manual_seed(seed + 2)
    bias = torch.full((V,), -boost)
    for i in ((256, 257, 258, 259) if whole_actions_only else (256, 257, 258, 259, 260, 261, 262, 263, 264, 10, 266, 267)):     # action pieces and the eos
        bias[i] = 0.0
    head = heads_to_engine_layout({"dense1.kernel": torch.randn(d, d, generator=g) * 0.05, "dense1.bias": torch.zeros(d),
                                   "dense2.kernel": torch.randn(d, V, generator=g) * 0.3, "dense2.bias": bias}, cfg.vocab_padded, dev)
    env = M.setup_maze_env("double_t_maze", describe, "standard_reward", last_k=last_k, max_steps=max_steps)
    eng = MazeRolloutEngine(pi, tok, env, B, max_new_tokens=max_new, eos_token_id=tok.eos_token_id, prefix_cache=prefix_cache,
                            prefix_indexed=prefix_indexed, value_engine=vb, q1_head=head, q2_head=None, beta=1.0, max_input_length=max_input_length)
    return eng, tok, pi, vb, head, env


def _snapshot(eng):
    """Host copy of the record with everything beyond the valid extents (n_turns, gen_len) zeroed — those cells keep older contents."""
    torch.cuda.synchronize()
    h = {k: v.cpu().numpy().copy() for k, v in eng.traj.items() if k in ("pos", "gen", "gen_len", "action", "reward", "kind", "n_turns", "live",
                                                                         "ep_reward")}
    T, G = h["gen"].shape[1:]
    turn_ok = np.arange(T)[None, :] < h["n_turns"][:, None]
    for k in ("pos", "gen_len", "action", "reward", "kind"):
        h[k] = np.where(turn_ok, h[k], 0)
    h["gen"] = np.where(turn_ok[:, :, None] & (np.arange(G)[None, None, :] < h["gen_len"][:, :, None]), h["gen"], 0)
    return h, eng.env.positions()


@pytest.mark.parametrize("max_new,describe", [(1, "describe_observation_give_position"), (3, "describe_observation_only_walls")])
def test_device_loop_matches_oracle_env_and_host_decoding(max_new, describe):
    from lmrl_gym_amd import _lib
    from lmrl_gym_amd.maze_rollout import maze_out_str_process
    from oracle.maze import ACTIONS, OracleMazeEnv
    dev = _lib.require_gpu()
    B, max_steps = 96, 12
    eng, tok, *_ = _setup(dev, B, max_new, max_steps, describe)
    seeds = [1000 + 7 * i for i in range(B)]
    options = [None if i % 3 else {"init_position": [1, 1 + (i % 5)]} for i in range(B)]
    eng.run_episode(seeds, options, temperature=1.0, sample_seed=5, use_graph=False, sync_every=0)
    torch.cuda.synchronize()
    recs = eng.records()
    names = list(ACTIONS)
    n_moves = n_actions = 0
    for b, rec in enumerate(recs):
        o = OracleMazeEnv("double_t_maze", describe, "standard_reward", last_k=1, max_steps=max_steps)
        hist = o.reset(seeds[b], options[b])
        assert not rec["live"]
        done = False
        for i in range(len(rec["gen"])):
            assert not done
            assert tuple(rec["pos"][i]) == tuple(o.position), (b, i)
            text = maze_out_str_process(tok.decode([t for t in rec["gen"][i] if t not in tok.all_special_ids]))     # the host text path
            code = {"move left\n": 0, "move right\n": 1, "move up\n": 2, "move down\n": 3}.get(text, 4)
            assert code == rec["action"][i], (b, i, rec["gen"][i], text)
            n_actions += code != 4
            before = tuple(o.position)
            hist, r, done = o.step(tuple(hist) + ((text, True),))
            n_moves += tuple(o.position) != before
            assert r == rec["reward"][i]
            kind = rec["kind"][i]
            assert (hist[0][0] == "Failure\n") == (kind == 1) and (hist[0][0] == "Success\n") == (kind == 2)
            if kind in (0, 3):
                assert (kind == 0) == (text in names)
        assert done                                              # every episode ended (goal, or Failure after max_steps steps)
        assert rec["final_pos"] == tuple(o.position) and rec["goal"] == tuple(o.goal)
    # the steered policy does walk (max_new = 3 spells an action only now and then: most texts there exercise the OTHER paths)
    assert (n_actions > B and n_moves > B // 2) if max_new == 1 else (n_actions >= 8 and n_moves >= 3), (n_actions, n_moves)
    # interactions(): the InteractionTransition view, texts re-rendered on the host
    inter = eng.interactions()
    assert all(len(ep) == len(rec["gen"]) and ep[-1].done for ep, rec in zip(inter, recs))
    assert all(t.post_action_history[-1].is_action and t.post_action_history[-1].text.endswith("\n") for ep in inter for t in ep)


@pytest.mark.parametrize("width,B", [("tiny", 64), ("small", 256)])
def test_prefix_cache_graph_and_per_turn_prefill_agree(width, B):
    from lmrl_gym_amd import _lib
    dev = _lib.require_gpu()
    max_new, max_steps = 3, 6
    seeds = [31 * i + 3 for i in range(B)]
    snaps = []
    # indexed prefix (rows read from the prefix cache) eager / graph, per-turn prefill, and the copying form of the cache
    for prefix_cache, use_graph, indexed in ((True, False, True), (True, True, True), (False, False, True), (True, True, False)):
        eng, *_ = _setup(dev, B, max_new, max_steps, prefix_cache=prefix_cache, prefix_indexed=indexed, width=width)
        if width == "small":
            assert eng.max_obs_len >= 100                 # real prompt rows (the byte tokenizer's ~129-token observations)
        eng.run_episode(seeds, None, temperature=0.9, sample_seed=11, episode=2, use_graph=use_graph, sync_every=0 if use_graph else 4)
        snaps.append(_snapshot(eng))
        if use_graph:                                  # a second episode on the same captured graph: fresh seeds, fresh noise
            eng.run_episode([s + 1 for s in seeds], None, temperature=0.9, sample_seed=11, episode=3, use_graph=True)
            again = _snapshot(eng)
            assert not np.array_equal(again[0]["gen"], snaps[-1][0]["gen"])
        eng.close()
    (a, pa), (g, pg), (p, pp), (c, pc) = snaps
    for k in a:
        assert np.array_equal(a[k], g[k]), f"graph replay differs from eager launches in {k}"
        assert np.array_equal(a[k], p[k]), f"prompt-prefix cache differs from per-turn prefill in {k}"
        assert np.array_equal(a[k], c[k]), f"indexed prefix rows differ from copied prefix rows in {k}"
    assert np.array_equal(pa, pg) and np.array_equal(pa, pp) and np.array_equal(pa, pc)
    assert int(a["n_turns"].max()) == max_steps + 1


@pytest.mark.parametrize("width,B", [("tiny", 48), ("small", 256), ("small", 8), ("tiny", 16)])      # B <= 16: the skinny-M products + split decode attention
def test_greedy_device_loop_equals_generic_text_path(width, B):
    """Same weights, greedy decoding: the device loop's transitions == interact_environment(VectorMazeEnv, GPT2ValuePolicy) — the host
    path that renders, tokenises, prefills and decodes every turn (LLM_RL/environment.py:154-207).  width="small": at GPT-2-small's 12 layers /
    12 heads / d = 768 and 256 envs (the device loop had only been compared at d = 128)."""
    from lmrl_gym_amd import _lib, environment as E
    from lmrl_gym_amd.maze_rollout import maze_out_str_process
    from lmrl_gym_amd.policies import GPT2ValuePolicy
    dev = _lib.require_gpu()
    max_new, max_steps = 3, 10
    eng, tok, pi, vb, head, env = _setup(dev, B, max_new, max_steps, boost=6.0, width=width)
    seeds = [5 + 13 * i for i in range(B)]
    eng.run_episode(seeds, None, temperature=0.0, use_graph=True)
    torch.cuda.synchronize()
    mine = eng.interactions()
    pol = GPT2ValuePolicy(pi, vb, head, None, 1.0, tok, max_input_length=256, max_new_tokens=max_new, do_sample=False,
                          eos_token_id=tok.eos_token_id, out_str_process=maze_out_str_process)
    ref = E.interact_environment(env, pol, env_seed=seeds, bsize=B)
    assert len(mine) == len(ref) == B
    n_steps = 0
    for b, (x, y) in enumerate(zip(mine, ref)):
        assert len(x) == len(y), (b, len(x), len(y))
        for i, (tx, ty) in enumerate(zip(x, y)):
            assert tx == ty, (b, i, tx, ty)
            n_steps += 1
    assert n_steps >= B * 3
    kinds = {t.post_action_history[-1].text for ep in mine for t in ep}
    assert len(kinds) >= (3 if B > 16 else 2), kinds             # several distinct actions were taken
    from lmrl_gym_amd.gpt2 import FWD_SKINNY
    assert bool(eng.ses.flags & FWD_SKINNY) == (B <= 16)


@pytest.mark.parametrize("last_k,max_input_length,max_steps,describe,max_new", [
    (40, 1024, 7, "describe_observation_only_walls", 1), (5, 512, 8, "describe_observation_only_walls", 1),
    (40, 160, 6, "describe_observation_give_position", 1), (40, 1024, 7, "describe_observation_only_walls", 3), (4, 512, 6, "describe_observation_give_position", 2)])
def test_history_windows_on_the_device_loop_equal_the_generic_text_path(last_k, max_input_length, max_steps, describe, max_new):
    """MazeEnv(last_k > 1) — partially_observed_bc.py:241 runs last_k = 40 — on the device loop: the prompt is the window of the last k history items
    (maze/env/env.py:182-184), left-truncated to max_input_length tokens (ppo/gpt2/interface.py:519-524).  Greedy decoding, same weights: the
    transitions (windowed pre / post-action / post-transition histories, rewards, done) == interact_environment(env, GPT2ValuePolicy), the host
    path that renders, tokenises and prefills every turn.  Regimes: the window only grows (append turns: just the action's tail + the new
    observation are forwarded), it slides after two turns (last_k = 5 / 4: re-prefill turns), the token budget cuts it (max_input_length = 160),
    and (max_new = 3: this policy then spells strings outside the action dict) the env answers with (observation,) alone, restarting the window
    (env.py:179-180).  max_new = 1: one-token actions ('move left' + the forced newline) — legal moves, real windows.
    Eager turns == graph replays; the schedule flags stay clear."""
    from lmrl_gym_amd import _lib, environment as E
    from lmrl_gym_amd.maze_rollout import maze_out_str_process
    from lmrl_gym_amd.policies import GPT2ValuePolicy
    dev = _lib.require_gpu()
    B = 40
    eng, tok, pi, vb, head, env = _setup(dev, B, max_new, max_steps, describe, boost=6.0 if max_new > 1 else 30.0, last_k=last_k, max_input_length=max_input_length,
                                         whole_actions_only=max_new == 1)
    assert eng.last_k == last_k and not eng.prefix_cache
    kinds = {(("a",) if eng._append_turn[i] else "r") for i in range(eng.T)}
    if last_k == 40 and max_input_length == 1024:
        assert kinds == {("a",)}                                   # the window never slides nor overflows: every turn appends
    else:
        assert "r" in kinds and ("a",) in kinds
    seeds = [5 + 13 * i for i in range(B)]
    eng.run_episode(seeds, None, temperature=0.0, use_graph=False, sync_every=0)
    torch.cuda.synchronize()
    eager = eng.interactions()
    assert eng.history_flags() == 0
    eng.run_episode(seeds, None, temperature=0.0, use_graph=True, sync_every=0)
    torch.cuda.synchronize()
    mine = eng.interactions()
    assert eng.history_flags() == 0
    assert mine == eager
    pol = GPT2ValuePolicy(pi, vb, head, None, 1.0, tok, max_input_length=max_input_length, max_new_tokens=max_new, do_sample=False,
                          eos_token_id=tok.eos_token_id, out_str_process=maze_out_str_process)
    ref = E.interact_environment(env, pol, env_seed=seeds, bsize=B)
    assert len(mine) == len(ref) == B
    n_steps = longest = 0
    for b, (x, y) in enumerate(zip(mine, ref)):
        assert len(x) == len(y), (b, len(x), len(y))
        for i, (tx, ty) in enumerate(zip(x, y)):
            assert tx == ty, (b, i, tx, ty)
            n_steps += 1
            longest = max(longest, len(tx.pre_action_history))
    assert n_steps >= B * 3
    if max_new == 1:
        assert longest == min(last_k, 2 * max_steps + 1)             # legal moves all the way: the window reached its full size
    elif max_new == 3:
        assert longest == 1                                          # every action string was illegal: the window restarted every turn
    # ---- the finished episodes as the partially observed online script's PPO chains (partially_observed_ppo_online.py:372-398: the window's item texts
    # joined by single spaces as ONE non-action text, then the action), built on the device from the record == the script's loop on the host lists
    from lmrl_gym_amd.algorithms.ppo_inference import text_trajectory_chains_partially_observed
    chains = [E.TokenTrajectoryChain.from_text_trajectory_chain(c, tok) for c in text_trajectory_chains_partially_observed(mine)]
    rec = eng.ppo_records()
    flat = [tt for c in chains for tt in c.to_list()]
    assert rec.n == len(flat) == n_steps and rec.n_chains == B == len(chains) and rec.cap == max(len(tt.tokens) for tt in flat)
    h = {k: getattr(rec, k).cpu().numpy() for k in ("tokens", "is_action", "reward", "n_tok", "done", "chain", "pos", "last")}
    k = 0
    for c, ch in enumerate(chains):
        lst, p = ch.to_list(), 0
        for i, tt in enumerate(lst):
            n = int(h["n_tok"][k])
            assert n == len(tt.tokens) and h["tokens"][k, :n].tolist() == tt.tokens.tolist(), (c, i)
            assert h["is_action"][k, :n].astype(bool).tolist() == tt.is_action.tolist() and np.array_equal(h["reward"][k, :n], tt.reward)
            assert h["chain"][k] == c and h["pos"][k] == p and bool(h["last"][k]) == (i == len(lst) - 1)
            p += n - 1
            k += 1
        assert bool(h["done"][c]) == bool(lst[-1].done)
    eng.close()


def test_ppo_records_re_encode_actions_spelled_with_merged_tokens():
    """`ppo_records` (last_k = 1) with a tokenizer whose generated ids are NOT the encoding of the decoded action: the policy spells actions with
    multi-byte tokens ("move left" as one id), the reference's chains hold `tokenizer.encode(action text)` (TokenTrajectory.from_text_trajectory,
    LLM_RL/environment.py:359-370) — bytes for this tokenizer.  A legal action is exported as the encoding of its dict key, any other string as the
    bytes of its decoded tokens: records == the script's loop (train_ppo_online.py:444-465) on the host lists, token for token."""
    from lmrl_gym_amd import _lib, environment as E
    from lmrl_gym_amd.algorithms.ppo_inference import text_trajectory_chains_from_transitions
    dev = _lib.require_gpu()
    B = 32
    eng, tok, *_ = _setup(dev, B, 2, 7, "describe_observation_give_position", boost=14.0)
    eng.run_episode([3 + 7 * i for i in range(B)], None, temperature=1.0, sample_seed=9, use_graph=True, sync_every=0)
    torch.cuda.synchronize()
    inter = eng.interactions()
    chains = [E.TokenTrajectoryChain.from_text_trajectory_chain(c, tok) for c in text_trajectory_chains_from_transitions(inter)]
    rec = eng.ppo_records()
    h = {k: getattr(rec, k).cpu().numpy() for k in ("tokens", "is_action", "reward", "n_tok")}
    gen = eng.traj["gen"].cpu().numpy()
    assert (gen >= 256).any()                                       # multi-byte ids were generated
    codes = eng.traj["action"].cpu().numpy()
    k = legal = other = 0
    for c, ch in enumerate(chains):
        for i, tt in enumerate(ch.to_list()):
            n = int(h["n_tok"][k])
            assert n == len(tt.tokens) and h["tokens"][k, :n].tolist() == tt.tokens.tolist(), (c, i)
            assert h["is_action"][k, :n].astype(bool).tolist() == tt.is_action.tolist() and np.array_equal(h["reward"][k, :n], tt.reward)
            legal += int(codes[c, i] < 4); other += int(codes[c, i] >= 4)
            k += 1
    assert k == rec.n and legal > 3 and other > 10, (legal, other)
    eng.close()


@pytest.mark.parametrize("last_k,max_new,max_steps", [(40, 1, 7), (4, 2, 6)])
def test_history_window_ppo_data_on_the_device_equals_the_host_chain_path(last_k, max_new, max_steps):
    """`MazeRolloutEngine.ppo_data` for item windows (partially_observed_ppo_online.py runs PPO with last_k = 40): the device PPO data built from
    `lmrl_maze_tok_ppo_records_hist`'s chains (one joined-window state + action per transition, chained per episode) == the host-array
    `get_ppo_data_from_token_trajectory_chain` on the chains the script's loop builds from the same episodes' host lists — ids and masks identical,
    log-probs / values 1e-5, returns 1e-5, advantages 2e-5 (whitened over all transitions), KL list; growing 40-item windows with legal moves, and a
    4-item sliding window with illegal action strings (window restarts)."""
    from lmrl_gym_amd import _lib, environment as E
    from lmrl_gym_amd.algorithms import ppo
    from lmrl_gym_amd.algorithms.common import BlockingStrategy, Padding, Truncation
    from lmrl_gym_amd.algorithms.ppo_inference import GPT2PPOInference, text_trajectory_chains_partially_observed
    from lmrl_gym_amd.gpt2 import init_hf_style_state_dict
    from lmrl_gym_amd.train.gpt2_f32 import GPT2F32, LinearHeadF32
    dev = _lib.require_gpu()
    B = 24
    # (boost 14: the sampled ids stay inside the action pieces + newline — a stray byte >= 0x80 decodes to U+FFFD on the host, whose re-encoding is
    # three bytes: the device exports an illegal action string as the bytes of its decoded tokens, exact for valid UTF-8)
    eng, tok, pi, vb, head, env = _setup(dev, B, max_new, max_steps, "describe_observation_only_walls", boost=14.0 if max_new > 1 else 30.0, last_k=last_k,
                                         max_input_length=1024 if last_k == 40 else 512, whole_actions_only=max_new == 1)
    eng.run_episode([7 + 5 * i for i in range(B)], None, temperature=1.0, sample_seed=4, use_graph=True, sync_every=0)
    torch.cuda.synchronize()
    assert eng.history_flags() == 0
    inter = eng.interactions()
    chains = [E.TokenTrajectoryChain.from_text_trajectory_chain(c, tok) for c in text_trajectory_chains_partially_observed(inter)]
    rec = eng.ppo_records()
    assert rec.n == sum(len(x) for x in inter) and rec.n_chains == B
    if max_new > 1:
        assert any(len(tr.pre_action_history) == 1 for ep in inter for tr in ep[1:])          # a window restarted after an illegal action string
    else:
        assert max(len(tr.pre_action_history) for ep in inter for tr in ep) == min(last_k, 2 * max_steps + 1)
    from lmrl_gym_amd.gpt2 import GPT2Config
    cfg = GPT2Config(2, 2, 128, 256, pi.cfg.vocab, 1024)           # the data models: positions for the longest joined window
    d = cfg.d_model
    g = torch.Generator().manual_seed(16)
    sd = init_hf_style_state_dict(cfg, seed=3)
    sd2 = {kk: v + 0.02 * torch.randn(v.shape, generator=g) * v.abs().mean().clamp_min(1e-3) for kk, v in sd.items()}
    pol, init = GPT2F32(sd2, cfg.n_head, device=dev), GPT2F32(sd, cfg.n_head, device=dev)
    vh = LinearHeadF32(dict(kernel=torch.randn(d, 1, generator=g) * 0.05, bias=torch.tensor([0.2])), dev)
    inf = GPT2PPOInference(pol, vh, tok.pad_token_id, initial_policy=init)
    kw = dict(gamma=0.97, lam=0.9, kl_weight=0.05)
    max_length = rec.cap + 1
    assert max_length <= cfg.n_pos
    ds, kls = eng.ppo_data(inf, max_length=max_length, bsize=32, **kw)
    datas, kls_h = inf.get_ppo_data_from_token_trajectory_chain(chains, bsize=16, max_length=max_length, **kw)
    host = ppo.PPODataset.from_ppo_data_list(datas, tok, BlockingStrategy(Padding.RIGHT, Truncation.RIGHT, max_length))
    got = ds.to_host()
    assert (got.input_ids == host.input_ids).all() and (got.should_take_action == host.should_take_action).all()
    np.testing.assert_allclose(got.old_logprobs, host.old_logprobs, rtol=0, atol=1e-5)
    np.testing.assert_allclose(got.old_values, host.old_values, rtol=0, atol=1e-5)
    np.testing.assert_allclose(got.old_returns, host.old_returns, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got.old_advantages, host.old_advantages, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(kls.cpu().numpy(), kls_h, rtol=1e-5, atol=5e-6)
    # a window that no longer fits max_length is the reference's truncation assert (base_interface.py:318-327), not a silent cut
    with pytest.raises(ValueError):
        eng.ppo_data(inf, max_length=rec.cap - 2, bsize=32, **kw)
    eng.close()


def test_text_env_eval_lanes_return_the_one_lane_interactions():
    """`MazeRolloutEngine.text_env_eval(concurrent=2)`: two episode batches in flight on twin engines / two HIP streams (their turns enqueued
    alternately).  Every batch draws its own noise (the episode word), independent of the lane it runs on: the interactions of 5 batches must
    equal the one-lane call's, transition for transition and in the same order; the early-exit peeks are per lane."""
    from lmrl_gym_amd import _lib
    dev = _lib.require_gpu()
    B, max_new, max_steps = 32, 3, 9
    out = []
    for lanes in (1, 2):
        eng, *_ = _setup(dev, B, max_new, max_steps)
        inter, summ = eng.text_env_eval(5 * B - 3, seed_generator=iter(range(400, 4000)), temperature=0.8, sample_seed=3, use_graph=True, concurrent=lanes,
                                        sync_every=4)
        assert (eng._lanes is not None and len(eng._lanes) == 2) == (lanes == 2)
        out.append((inter, summ))
        eng.close()
    assert len(out[0][0]) == 5 * B - 3 and out[0][0] == out[1][0] and out[0][1] == out[1][1]


def test_maze_ppo_records_and_device_ppo_data_equal_the_host_chain_path():
    """Round 5: the Maze twin of `WordleRolloutEngine.ppo_data`.  The finished device episodes as token-trajectory chains in HBM
    (`MazeRolloutEngine.ppo_records`: one trajectory per transition, chained per episode) == the chains the online script builds from
    `raw_results` (llm_rl_scripts/maze/ppo/train_ppo_online.py:444-465) after `TokenTrajectory.from_text_trajectory` — tokens, action flags, reward
    placement, done — and the device PPO data built from them == the host-array `get_ppo_data_from_token_trajectory_chain` on those chains
    (multi-trajectory chains: bootstrap from the chain's last trajectory, GAE across trajectory boundaries, whitening over all transitions)."""
    from lmrl_gym_amd import _lib, environment as E
    from lmrl_gym_amd.algorithms import ppo
    from lmrl_gym_amd.algorithms.common import BlockingStrategy, Padding, Truncation
    from lmrl_gym_amd.algorithms.ppo_inference import GPT2PPOInference, text_trajectory_chains_from_transitions
    from lmrl_gym_amd.datasets import ByteTokenizer
    from lmrl_gym_amd.envs import maze as M
    from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine, init_hf_style_state_dict
    from lmrl_gym_amd.maze_rollout import MazeRolloutEngine
    from lmrl_gym_amd.policies import heads_to_engine_layout
    from lmrl_gym_amd.train.gpt2_f32 import GPT2F32, LinearHeadF32
    dev = _lib.require_gpu()
    tok = ByteTokenizer()
    cfg = GPT2Config(2, 2, 128, 256, len(tok) + 3, 256)
    sd = init_hf_style_state_dict(cfg, seed=3)
    pi, vb = GPT2Engine(cfg, sd, dev), GPT2Engine.random_init(cfg, seed=4, device=dev)
    d, V = cfg.d_model, cfg.vocab
    g = torch.Generator().manual_seed(6)
    bias = torch.full((V,), -30.0)                                 # sampled ids: the letters of the action strings, ' ' and the newline (plain ASCII bytes)
    for ch in "movelftrighupdwn \n":
        bias[ord(ch)] = 0.0
    bias[10] = 1.5
    head = heads_to_engine_layout({"dense1.kernel": torch.randn(d, d, generator=g) * 0.05, "dense1.bias": torch.zeros(d),
                                   "dense2.kernel": torch.randn(d, V, generator=g) * 0.3, "dense2.bias": bias}, cfg.vocab_padded, dev)
    env = M.setup_maze_env("double_t_maze", "describe_observation_give_position", "standard_reward", last_k=1, max_steps=7)
    B = 40
    eng = MazeRolloutEngine(pi, tok, env, B, max_new_tokens=4, eos_token_id=tok.eos_token_id, value_engine=vb, q1_head=head, q2_head=None, beta=1.0)
    eng.run_episode([11 * i + 2 for i in range(B)], None, temperature=1.0, sample_seed=8, use_graph=False, sync_every=0)
    torch.cuda.synchronize()
    # ---- records == the script's chains
    chains = [E.TokenTrajectoryChain.from_text_trajectory_chain(c, tok) for c in text_trajectory_chains_from_transitions(eng.interactions())]
    rec = eng.ppo_records()
    flat = [tt for c in chains for tt in c.to_list()]
    assert rec.n == len(flat) > 3 * B and rec.n_chains == B == len(chains)
    h = {k: getattr(rec, k).cpu().numpy() for k in ("tokens", "is_action", "reward", "n_tok", "done", "chain", "pos", "last")}
    k = 0
    n_act_lens = set()
    for c, ch in enumerate(chains):
        lst, p = ch.to_list(), 0
        for i, tt in enumerate(lst):
            n = int(h["n_tok"][k])
            assert n == len(tt.tokens) and h["tokens"][k, :n].tolist() == tt.tokens.tolist(), (c, i)
            assert h["is_action"][k, :n].astype(bool).tolist() == tt.is_action.tolist() and np.array_equal(h["reward"][k, :n], tt.reward)
            assert h["chain"][k] == c and h["pos"][k] == p and bool(h["last"][k]) == (i == len(lst) - 1)
            n_act_lens.add(int(tt.is_action.sum()))
            p += n - 1
            k += 1
        assert bool(h["done"][c]) == bool(lst[-1].done)
    assert len(n_act_lens) >= 3                                    # actions of 1 .. 5 tokens (early newline, forced newline after max_new ids)
    # ---- device PPO data == host-array form on the same chains
    sd2 = {kk: v + 0.02 * torch.randn(v.shape, generator=g) * v.abs().mean().clamp_min(1e-3) for kk, v in sd.items()}
    pol = GPT2F32(sd2, cfg.n_head, device=dev)
    init = GPT2F32(sd, cfg.n_head, device=dev)
    vh = LinearHeadF32(dict(kernel=torch.randn(d, 1, generator=g) * 0.05, bias=torch.tensor([0.2])), dev)
    inf = GPT2PPOInference(pol, vh, tok.pad_token_id, initial_policy=init)
    kw = dict(gamma=0.97, lam=0.9, kl_weight=0.05)
    max_length = rec.cap + 1
    ds, kls = eng.ppo_data(inf, max_length=max_length, bsize=64, **kw)
    datas, kls_h = inf.get_ppo_data_from_token_trajectory_chain(chains, bsize=32, max_length=max_length, **kw)
    host = ppo.PPODataset.from_ppo_data_list(datas, tok, BlockingStrategy(Padding.RIGHT, Truncation.RIGHT, max_length))
    got = ds.to_host()
    assert (got.input_ids == host.input_ids).all() and (got.should_take_action == host.should_take_action).all()
    np.testing.assert_allclose(got.old_logprobs, host.old_logprobs, rtol=0, atol=1e-5)
    np.testing.assert_allclose(got.old_values, host.old_values, rtol=0, atol=1e-5)
    np.testing.assert_allclose(got.old_returns, host.old_returns, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got.old_advantages, host.old_advantages, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(kls.cpu().numpy(), kls_h, rtol=1e-5, atol=5e-6)
    # ---- a partial export (the last episode batch of a round: the first n envs only) is the prefix of the full one
    part = eng.ppo_records(12)
    assert part.n_chains == 12 and part.n == int((h["chain"] < 12).sum())
    assert (part.tokens.cpu().numpy() == h["tokens"][:part.n]).all() and (part.done.cpu().numpy() == h["done"][:12]).all()
    # ---- one data-collection round (text_env_eval + ppo_dataset_loader of the online script): 52 rollouts = one full batch + 12 envs of a second
    seeds = iter(range(100, 1000))
    ds2, kls2, summ = eng.ppo_rollouts(inf, 52, seeds, None, max_length=max_length, bsize=64, temperature=1.0, sample_seed=8, use_graph=True, **kw)
    sta = ds2.should_take_action
    assert ds2.input_ids.shape[1] == max_length and kls2.numel() == int(sta.sum().item())
    assert 52 * 1 <= ds2.input_ids.shape[0] <= 52 * 8 and abs(ds2.input_ids.shape[0] - 52 * float(summ["length"]["mean"])) < 1e-3
    adv = ds2.old_advantages[sta.bool()].double()
    assert abs(float(adv.mean())) < 1e-4 and abs(float(adv.var(unbiased=False)) - 1.0) < 1e-3       # whitened once over the whole round
    assert summ["reward"]["min"] >= -8 * 4 and 0.0 <= summ["done"]["mean"] <= 1.0
    # ---- the trained weights back into the engine in place (+ the observation prefix cache recomputed) == an engine built from them
    eng.load_params(pol.p)
    fresh = MazeRolloutEngine(GPT2Engine(cfg, sd2, dev), tok, env, B, max_new_tokens=4, eos_token_id=tok.eos_token_id, value_engine=vb, q1_head=head,
                              q2_head=None, beta=1.0)
    recs = []
    for e in (eng, fresh):
        e.run_episode([5 * i + 1 for i in range(B)], None, temperature=1.0, sample_seed=3, episode=0, use_graph=True, sync_every=0)
        torch.cuda.synchronize()
        recs.append({k2: e.traj[k2].cpu().numpy().copy() for k2 in ("gen", "gen_len", "action", "reward", "n_turns")})
    live = np.arange(recs[0]["gen"].shape[1])[None, :] < recs[0]["n_turns"][:, None]                      # (slots past an episode's end keep older episodes' ids)
    for k2 in ("n_turns", "gen_len", "action", "reward"):
        assert (np.where(live, recs[0][k2], 0) == np.where(live, recs[1][k2], 0)).all() if k2 != "n_turns" else (recs[0][k2] == recs[1][k2]).all(), k2
    tokm = live[:, :, None] & (np.arange(recs[0]["gen"].shape[2])[None, None, :] < recs[0]["gen_len"][:, :, None])
    assert (np.where(tokm, recs[0]["gen"], 0) == np.where(tokm, recs[1]["gen"], 0)).all() and tokm.sum() > 3 * B
    fresh.close()
    eng.close()
    eng.close()
