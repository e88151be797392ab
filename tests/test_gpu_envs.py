"""GPU tier (-m gpu): parity of the HIP env kernels and RL reductions, called through the C ABI
(liblmrl_amd.so via ctypes), against the CPU oracle and the reference-generated golden traces."""
import ctypes
import os
import random

import numpy as np
import pytest

import lmrl_gym_amd  # noqa: F401
from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    from lmrl_gym_amd import _lib
    d = _lib.require_gpu()          # raises if no GPU: GPU tests must not silently pass on CPU
    arch = _lib.lib().lmrl_device_arch().decode()
    assert arch.startswith("gfx950"), arch
    return d


def _mt_buf(n, dev):
    from lmrl_gym_amd import _lib
    return torch.zeros(_lib.lib().lmrl_mt_bytes(n), dtype=torch.uint8, device=dev)


def _u64(vals, dev):
    return torch.from_numpy(np.asarray(vals, dtype=np.uint64).view(np.int64).copy()).to(dev)


# ------------------------------------------------------------------ MT19937
def test_mt19937_streams_bit_exact(dev):
    from lmrl_gym_amd import _lib
    L = _lib.lib()
    g = load_golden("mt19937.json")
    cases = [c for c in g["cases"] if abs(int(c["seed"])) < 1 << 64]
    rng = random.Random(11)
    extra = [rng.getrandbits(rng.choice([1, 20, 32, 33, 64])) for _ in range(200)]
    seeds = [abs(int(c["seed"])) for c in cases] + extra
    n, n_out = len(seeds), 1301
    mt = _mt_buf(n, dev)
    seeds_d0 = _u64(seeds, dev)
    _lib.check(L.lmrl_mt_seed(_lib.ptr(mt), _lib.ptr(seeds_d0), None, n, _lib.stream_ptr()))
    out = torch.zeros((n_out, n), dtype=torch.int32, device=dev)
    _lib.check(L.lmrl_mt_stream(_lib.ptr(mt), _lib.ptr(out), n_out, n, _lib.stream_ptr()))
    out = out.cpu().numpy().view(np.uint32)
    for i, c in enumerate(cases):
        assert out[:8, i].tolist() == c["first"]
        assert [int(out[k, i]) for k in c["long_idx"]] == c["long_vals"]
    for j, s in enumerate(extra):
        r = random.Random(s)
        assert out[:, len(cases) + j].tolist() == [r.getrandbits(32) for _ in range(n_out)]
    # masked re-seed leaves the other streams alone; randbelow == random.Random.choice index
    mask = np.zeros(n, dtype=np.uint8); mask[::3] = 1
    seeds_d, mask_d = _u64(seeds, dev), torch.from_numpy(mask).to(dev)   # keep argument tensors alive across the launch
    _lib.check(L.lmrl_mt_seed(_lib.ptr(mt), _lib.ptr(seeds_d), _lib.ptr(mask_d), n, _lib.stream_ptr()))
    bounds = [431, 2315, 1, 2, 3, 7, 12971, 100, 25, 24, 431, 17, 5, 2315, 64, 65]
    ob = torch.zeros((len(bounds), n), dtype=torch.int32, device=dev)
    bounds_d = torch.tensor(bounds, dtype=torch.int32, device=dev)
    _lib.check(L.lmrl_mt_randbelow(_lib.ptr(mt), _lib.ptr(bounds_d), _lib.ptr(ob), len(bounds), n, _lib.stream_ptr()))
    ob = ob.cpu().numpy()
    for i, s in enumerate(seeds):
        r = random.Random(s)
        if not mask[i]:
            for _ in range(n_out):
                r.getrandbits(32)
        assert ob[:, i].tolist() == [r.choice(range(b)) for b in bounds], (i, s)


# ------------------------------------------------------------------ Wordle
def _run_golden_wordle(dev, tag, fname):
    from lmrl_gym_amd import environment as E
    from lmrl_gym_amd.envs import wordle as W
    g = load_golden(f"wordle_traces_{tag}.json")
    vocab = W.Vocabulary.builtin(fname)
    groups = {}
    for ep in g["episodes"]:
        groups.setdefault((ep["require_in_vocab"], ep["bad_word_reward"]), []).append(ep)
    n_steps = 0
    for (req, bad), eps in groups.items():
        env = W.VectorWordleEnv(vocab, require_words_in_vocab=req, bad_word_reward=bad, reformat=True)
        hist = env.reset([ep["seed"] for ep in eps])
        done = [False] * len(eps)
        t = 0
        while not all(done):
            acted = [None if d else h + (E.Text(ep["steps"][t]["action"], True),) for h, d, ep in zip(hist, done, eps)]
            res = env.step(acted, done)
            trits, nf, na = env.export_state()
            for i, ep in enumerate(eps):
                if done[i]:
                    assert res[i] is None
                    continue
                st = ep["steps"][t]
                h, r, d = res[i]
                assert h[-1] == E.Text(st["obs"], False), (ep["seed"], t)
                assert float(r) == st["reward"] and isinstance(r, int) == st["reward_is_int"]
                assert d == st["done"]
                assert "".join(map(str, trits[i].reshape(-1).tolist())) == st["state"]
                assert int(nf[i]) == st["n_filtered"] and int(na[i]) == t + 1
                hist[i], done[i] = h, d
                n_steps += 1
            t += 1
        env.close()
    return n_steps


def test_wordle_golden_traces_v431(dev):
    assert _run_golden_wordle(dev, "v431", "wordle_official_400.txt") > 1000


def test_wordle_golden_traces_v2315(dev):
    assert _run_golden_wordle(dev, "v2315", "wordle_official.txt") > 200


@pytest.mark.parametrize("fname,n", [("wordle_official_400.txt", 2048), ("wordle_official.txt", 1024)])
def test_wordle_vs_oracle_batched(dev, fname, n):
    """N seeded envs, scripted guesses (70% consistent-ish vocab words, 20% random vocab words, 10% junk),
    every output and the full knowledge state compared with the C oracle at every step."""
    from oracle.wordle import OracleWordleEnv
    from lmrl_gym_amd.envs import wordle as W
    vocab = W.Vocabulary.builtin(fname)
    words = vocab.all_vocab
    rng = np.random.RandomState(5)
    seeds = rng.randint(0, 2**31 - 1, size=n).astype(np.uint64)
    seeds[::7] += np.uint64(2**32)      # two-limb keys
    env = W.VectorWordleEnv(vocab, True, -10.0)
    env.reset_device(seeds)
    oracles = [OracleWordleEnv(words, True, -10.0) for _ in range(n)]
    for o, s in zip(oracles, seeds):
        o.reset(int(s))
    done = np.zeros(n, dtype=bool)
    for t in range(6):
        gi = rng.randint(0, len(words), size=n)
        junk = rng.rand(n) < 0.1
        guess = np.array([W.pack_guess(words[k]) for k in gi], dtype=np.uint32)
        texts = [words[k] for k in gi]
        for i in np.nonzero(junk)[0]:
            texts[i] = "".join(rng.choice(list("abcdefghijklmnopqrstuvwxyz"), 5)) if rng.rand() < 0.7 else "xy"
            guess[i] = W.pack_guess(texts[i])
        active = (~done).astype(np.uint8)
        env.step_device(torch.from_numpy(guess.view(np.int32)).to(dev), torch.from_numpy(active).to(dev))
        obs = env.obs.cpu().numpy().view(np.uint32); rew = env.reward.cpu().numpy(); flg = env.flags.cpu().numpy()
        trits, nf, na = env.export_state()
        for i in range(n):
            if done[i]:
                continue
            h, r, d = oracles[i].step((("Wordle:\n", False), (texts[i], True)))
            assert h[-1][0] == W.reformat_history((W.Text(W.transition_text(int(obs[i])), False),))[-1].text
            assert float(r) == float(rew[i]) and d == bool(flg[i] & 1)
            ot, onf = oracles[i].state()
            assert np.array_equal(ot, trits[i]) and onf == int(nf[i])
            done[i] = d
    assert done.all()
    env.close()


def test_wordle_lane_per_env_kernel_is_bit_identical_to_wave_per_env(dev):
    """`lmrl_wordle_step` has two kernel forms (one wavefront per env for rollout-sized batches, one LANE per env from 16 384 envs up):
    same state words, observations, rewards, flags and RNG consumption, step for step, on 20 000 envs with valid / unknown / malformed
    guesses and inactive slots — and the lane form against the C oracle directly on a sample of envs."""
    from oracle.wordle import OracleWordleEnv
    from lmrl_gym_amd import _lib
    from lmrl_gym_amd.envs import wordle as W
    L = _lib.lib()
    for fname in ("wordle_official_400.txt", "wordle_official.txt"):
        vocab = W.Vocabulary.builtin(fname)
        words = vocab.all_vocab
        n = 20000
        rng = np.random.RandomState(3)
        seeds = rng.randint(0, 2**31 - 1, size=n).astype(np.uint64)
        packed = np.array([W.pack_guess(w) for w in words], dtype=np.uint32)
        envs = []
        for variant in (1, 2):
            e = W.VectorWordleEnv(vocab, True, -10.0)
            e.reset_device(seeds)
            _lib.check(L.lmrl_wordle_set_variant(e._ctx, variant))
            envs.append(e)
        sample = list(range(0, n, n // 48))
        oracles = {i: OracleWordleEnv(words, True, -10.0) for i in sample}
        for i, o in oracles.items():
            o.reset(int(seeds[i]))
        done = np.zeros(n, dtype=bool)
        for t in range(6):
            gi = rng.randint(0, len(words), size=n)
            g = packed[gi].copy()
            texts = [words[k] for k in gi]
            junk = np.nonzero(rng.rand(n) < 0.15)[0]
            for i in junk:
                texts[i] = "".join(rng.choice(list("abcdefghijklmnopqrstuvwxyz"), 5)) if rng.rand() < 0.6 else "q"
                g[i] = W.pack_guess(texts[i])
            active = (~done & (rng.rand(n) < 0.97)).astype(np.uint8)          # a few live envs sit a step out
            gd, ad = torch.from_numpy(g.view(np.int32)).to(dev), torch.from_numpy(active).to(dev)
            for e in envs:
                e.step_device(gd, ad)
            a, b = envs
            act = torch.from_numpy(active.astype(bool)).to(dev)
            assert torch.equal(a.state, b.state) and torch.equal(a.mt, b.mt)
            assert torch.equal(a.obs[act], b.obs[act]) and torch.equal(a.reward[act], b.reward[act]) and torch.equal(a.flags[act], b.flags[act])
            obs = b.obs.cpu().numpy().view(np.uint32); rew = b.reward.cpu().numpy(); flg = b.flags.cpu().numpy()
            for i in sample:
                if active[i]:
                    h, r, d = oracles[i].step((("Wordle:\n", False), (texts[i], True)))
                    assert h[-1][0] == W.reformat_history((W.Text(W.transition_text(int(obs[i])), False),))[-1].text
                    assert float(r) == float(rew[i]) and d == bool(flg[i] & 1)
            done |= (flg & 1).astype(bool) & active.astype(bool)
        for e in envs:
            e.close()


def test_wordle_full_size_properties(dev):
    """BASELINE-size batch (65 536 envs, V=2315): determinism, termination after <= 6 steps, reward range,
    win <=> filtered set collapsed to one word; an inactive slot is untouched."""
    from lmrl_gym_amd.envs import wordle as W
    vocab = W.Vocabulary.builtin("wordle_official.txt")
    n = 65536
    rng = np.random.RandomState(0)
    seeds = np.arange(n, dtype=np.uint64)
    guesses = np.array([W.pack_guess(w) for w in vocab.all_vocab], dtype=np.uint32)[rng.randint(0, len(vocab.all_vocab), size=(6, n))]
    runs = []
    for _ in range(2):
        env = W.VectorWordleEnv(vocab, True, -10.0)
        env.reset_device(seeds)
        done = torch.zeros(n, dtype=torch.bool, device=dev)
        log = []
        for t in range(6):
            active = (~done).to(torch.uint8)
            env.step_device(torch.from_numpy(guesses[t].view(np.int32)).to(dev), active)
            flg = env.flags.clone(); rew = env.reward.clone(); obs = env.obs.clone()
            newly = (flg & 1).bool() & ~done
            log.append((obs.cpu().numpy()[(~done).cpu().numpy()], rew.cpu().numpy()[(~done).cpu().numpy()]))
            trits, nf, na = env.export_state()
            act = (~done).cpu().numpy()
            r = rew.cpu().numpy()
            assert set(np.unique(r[act]).tolist()) <= {-1.0, 0.0}
            assert np.all(nf[act & (r == 0.0)] == 1)
            done = done | newly
        assert bool(done.all())
        runs.append((log, env.state.clone()))
        env.close()
    for (o1, r1), (o2, r2) in zip(runs[0][0], runs[1][0]):
        assert np.array_equal(o1, o2) and np.array_equal(r1, r2)
    assert torch.equal(runs[0][1], runs[1][1])


def test_wordle_text_api_through_interact_environment(dev):
    """Drop-in protocol path: text_env_eval over the device env reproduces the reference's transitions."""
    from lmrl_gym_amd import environment as E
    from lmrl_gym_amd.envs import wordle as W
    g = load_golden("rl_helpers.json")["text_env_eval"]
    vocab = W.Vocabulary.builtin("wordle_official_400.txt")
    env = W.ReformatWordleEnvironment(W.WordleEnvironment(vocab, require_words_in_vocab=True, bad_word_reward=-10.0))

    class Pol(E.TextPolicy):
        def __init__(self):
            self.words, self.rng = list(vocab.all_vocab) + g["extra_words"], random.Random(g["policy_seed"])

        def act(self, h):
            return h + (E.Text(" ".join(self.rng.choice(self.words)) + "\n", True),)

    inter, summary = E.text_env_eval(env, Pol(), n_rollouts=g["n_rollouts"], bsize=g["bsize"], verbose=False,
                                     seed_generator=iter(range(g["first_seed"], g["first_seed"] + 100)))
    th = lambda h: [[t.text, t.is_action] for t in h]
    got = [[dict(pre=th(tr.pre_action_history), post_action=th(tr.post_action_history), post_transition=th(tr.post_transition_history),
                 reward=float(tr.reward), done=bool(tr.done)) for tr in ep] for ep in inter]
    assert got == g["interactions"]
    assert all(float(summary[k][kk]) == g["summary"][k][kk] for k in g["summary"] for kk in g["summary"][k])


# ------------------------------------------------------------------ Maze
def test_maze_golden_traces(dev):
    from lmrl_gym_amd import environment as E
    from lmrl_gym_amd.envs import maze as M
    g = load_golden("maze_traces.json")
    groups = {}
    for ep in g["episodes"]:
        groups.setdefault((ep["maze"], ep["describe"], ep["reward_fn"], ep["last_k"], ep["max_steps"]), []).append(ep)
    n_steps = 0
    for (mz, desc, rew, last_k, max_steps), eps in groups.items():
        env = M.setup_maze_env(mz, desc, rew, last_k=last_k, max_steps=max_steps).as_batched()
        hist = env.reset([ep["seed"] for ep in eps], [ep["options"] for ep in eps])
        st = env.positions()
        for i, ep in enumerate(eps):
            assert hist[i] == (E.Text(ep["reset_obs"], False),)
            assert st[i, :2].tolist() == ep["init_position"] and st[i, 2:4].tolist() == ep["goal"]
        done = [False] * len(eps)
        t = 0
        while not all(done):
            acted = [None if d or t >= len(ep["steps"]) else h + (E.Text(ep["steps"][t]["action"], True),)
                     for h, d, ep in zip(hist, done, eps)]
            done = [d or t >= len(ep["steps"]) for d, ep in zip(done, eps)]
            if all(done):
                break
            res = env.step(acted, done)
            st = env.positions()
            for i, ep in enumerate(eps):
                if done[i]:
                    continue
                s = ep["steps"][t]
                h, r, d = res[i]
                assert [[x.text, x.is_action] for x in h] == s["history"]
                assert r == s["reward"] and d == s["done"]
                assert st[i, :2].tolist() == s["position"] and int(st[i, 4]) == s["num_steps"]
                hist[i], done[i] = h, d
                n_steps += 1
            t += 1
        env.close()
    assert n_steps > 1000


def test_maze_vs_oracle_large_batch(dev):
    from oracle.maze import OracleMazeEnv
    from lmrl_gym_amd import environment as E
    from lmrl_gym_amd.envs import maze as M
    n = 4096
    rng = np.random.RandomState(3)
    env = M.setup_maze_env("double_t_maze", "describe_observation_give_position", "standard_reward", last_k=3, max_steps=20).as_batched()
    seeds = rng.randint(0, 2**31 - 1, size=n).tolist()
    hist = env.reset(seeds)
    oracles = [OracleMazeEnv("double_t_maze", "describe_observation_give_position", "standard_reward", last_k=3, max_steps=20) for _ in range(n)]
    for i, (o, s) in enumerate(zip(oracles, seeds)):
        assert tuple((t.text, t.is_action) for t in hist[i]) == o.reset(s)
    done = [False] * n
    acts = list(M.manhatten_actions) + ["hop\n"]
    for t in range(22):
        a = rng.randint(0, 5, size=n)
        acted = [None if d else h + (E.Text(acts[k], True),) for h, d, k in zip(hist, done, a)]
        res = env.step(acted, done)
        for i in range(n):
            if done[i]:
                continue
            oh, orw, od = oracles[i].step(tuple((x.text, x.is_action) for x in acted[i]))
            h, r, d = res[i]
            assert tuple((x.text, x.is_action) for x in h) == oh and r == orw and d == od
            hist[i], done[i] = h, d
        if all(done):
            break
    assert all(done)
    env.close()


# ------------------------------------------------------------------ RL reductions
def _chains(rng, B, L):
    sta = rng.rand(B, L) < 0.35
    lens = rng.randint(1, L + 1, size=B).astype(np.int32)
    lens[0] = L
    sta[1] = False                       # a chain with no action tokens
    for b in range(B):
        sta[b, lens[b]:] = False
    values = rng.randn(B, L + 1).astype(np.float32)
    rewards = rng.randn(B, L).astype(np.float32)
    return sta, lens, values, rewards


# (L <= 512: the register-resident scan kernels with 1 .. 8 chunks of 64 slots; longer chains: the LDS-compaction kernel)
@pytest.mark.parametrize("B,L,gamma,lam", [(64, 37, 0.99, 0.95), (512, 200, 1.0, 0.95), (33, 1024, 0.9, 0.5), (5, 3, 1.0, 1.0), (700, 96, 0.99, 0.95),
                                           (41, 128, 1.0, 0.95), (37, 300, 0.97, 0.9), (19, 512, 0.99, 0.95), (19, 513, 0.99, 0.95), (9, 64, 0.5, 1.0)])
def test_gae_kernel(dev, B, L, gamma, lam):
    from oracle import rl
    from lmrl_gym_amd import _lib
    rng = np.random.RandomState(B + L)
    sta, lens, values, rewards = _chains(rng, B, L)
    dv = lambda x: torch.from_numpy(x).to(dev)
    adv = torch.full((B, L), 7.0, device=dev); ret = torch.full((B, L), 7.0, device=dev)
    v_d, r_d, s_d, l_d = dv(values), dv(rewards), dv(sta.astype(np.uint8)), dv(lens)
    _lib.check(_lib.lib().lmrl_gae(_lib.ptr(v_d), _lib.ptr(r_d), _lib.ptr(s_d), _lib.ptr(l_d),
                                   _lib.ptr(adv), _lib.ptr(ret), B, L, gamma, lam, _lib.stream_ptr()))
    adv, ret = adv.cpu().numpy(), ret.cpu().numpy()
    for b in range(B):
        m = sta[b, :lens[b]]
        a_idx, s_idx, n_idx = rl.get_action_state_next_state_idxs(m)
        vrow = np.concatenate([values[b, :lens[b]], values[b, lens[b]:lens[b] + 1]])
        ea = np.zeros(L, np.float32); er = np.zeros(L, np.float32)
        if len(a_idx):
            a, r = rl.gae(vrow[s_idx][None], vrow[n_idx][None], rewards[b][a_idx][None], gamma, lam)
            ea[a_idx], er[a_idx] = a[0], r[0]
        # fp32 tolerance: the kernel's shuffle scan sums in a different order than the reference's sequential loop
        np.testing.assert_allclose(adv[b], ea, rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(ret[b], er, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("B,L,gamma", [(40, 61, 0.99), (300, 130, 1.0), (7, 1024, 0.95), (500, 96, 0.99), (23, 400, 1.0), (11, 512, 0.9)])
def test_rtg_kernel(dev, B, L, gamma):
    from oracle import rl
    from lmrl_gym_amd import _lib
    rng = np.random.RandomState(L)
    sta, lens, _, rewards = _chains(rng, B, L)
    dv = lambda x: torch.from_numpy(x).to(dev)
    out = torch.full((B, L), 3.0, device=dev)
    r_d, s_d, l_d = dv(rewards), dv(sta.astype(np.uint8)), dv(lens)
    _lib.check(_lib.lib().lmrl_rtg(_lib.ptr(r_d), _lib.ptr(s_d), _lib.ptr(l_d), _lib.ptr(out), B, L,
                                   gamma, _lib.stream_ptr()))
    out = out.cpu().numpy()
    for b in range(B):
        m = sta[b, :lens[b]]
        exp = np.zeros(L, np.float32)
        if m.any():
            exp[np.where(m)[0]] = rl.get_rtg(rewards[b, :lens[b]][m], gamma)
        np.testing.assert_allclose(out[b], exp, rtol=3e-5, atol=3e-5)


def test_register_scan_kernels_equal_the_compaction_kernels(dev):
    """The round-5 register-resident scans against the round-1 LDS-compaction kernels on the same chains (`lmrl_rl_reduce_set_variant(1)`): the two
    associate the same recurrence differently — equal within fp32 rounding, identical zero pattern."""
    from lmrl_gym_amd import _lib
    L_ = _lib.lib()
    rng = np.random.RandomState(3)
    B, L = 3000, 96
    sta, lens, values, rewards = _chains(rng, B, L)
    dv = lambda x: torch.from_numpy(x).to(dev)
    v_d, r_d, s_d, l_d = dv(values), dv(rewards), dv(sta.astype(np.uint8)), dv(lens)
    outs = []
    try:
        for variant in (0, 1, 2, 3):
            L_.lmrl_rl_reduce_set_variant(variant)
            adv, ret, rtg = (torch.full((B, L), 5.0, device=dev) for _ in range(3))
            _lib.check(L_.lmrl_gae(_lib.ptr(v_d), _lib.ptr(r_d), _lib.ptr(s_d), _lib.ptr(l_d), _lib.ptr(adv), _lib.ptr(ret), B, L, 0.99, 0.95, _lib.stream_ptr()))
            _lib.check(L_.lmrl_rtg(_lib.ptr(r_d), _lib.ptr(s_d), _lib.ptr(l_d), _lib.ptr(rtg), B, L, 0.99, _lib.stream_ptr()))
            outs.append([t.cpu().numpy() for t in (adv, ret, rtg)])
    finally:
        L_.lmrl_rl_reduce_set_variant(0)
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert ((a == 0) == (b == 0)).all() and (a[~sta] == 0).all()
            np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("n", [100_003, 100_004, 65536 * 96])
def test_whiten_kernel(dev, n):
    from oracle import rl
    from lmrl_gym_amd import _lib
    rng = np.random.RandomState(9)
    x = (rng.randn(n) * 3 + 1.5).astype(np.float32)
    mask = rng.rand(x.size) < 0.3
    L = _lib.lib()
    xd = torch.from_numpy(x).to(dev); md = torch.from_numpy(mask.astype(np.uint8)).to(dev)
    mom = torch.zeros(3, dtype=torch.float64, device=dev); y = torch.empty_like(xd)
    for shift in (1, 0):
        _lib.check(L.lmrl_whiten_moments(_lib.ptr(xd), _lib.ptr(md), _lib.ptr(mom), x.size, _lib.stream_ptr()))
        _lib.check(L.lmrl_whiten_apply(_lib.ptr(xd), _lib.ptr(md), _lib.ptr(mom), _lib.ptr(y), x.size, shift, _lib.stream_ptr()))
        got = y.cpu().numpy()
        exp = x.copy(); exp[mask] = rl.whiten(x[mask], shift_mean=bool(shift))
        np.testing.assert_allclose(got, exp, rtol=1e-5, atol=1e-5)
        assert mom.cpu().numpy()[2] == mask.sum()


@pytest.mark.parametrize("B,L", [(4096, 96), (777, 70), (65536, 96), (50, 128)])
def test_gae_with_fused_whitening_moments(dev, B, L):
    """`lmrl_gae_moments` (the GAE launch also leaves per-workgroup partial moments of the advantages on action slots) + `lmrl_whiten_apply_partials`
    (every workgroup adds the partials in a fixed order, then applies) == `lmrl_gae` + `lmrl_whiten_moments` + `lmrl_whiten_apply`: advantages and
    returns bit for bit (the same scan), the moments to fp64 rounding, the whitened advantages to 1 ulp-ish; `lmrl_whiten_finish` gives ranks
    the three moments to all-reduce.  Odd row starts of `values` (L + 1 floats per row) are read as unaligned 8-byte pairs."""
    from lmrl_gym_amd import _lib
    Lb = _lib.lib()
    rng = np.random.RandomState(B + L)
    sta, lens, values, rewards = _chains(rng, B, L)
    dv = lambda x: torch.from_numpy(x).to(dev)
    v_d, r_d, s_d, l_d = dv(values), dv(rewards), dv(sta.astype(np.uint8)), dv(lens)
    sp, p = _lib.stream_ptr, _lib.ptr
    adv0, ret0, adv1, ret1 = (torch.full((B, L), 7.0, device=dev) for _ in range(4))
    _lib.check(Lb.lmrl_gae(p(v_d), p(r_d), p(s_d), p(l_d), p(adv0), p(ret0), B, L, 0.99, 0.95, sp()))
    npart = Lb.lmrl_gae_moments_partials(B, L)
    assert npart == -(-B // 16)
    part = torch.zeros(npart, 3, dtype=torch.float64, device=dev)
    _lib.check(Lb.lmrl_gae_moments(p(v_d), p(r_d), p(s_d), p(l_d), p(adv1), p(ret1), B, L, 0.99, 0.95, p(part), sp()))
    assert torch.equal(adv0, adv1) and torch.equal(ret0, ret1)
    mom0, mom1 = torch.zeros(3, dtype=torch.float64, device=dev), torch.zeros(3, dtype=torch.float64, device=dev)
    mask = s_d.view(-1)
    _lib.check(Lb.lmrl_whiten_moments(p(adv0), p(mask), p(mom0), B * L, sp()))
    _lib.check(Lb.lmrl_whiten_finish(p(part), npart, p(mom1), sp()))
    m0, m1 = mom0.cpu().numpy(), mom1.cpu().numpy()
    a = adv0.cpu().numpy().astype(np.float64)[sta]
    assert m1[2] == m0[2] == sta.sum()
    np.testing.assert_allclose(m1[:2], [a.sum(), (a * a).sum()], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(m1, m0, rtol=1e-12, atol=1e-9)
    for shift in (1, 0):
        y0, y1 = torch.empty_like(adv0), torch.empty_like(adv0)
        _lib.check(Lb.lmrl_whiten_apply(p(adv0), p(mask), p(mom0), p(y0), B * L, shift, sp()))
        _lib.check(Lb.lmrl_whiten_apply_partials(p(adv1), p(mask), p(part), npart, p(y1), B * L, shift, sp()))
        np.testing.assert_allclose(y1.cpu().numpy(), y0.cpu().numpy(), rtol=1e-6, atol=1e-6)
        assert (y1.cpu().numpy()[~sta] == 0).all()
    assert Lb.lmrl_gae_moments_partials(100, 97) == 0 and Lb.lmrl_gae_moments_partials(100, 130) == 0
