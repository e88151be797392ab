"""CPU tier: host-side logic of the product package + the kernels' scalar cores (built for the host)
against the reference-generated golden fixtures.  No GPU, no HIP calls."""
import ctypes
import os
import random
import re

import numpy as np
import pytest

import lmrl_gym_amd  # noqa: F401  (shim -> lmrl-gym_amd/)
from conftest import ROOT, load_golden
from lmrl_gym_amd import environment as E
from lmrl_gym_amd.envs import maze as M
from lmrl_gym_amd.envs import wordle as W

HOST_SO = os.path.join(ROOT, "tests", "_build", "libcore_host.so")


def _host_lib():
    src = os.path.join(ROOT, "tests", "host_core_harness.cpp")
    deps = [src, os.path.join(ROOT, "lmrl-gym_amd", "csrc", "mt19937.h"), os.path.join(ROOT, "lmrl-gym_amd", "csrc", "wordle_core.h")]
    if not os.path.exists(HOST_SO) or any(os.path.getmtime(d) > os.path.getmtime(HOST_SO) for d in deps):
        import subprocess
        os.makedirs(os.path.dirname(HOST_SO), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-std=c++17", "-o", HOST_SO, src])
    L = ctypes.CDLL(HOST_SO)
    L.host_wordle_create.restype = ctypes.c_void_p
    L.host_wordle_create.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_float]
    L.host_wordle_destroy.argtypes = [ctypes.c_void_p]
    L.host_wordle_reset.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
    L.host_wordle_step.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32),
                                   ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_uint8)]
    L.host_wordle_trits.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint32)]
    L.host_mt_stream.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32)]
    L.host_mt_randbelow.argtypes = [ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint32), ctypes.c_int, ctypes.POINTER(ctypes.c_uint32)]
    return L


# ------------------------------------------------------------------ C ABI surface
def test_cabi_library_exports_every_declared_symbol():
    from lmrl_gym_amd import _lib
    L = _lib.lib()   # raises loudly if the .so was not built
    hdr = open(os.path.join(ROOT, "include", "lmrl_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(lmrl_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), name
        assert name in _lib._SIGS, f"{name} missing from the ctypes signature table"
    assert L.lmrl_version() >= 100


# ------------------------------------------------------------------ MT19937 core (device code built for the host)
def test_device_mt19937_core_matches_cpython():
    L = _host_lib()
    g = load_golden("mt19937.json")
    for c in g["cases"]:
        seed = abs(int(c["seed"]))
        if seed >= 1 << 64:
            continue   # device key is at most two 32-bit limbs
        out = (ctypes.c_uint32 * 1301)()
        L.host_mt_stream(seed, 1301, out)
        assert list(out[:8]) == c["first"]
        assert [out[i] for i in c["long_idx"]] == c["long_vals"]
        ns = (ctypes.c_uint32 * len(c["ns"]))(*c["ns"])
        ch = (ctypes.c_uint32 * len(c["ns"]))()
        L.host_mt_randbelow(seed, ns, len(c["ns"]), ch)
        assert list(ch) == c["choices"]
    rng = random.Random(7)
    for _ in range(40):
        seed = rng.getrandbits(rng.choice([1, 16, 32, 33, 64]))
        out = (ctypes.c_uint32 * 640)()
        L.host_mt_stream(seed, 640, out)
        r = random.Random(seed)
        assert list(out) == [r.getrandbits(32) for _ in range(640)]


# ------------------------------------------------------------------ Wordle: mask formulation + host text path vs reference traces
@pytest.mark.parametrize("tag,fname", [("v431", "wordle_official_400.txt"), ("v2315", "wordle_official.txt")])
def test_wordle_mask_core_and_text_path(tag, fname):
    L = _host_lib()
    g = load_golden(f"wordle_traces_{tag}.json")
    vocab = W.Vocabulary.builtin(fname)
    assert vocab.all_vocab_size() == g["n_words"]
    blob = "".join(vocab.all_vocab).encode()
    for ep in g["episodes"]:
        if abs(ep["seed"]) >= 1 << 64:
            continue
        h = L.host_wordle_create(blob, len(vocab.all_vocab), int(ep["require_in_vocab"]), ep["bad_word_reward"])
        L.host_wordle_reset(h, abs(ep["seed"]))
        hist = (E.Text("Wordle:\n", False),)
        for st in ep["steps"]:
            hist = hist + (E.Text(st["action"], True),)
            raw = W.deformat_history(hist)
            obs, rew, flg = ctypes.c_uint32(), ctypes.c_float(), ctypes.c_uint8()
            L.host_wordle_step(h, W.pack_guess(raw[-1].text), ctypes.byref(obs), ctypes.byref(rew), ctypes.byref(flg))
            hist = W.reformat_history(raw + (E.Text(W.transition_text(obs.value), False),))
            assert hist[-1].text == st["obs"]
            assert rew.value == pytest.approx(st["reward"]) and bool(flg.value & 4) == (not st["reward_is_int"])
            assert bool(flg.value & 1) == st["done"]
            tr = (ctypes.c_uint8 * 130)(); nf = ctypes.c_uint32()
            L.host_wordle_trits(h, tr, ctypes.byref(nf))
            assert "".join(map(str, tr)) == st["state"] and nf.value == st["n_filtered"]
        L.host_wordle_destroy(h)


def test_wordle_text_helpers():
    assert W.pack_guess("stare") == sum((ord(c) - 97) << (5 * i) for i, c in enumerate("stare"))
    assert W.unpack_word(W.pack_guess("zebra")) == "zebra"
    for bad in ["", "abcd", "abcdef", "abcdE", "ab-de", "abcée"]:
        assert W.pack_guess(bad) == W.BAD_GUESS
    h = (E.Text("Wordle:\n", False), E.Text("  st a re \n", True), E.Text("\n", False), E.Text("g y b b y\n", False))
    raw = W.deformat_history(h)
    assert [t.text for t in raw] == ["stare", "<>", "<g><y><b><b><y>"]
    assert W.reformat_history(raw) == (E.Text("Wordle:\n", False), E.Text("s t a r e\n", True), E.Text("\n", False),
                                       E.Text("g y b b y\n", False))


# ------------------------------------------------------------------ rollout driver vs the reference's interact_environment / text_env_eval
class _OracleWordleTextEnv(E.TextEnv):
    """TextEnv face of the CPU oracle (test-only) so the product's rollout driver can run without a GPU."""

    def __init__(self, words, bad):
        from oracle.wordle import OracleWordleEnv
        self._mk = lambda: OracleWordleEnv(words, True, bad)
        self._e = self._mk()

    def reset(self, seed=None, options=None):
        return tuple(E.Text(t, a) for t, a in self._e.reset(seed))

    def step(self, text_history):
        # the reference env normalises the whole history on every step (deformat -> reformat)
        norm = W.reformat_history(W.deformat_history(text_history))
        h, r, d = self._e.step(tuple((t.text, t.is_action) for t in text_history))
        return norm + (E.Text(h[-1][0], False),), r, d

    def copy(self):
        c = _OracleWordleTextEnv.__new__(_OracleWordleTextEnv)
        c._mk = self._mk
        c._e = self._mk()
        return c


class _ScriptedPolicy(E.TextPolicy):
    def __init__(self, words, seed):
        self.words, self.rng = words, random.Random(seed)

    def act(self, text_history):
        return text_history + (E.Text(" ".join(self.rng.choice(self.words)) + "\n", True),)


def test_text_env_eval_matches_reference():
    g = load_golden("rl_helpers.json")["text_env_eval"]
    vocab = W.Vocabulary.builtin("wordle_official_400.txt")
    env = _OracleWordleTextEnv(vocab.all_vocab, -10.0)
    pol = _ScriptedPolicy(list(vocab.all_vocab) + g["extra_words"], g["policy_seed"])
    inter, summary = E.text_env_eval(env, pol, n_rollouts=g["n_rollouts"], seed_generator=iter(range(g["first_seed"], g["first_seed"] + 100)),
                                     bsize=g["bsize"], verbose=False)
    th = lambda h: [[t.text, t.is_action] for t in h]
    got = [[dict(pre=th(tr.pre_action_history), post_action=th(tr.post_action_history),
                 post_transition=th(tr.post_transition_history), reward=float(tr.reward), done=bool(tr.done)) for tr in ep]
           for ep in inter]
    assert got == g["interactions"]
    for k in ("reward", "done", "length"):
        for kk in ("mean", "std", "min", "max"):
            assert float(summary[k][kk]) == g["summary"][k][kk]
    assert summary["reward"]["mean"].dtype == np.float32


def test_interact_environment_accepts_initial_history_and_padding():
    class CountEnv(E.TextEnv):
        def reset(self, seed=None, options=None):
            self.k = 0
            return (E.Text("s0\n", False),)

        def step(self, h):
            self.k += 1
            return h + (E.Text(f"s{self.k}\n", False),), -1.0, self.k >= (2 if h[0].text == "s0\n" else 3)

    seen = []

    class Pol(E.BatchedTextPolicy):
        def act(self, hs, done=None):
            seen.append((len(hs), list(done)))
            return [None if d else h + (E.Text("a\n", True),) for h, d in zip(hs, done)]

    out = E.interact_environment(CountEnv(), Pol(), env_seed=[1, 2], bsize=2, npad=1)
    assert [len(x) for x in out] == [2, 2] and seen[0] == (3, [False, False, True])
    env = E.TextEnvToBatchedTextEnv(CountEnv()); env.reset([0], [None])
    out = E.interact_environment(env, Pol(), initial_text_history=(E.Text("x\n", False),), bsize=1)
    assert len(out[0]) == 3 and out[0][-1].done


# ------------------------------------------------------------------ token containers / trajectory records
class _CharTok:
    pad_token_id = 0

    def encode(self, s):
        return [ord(c) for c in s]


def test_token_containers_match_reference():
    g = load_golden("rl_helpers.json")
    th = tuple(E.Text(t, a) for t, a in g["token_history"]["history"])
    tok = E.TokenHistory.from_text_history(th, _CharTok())
    assert tok.tokens.tolist() == g["token_history"]["tokens"] and tok.tokens.dtype == np.int32
    assert tok.is_action.astype(int).tolist() == g["token_history"]["is_action"]
    for ch in g["chains"]:
        node = None
        for item in reversed(ch["text_chain"]):
            tt = E.TextTrajectory(tuple(E.Text(t, a) for t, a in item["history"]), tuple(item["reward"]), item["done"])
            node = E.TextTrajectoryChain(tt, node)
        tch = E.TokenTrajectoryChain.from_text_trajectory_chain(node, _CharTok())
        got = [dict(tokens=t.tokens.tolist(), is_action=t.is_action.astype(int).tolist(), reward=t.reward.tolist(), done=bool(t.done))
               for t in tch.to_list()]
        assert got == ch["token_chain"]
    with pytest.raises(AssertionError):
        E.TextTrajectory((E.Text("a", False),), (1.0,), False)
    with pytest.raises(AssertionError):
        E.TextTrajectory((E.Text("a", True),), (1.0, 2.0), False)


# ------------------------------------------------------------------ maze host logic
def test_maze_host_tables_and_text():
    g = load_golden("maze_traces.json")
    assert M.double_t_maze().tolist() == g["double_t_maze"] and M.maze2d_umaze().tolist() == g["umaze"]
    sol = M.maze_solver(1 - M.double_t_maze(), [(8, 6)])
    for pos, mv in g["double_t_maze_optimal_directions"]:
        assert sol[tuple(pos)] == mv
    assert list(M._ACTION_CODE.items()) == [("move left\n", 0), ("move right\n", 1), ("move up\n", 2), ("move down\n", 3)]
    for name, fn in M._REWARDS.items():
        assert M._reward_table(fn) == M._reward_table(lambda a, gl, p, acts, fn=fn: fn(a, gl, p, acts))
    # observation text for every recorded position
    mazes = {"double_t_maze": M.double_t_maze(), "umaze": M.maze2d_umaze()}
    n = 0
    for ep in g["episodes"]:
        f = M._DESCRIBERS[ep["describe"]]
        assert f(mazes[ep["maze"]], ep["init_position"], ep["goal"]) == ep["reset_obs"]
        for st in ep["steps"]:
            last = st["history"][-1][0]
            if last not in ("Success\n", "Failure\n"):
                assert f(mazes[ep["maze"]], st["position"], ep["goal"]) == last
                n += 1
    assert n > 500


# ------------------------------------------------------------------ rollout token tables
def test_token_class_table():
    from lmrl_gym_amd.rollout import classify_token_string as c
    assert c("a") == (0 | 1 << 25) and c(" b") == (1 | 1 << 25)
    assert c("st") == ((18 | 19 << 5) | 2 << 25)
    assert c("\n") == 1 << 28 and c("  ") == 0 and c("") == 0
    assert c("a\n") == (0 | 1 << 25 | 1 << 29) and c("\ta") == (0 | 1 << 25 | 1 << 28)
    assert c("a\tb") == 7 << 25 and c("A") == 7 << 25 and c("abcdef") == 7 << 25 and c("a-") == 7 << 25


def test_wordle_token_table_roundtrip():
    from lmrl_gym_amd.rollout import WordleTokenTable
    t = WordleTokenTable.default_gpt2()
    ids = t.encode_text("Wordle:\ns t a r e\ng y b b y\n\n")
    assert ids[:4] == t.header and ids[4] == t.letter_first[18] and ids[5] == t.letter_sp[19] and ids[-1] == t.newline
    assert len(ids) == 4 + 6 + 6 + 1
    allids = t.letter_first + t.letter_sp + [t.newline] + t.header[:3]
    assert len(set(allids)) == len(allids)
    cls = t.token_class(50257)
    assert cls[t.letter_sp[4]] == (4 | 1 << 25) and cls[12345] == 7 << 25


def test_maze_move_accuracy_and_optimal_table():
    """compute_move_accuracy (maze_utils.py:63-89) and the optimal-direction table vs the reference's own known answers."""
    import re
    from conftest import load_golden
    from lmrl_gym_amd.envs import maze as M
    from lmrl_gym_amd.environment import Text
    ref = {tuple(k): v for k, v in load_golden("maze_traces.json")["double_t_maze_optimal_directions"]}
    tab = M.double_t_maze_optimal_directions()
    assert tab == ref
    assert M.update_position(M.double_t_maze(), (1, 1), "move right\n") == (1, 2)
    assert M.update_position(M.double_t_maze(), (1, 1), "move up\n") == (1, 1) and M.update_position(M.double_t_maze(), (1, 1), "jump\n") == (1, 1)

    class Perfect:                      # BatchedTextPolicy face
        def __init__(self, wrong_rows=()):
            self.wrong_rows = wrong_rows

        def act(self, hs, done=None):
            out = []
            maze = M.double_t_maze()
            by_obs = {M.describe_observation_give_position(maze, tuple(p), (8, 6)): tuple(p) for p in np.argwhere(maze == 0).tolist()}
            for h in hs:
                pos = by_obs[h[0].text]
                a = "move up\n" if pos[0] in self.wrong_rows else tab.get(pos, "move up\n")
                out.append(h + (Text(a, True),))
            return out

    assert M.compute_move_accuracy(Perfect()) == 100.0
    n_row1 = sum(1 for p in tab if p[0] == 1)
    assert abs(M.compute_move_accuracy(Perfect(wrong_rows=(1,))) - (len(tab) - n_row1) / len(tab) * 100) < 1e-9


def test_compat_import_paths_resolve_to_the_package():
    """compat/: the reference's module paths are thin re-exports of lmrl_gym_amd (no second implementation)."""
    import importlib
    import sys
    root = os.path.join(os.path.dirname(__file__), "..", "compat")
    sys.path.insert(0, os.path.abspath(root))
    try:
        mods = []
        for d, _, fs in os.walk(root):
            for f in fs:
                if f.endswith(".py"):
                    rel = os.path.relpath(os.path.join(d, f), root)[:-3].replace(os.sep, ".")
                    mods.append(rel[:-9] if rel.endswith(".__init__") else rel)
        assert len(mods) >= 30
        for m in mods:
            importlib.import_module(m)
        import lmrl_gym_amd.environment as E
        import lmrl_gym_amd.envs.wordle as W
        from LLM_RL.environment import Text, interact_environment, text_env_eval
        from LLM_RL.algorithms.ppo.base_interface import get_advantages_and_returns, ppo_loss_fn
        from llm_rl_scripts.wordle.env.env import ReformatWordleEnvironment
        from llm_rl_scripts.maze.env.maze_utils import setup_maze_env
        assert Text is E.Text and text_env_eval is E.text_env_eval and interact_environment is E.interact_environment
        assert ReformatWordleEnvironment is W.ReformatWordleEnvironment and callable(setup_maze_env) and callable(ppo_loss_fn)
        assert callable(get_advantages_and_returns)
    finally:
        sys.path.remove(os.path.abspath(root))
        for k in [k for k in sys.modules if k == "LLM_RL" or k.startswith("LLM_RL.") or k == "llm_rl_scripts" or k.startswith("llm_rl_scripts.")]:
            del sys.modules[k]


def test_gpt2_single_byte_token_ids_follow_the_published_construction():
    """The byte-level part of `WordleTokenTable.default_gpt2()` is DERIVED from openai/gpt-2's `bytes_to_unicode` order (the merged-token
    ids stay quoted and unverifiable offline — `from_tokenizer` is the authoritative path)."""
    from lmrl_gym_amd.rollout import WordleTokenTable, gpt2_byte_token_id
    assert [gpt2_byte_token_id(b) for b in (33, 126, 161, 172, 174, 255)] == [0, 93, 94, 105, 106, 187]
    assert gpt2_byte_token_id(0) == 188 and gpt2_byte_token_id(10) == 198 and gpt2_byte_token_id(32) == 220 and gpt2_byte_token_id(173) == 255
    assert sorted(gpt2_byte_token_id(b) for b in range(256)) == list(range(256))
    t = WordleTokenTable.default_gpt2()
    assert t.letter_first == list(range(64, 90)) and t.newline == 198 and t.header[-2:] == [25, 198]
    ids = t.letter_first + t.letter_sp + [t.newline] + t.header
    assert len(set(t.letter_first + t.letter_sp + [t.newline])) == 53          # injective: the device decoding relies on it
