"""Pins oracle/wordle_oracle.c against the reference-generated golden traces and the CPython stdlib RNG."""
import ctypes
import os
import random

import numpy as np
import pytest

from conftest import ROOT, load_golden
from oracle import _lib
from oracle.wordle import OracleWordleEnv, load_words

VOCAB_DIR = os.path.join(ROOT, "lmrl-gym_amd", "data", "vocab")


def _stream(seed, n):
    key, klen = _lib.seed_key(seed)
    out = (ctypes.c_uint32 * n)()
    _lib.lib().orc_mt_stream(key, klen, out, n)
    return list(out)


def test_mt19937_golden():
    g = load_golden("mt19937.json")
    for c in g["cases"]:
        seed = int(c["seed"])
        s = _stream(seed, 1301)
        assert s[:8] == c["first"]
        assert [s[i] for i in c["long_idx"]] == c["long_vals"]
        key, klen = _lib.seed_key(seed)
        ns = (ctypes.c_uint32 * len(c["ns"]))(*c["ns"])
        out = (ctypes.c_uint32 * len(c["ns"]))()
        _lib.lib().orc_mt_choices(key, klen, ns, out, len(c["ns"]))
        assert list(out) == c["choices"]


def test_mt19937_vs_stdlib_random_seeds():
    rng = random.Random(2024)
    for _ in range(50):
        seed = rng.getrandbits(rng.choice([1, 8, 31, 32, 33, 63, 64, 70]))
        r = random.Random(seed)
        assert _stream(seed, 700) == [r.getrandbits(32) for _ in range(700)]


def test_mt19937_survey_sanity_value():
    """Random(123): getrandbits(9) x3 = 26, 137, 44, then choice(range(431)) = 393."""
    r = random.Random(123)
    assert [r.getrandbits(9) for _ in range(3)] == [26, 137, 44] and r.choice(range(431)) == 393
    s = _stream(123, 8)
    assert [w >> 23 for w in s[:3]] == [26, 137, 44]
    k, i = (431).bit_length(), 3                     # _randbelow(431): 9-bit draws until r < 431
    while (s[i] >> (32 - k)) >= 431:
        i += 1
    assert s[i] >> (32 - k) == 393


@pytest.mark.parametrize("tag,fname", [("v431", "wordle_official_400.txt"), ("v2315", "wordle_official.txt")])
def test_wordle_traces(tag, fname):
    g = load_golden(f"wordle_traces_{tag}.json")
    words = load_words(os.path.join(VOCAB_DIR, fname))
    assert len(words) == g["n_words"]
    n_steps = 0
    for ep in g["episodes"]:
        env = OracleWordleEnv(words, require_words_in_vocab=ep["require_in_vocab"], bad_word_reward=ep["bad_word_reward"])
        hist = env.reset(ep["seed"])
        assert hist == (("Wordle:\n", False),)
        for st in ep["steps"]:
            hist = hist + ((st["action"], True),)
            hist, r, done = env.step(hist)
            assert hist[-1] == (st["obs"], False)
            assert float(r) == st["reward"] and isinstance(r, int) == st["reward_is_int"]
            assert done == st["done"]
            trits, nf = env.state()
            assert "".join(map(str, trits.reshape(-1).tolist())) == st["state"]
            assert nf == st["n_filtered"]
            n_steps += 1
    assert n_steps > 200
