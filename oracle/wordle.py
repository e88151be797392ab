"""Oracle Wordle env (TEST INFRASTRUCTURE): Python face of oracle/wordle_oracle.c.

Mirrors `ReformatWordleEnvironment(WordleEnvironment(...))` of the reference
(llm_rl_scripts/wordle/env/env.py:7-55): same text formatting, same reset/step signature
on plain (text, is_action) tuples so it can be compared 1:1 with the golden traces.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib

TextItem = Tuple[str, bool]


def load_words(path: str) -> List[str]:
    # Vocabulary.from_file (game.py:163-170): keep only stripped lines of length 5
    out = []
    with open(path, "r") as f:
        for line in f:
            w = line.strip()
            if len(w) == 5:
                out.append(w)
    return out


def deformat_action(text: str) -> str:
    # deformat_history for an action item (env.py:22-23)
    return text.strip().replace(" ", "")


def reformat_obs(symbols: str) -> str:
    # reformat_history for a non-action item (env.py:13-16): '' -> '\n', '<g><y>' -> 'g y\n'
    if len(symbols) == 0:
        return "\n"
    return " ".join(symbols) + "\n"


class OracleWordleEnv:
    def __init__(self, words: Sequence[str], require_words_in_vocab: bool = True, bad_word_reward: float = -1.0):
        self.words = list(words)
        self._L = _lib.lib()
        blob = "".join(self.words).encode("ascii")
        self._h = self._L.orc_wordle_create(blob, len(self.words), int(require_words_in_vocab), float(bad_word_reward))
        self.reset(0)

    def __del__(self):
        try:
            self._L.orc_wordle_destroy(self._h)
        except Exception:
            pass

    def reset(self, seed: int) -> Tuple[TextItem, ...]:
        key, klen = _lib.seed_key(seed)
        self._L.orc_wordle_reset(self._h, key, klen)
        return (("Wordle:\n", False),)

    def step(self, text_history: Tuple[TextItem, ...]):
        assert text_history[-1][1]
        action = deformat_action(text_history[-1][0])
        # len(action) in the reference counts code points; a non-ASCII char is simply "not a-z"
        raw = "".join(c if ord(c) < 128 else "?" for c in action).encode("ascii")
        n = len(raw)
        obs = ctypes.create_string_buffer(8)
        ol, ri, dn = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        rw = ctypes.c_double()
        self._L.orc_wordle_step(self._h, raw, n, obs, ctypes.byref(ol), ctypes.byref(rw), ctypes.byref(ri), ctypes.byref(dn))
        if ol.value < 0:
            raise IndexError("Cannot choose from an empty sequence")     # random.choice([]) (game.py:178-179)
        sym = obs.raw[: ol.value].decode("ascii")
        reward = int(rw.value) if ri.value else float(rw.value)
        return text_history + ((reformat_obs(sym), False),), reward, bool(dn.value)

    def state(self):
        buf = (ctypes.c_uint8 * 130)()
        nf = ctypes.c_int()
        self._L.orc_wordle_get_state(self._h, buf, ctypes.byref(nf))
        return np.frombuffer(buf, dtype=np.uint8).reshape(26, 5).copy(), nf.value


def run_scripted_timed(words: Sequence[str], n_envs: int, guess_idx: np.ndarray, require: bool = True, bad: float = -10.0, threads: int = 1):
    """cpu_baseline driver -> (env steps executed, seconds of the stepping loop alone, threads used): the envs are created and reset outside the
    clock, then stepped on `threads` host threads (OpenMP over the independent envs; <= 0: all of them)."""
    import time
    L = _lib.lib()
    blob = "".join(words).encode("ascii")
    hs = (ctypes.c_void_p * n_envs)()
    for e in range(n_envs):
        hs[e] = L.orc_wordle_create(blob, len(words), int(require), float(bad))
        key, klen = _lib.seed_key(e)
        L.orc_wordle_reset(hs[e], key, klen)
    g = np.ascontiguousarray(guess_idx, dtype=np.int32)
    used = ctypes.c_int(0)
    t0 = time.perf_counter()
    n = L.orc_wordle_run_mt(hs, n_envs, g.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), g.shape[0], int(threads), ctypes.byref(used))
    dt = time.perf_counter() - t0
    for e in range(n_envs):
        L.orc_wordle_destroy(hs[e])
    return int(n), dt, int(used.value)


def run_scripted(words: Sequence[str], n_envs: int, guess_idx: np.ndarray, require: bool = True, bad: float = -10.0) -> int:
    """cpu_baseline driver: returns the number of env steps executed (single thread)."""
    L = _lib.lib()
    blob = "".join(words).encode("ascii")
    hs = (ctypes.c_void_p * n_envs)()
    for e in range(n_envs):
        hs[e] = L.orc_wordle_create(blob, len(words), int(require), float(bad))
        key, klen = _lib.seed_key(e)
        L.orc_wordle_reset(hs[e], key, klen)
    g = np.ascontiguousarray(guess_idx, dtype=np.int32)
    steps = g.shape[0]
    n = L.orc_wordle_run(hs, n_envs, g.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), steps)
    for e in range(n_envs):
        L.orc_wordle_destroy(hs[e])
    return int(n)
