"""Wordle on the MI355X: host text <-> packed device state around `lmrl_wordle_*` (csrc/wordle.hip).

Public names mirror the reference (`llm_rl_scripts/wordle/env/{env,game}.py`):
`Vocabulary`, `WordleEnvironment`, `ReformatWordleEnvironment`, `reformat_history`,
`deformat_history`; plus `VectorWordleEnv`, the lock-step `BatchedTextEnv` that steps N envs
with one kernel launch (what `interact_environment` uses via `as_batched()`).
"""
from __future__ import annotations

import os
import random
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .. import _lib
from ..environment import BatchedTextEnv, Text, TextEnv, TextHistory

N_CHARS = 5   # game.py:14
N_TRIES = 6   # game.py:15
BAD_GUESS = 0xFFFFFFFF
_SYM = {1: "g", 2: "y", 3: "b"}
DATA_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "vocab")


# ----------------------------------------------------------------------------- text <-> codes
def pack_guess(action: str) -> int:
    """De-formatted action string -> packed 5x5-bit letters, or BAD_GUESS (game.py:214 first two clauses)."""
    if len(action) != N_CHARS:
        return BAD_GUESS
    p = 0
    for i, c in enumerate(action):
        o = ord(c) - 97
        if o < 0 or o > 25:
            return BAD_GUESS
        p |= o << (5 * i)
    return p


def unpack_word(p: int) -> str:
    return "".join(chr(97 + ((p >> (5 * i)) & 31)) for i in range(N_CHARS))


def obs_symbols(code: int) -> str:
    n = (code >> 16) & 7
    return "".join(_SYM[(code >> (3 * k)) & 7] for k in range(n))


def transition_text(code: int) -> str:
    """Raw WordleGame.transition_sequence() element: '<g><y>...' or '' (game.py:273-288)."""
    return "".join(f"<{s}>" for s in obs_symbols(code))


def reformat_history(text_history: TextHistory) -> TextHistory:
    """Raw game history -> LM-facing text (env.py:7-17)."""
    out = (Text("Wordle:\n", False),)
    for item in text_history:
        if item.is_action:
            out += (Text(" ".join(list(item.text)) + "\n", True),)
        elif len(item.text) == 0:
            out += (Text("\n", False),)
        else:
            out += (Text(" ".join(item.text[1:-1].split("><")) + "\n", False),)
    return out


def deformat_history(text_history: TextHistory) -> TextHistory:
    """LM-facing text -> raw game history, dropping the header item (env.py:19-26)."""
    out: TextHistory = tuple()
    for item in text_history[1:]:
        body = item.text.strip().replace(" ", "")
        if item.is_action:
            out += (Text(body, True),)
        else:
            out += (Text("<" + "><".join(list(body)) + ">", False),)
    return out


class Vocabulary:
    """Ordered word list (file order) + membership set (game.py:134-191).  The filtered view and the RNG
    live on the device, per env."""

    def __init__(self, all_vocab: Sequence[str]):
        self.all_vocab = list(all_vocab)
        self.all_vocab_set = set(self.all_vocab)

    @classmethod
    def from_file(cls, vocab_file: str, fill_cache: bool = True, rng=None) -> "Vocabulary":
        words = []
        with open(vocab_file, "r") as f:
            for line in f:
                w = line.strip()
                if len(w) == N_CHARS:
                    words.append(w)
        return cls(words)

    @classmethod
    def builtin(cls, name: str = "wordle_official_400.txt") -> "Vocabulary":
        return cls.from_file(os.path.join(DATA_DIR, name))

    def all_vocab_size(self) -> int:
        return len(self.all_vocab)

    def __contains__(self, item: str) -> bool:
        return item in self.all_vocab_set


def _seed_magnitude(seed: Optional[int]) -> int:
    if seed is None:   # random.Random(None): OS entropy (not reproducible in the reference either)
        return random.SystemRandom().getrandbits(63)
    m = abs(int(seed))
    if m >= 1 << 64:
        raise ValueError("lmrl_gym_amd: env seeds must fit in 64 bits (device MT19937 key is 1-2 limbs)")
    return m


# ----------------------------------------------------------------------------- device-backed batched env
class VectorWordleEnv(BatchedTextEnv):
    """N Wordle envs stepped in lock-step by one `lmrl_wordle_step` launch.

    `reformat=True` speaks the LM-facing format of `ReformatWordleEnvironment` (header 'Wordle:\\n',
    's t a r e\\n' / 'g y b b y\\n'); `reformat=False` the raw format of `WordleEnvironment`.
    """

    def __init__(self, vocab: Vocabulary, require_words_in_vocab: bool = True, bad_word_reward: float = -1.0,
                 reformat: bool = True):
        import torch
        self.vocab = vocab
        self.require_words_in_vocab = require_words_in_vocab
        self.bad_word_reward = bad_word_reward
        self.reformat = reformat
        self.device = _lib.require_gpu()
        self._L = _lib.lib()
        blob = "".join(vocab.all_vocab).encode("ascii")
        self._ctx = self._L.lmrl_wordle_create(blob, len(vocab.all_vocab), int(require_words_in_vocab), float(bad_word_reward))
        if not self._ctx:
            raise _lib.LmrlError(self._L.lmrl_last_error().decode())
        self.n = 0
        self._torch = torch

    # -- device buffers
    def _alloc(self, n: int):
        t = self._torch
        self.n = n
        self.state = t.zeros(self._L.lmrl_wordle_state_bytes(n), dtype=t.uint8, device=self.device)
        self.mt = t.zeros(self._L.lmrl_mt_bytes(n), dtype=t.uint8, device=self.device)
        self.obs = t.zeros(n, dtype=t.int32, device=self.device)
        self.reward = t.zeros(n, dtype=t.float32, device=self.device)
        self.flags = t.zeros(n, dtype=t.uint8, device=self.device)

    def reset_device(self, seeds, mask=None) -> None:
        """seeds: numpy uint64 array OR an int64 device tensor (no host sync: lets consecutive episodes pipeline);
        mask: optional numpy / device uint8 array selecting the envs to reset."""
        t = self._torch
        n = len(seeds)
        if n != self.n:
            self._alloc(n)
        if isinstance(seeds, t.Tensor):
            seeds_d = seeds
        else:
            seeds_d = t.from_numpy(np.asarray(seeds, dtype=np.uint64).view(np.int64).copy()).to(self.device)
        if mask is None or isinstance(mask, t.Tensor):
            mask_d = mask
        else:
            mask_d = t.from_numpy(np.asarray(mask, dtype=np.uint8)).to(self.device)
        _lib.check(self._L.lmrl_wordle_reset(self._ctx, _lib.ptr(self.state), _lib.ptr(self.mt), _lib.ptr(seeds_d),
                                             _lib.ptr(mask_d), n, _lib.stream_ptr()), "lmrl_wordle_reset")

    def step_device(self, guess_d, active_d=None) -> None:
        """Device-side step: guess_d int32[N] packed guesses (BAD_GUESS as -1); results in self.obs/reward/flags."""
        _lib.check(self._L.lmrl_wordle_step(self._ctx, _lib.ptr(self.state), _lib.ptr(self.mt), _lib.ptr(guess_d),
                                            _lib.ptr(active_d), _lib.ptr(self.obs), _lib.ptr(self.reward),
                                            _lib.ptr(self.flags), self.n, _lib.stream_ptr()), "lmrl_wordle_step")

    def export_state(self):
        t = self._torch
        trits = t.empty((self.n, 26, 5), dtype=t.uint8, device=self.device)
        nf = t.empty(self.n, dtype=t.int32, device=self.device)
        na = t.empty(self.n, dtype=t.int32, device=self.device)
        _lib.check(self._L.lmrl_wordle_export_state(_lib.ptr(self.state), _lib.ptr(trits), _lib.ptr(nf), _lib.ptr(na),
                                                    self.n, _lib.stream_ptr()), "lmrl_wordle_export_state")
        return trits.cpu().numpy(), nf.cpu().numpy(), na.cpu().numpy()

    # -- BatchedTextEnv protocol
    def reset(self, seed=None, options=None) -> List[TextHistory]:
        if seed is None and options is None:
            seed = [None]
        elif seed is None:
            seed = [None] * len(options)
        self.reset_device(np.array([_seed_magnitude(s) for s in seed], dtype=np.uint64))
        first: TextHistory = (Text("Wordle:\n", False),) if self.reformat else tuple()
        return [first for _ in seed]

    def step(self, text_history, done=None):
        t = self._torch
        assert self.n > 0, "reset must be called before step"
        assert len(text_history) == self.n, "batch size must be the same as the number of environments initalized"
        if done is None:
            done = [False] * self.n
        assert len(done) == self.n
        guess = np.full(self.n, BAD_GUESS, dtype=np.uint32)
        active = np.zeros(self.n, dtype=np.uint8)
        raw_hist: List[Optional[TextHistory]] = [None] * self.n
        for i, (h, d) in enumerate(zip(text_history, done)):
            if d or h is None:
                continue
            rh = deformat_history(h) if self.reformat else tuple(h)
            assert rh[-1].is_action
            raw_hist[i] = rh
            guess[i] = pack_guess(rh[-1].text)
            active[i] = 1
        guess_d = t.from_numpy(guess.view(np.int32)).to(self.device)
        active_d = t.from_numpy(active).to(self.device)
        self.step_device(guess_d, active_d)
        obs = self.obs.cpu().numpy().view(np.uint32)
        rew = self.reward.cpu().numpy()
        flg = self.flags.cpu().numpy()
        if (flg[active > 0] & 8).any():     # random.choice([]) in the reference (game.py:178-179, 219): same exception type, same place
            raise IndexError("Cannot choose from an empty sequence (filtered vocabulary is empty)")
        out = []
        for i in range(self.n):
            if not active[i]:
                out.append(None)
                continue
            hist = raw_hist[i] + (Text(transition_text(int(obs[i])), False),)
            if self.reformat:
                hist = reformat_history(hist)
            # reward typing follows game.py:290-293: float bad_word_reward, else int -1 / 0
            r = self.bad_word_reward if (flg[i] & 4) else int(rew[i])
            out.append((hist, r, bool(flg[i] & 1)))
        return out

    def close(self) -> None:
        if getattr(self, "_ctx", None):
            self._L.lmrl_wordle_destroy(self._ctx)
            self._ctx = None

    def copy(self):
        return VectorWordleEnv(self.vocab, self.require_words_in_vocab, self.bad_word_reward, self.reformat)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class WordleEnvironment(TextEnv):
    """Single raw-format env (env.py:39-55), one device slot."""
    _reformat = False

    def __init__(self, vocab: Vocabulary, require_words_in_vocab: bool = True, bad_word_reward: float = -1.0):
        self.vocab = vocab
        self.require_words_in_vocab = require_words_in_vocab
        self.bad_word_reward = bad_word_reward
        self._vec: Optional[VectorWordleEnv] = None

    def as_batched(self) -> VectorWordleEnv:
        return VectorWordleEnv(self.vocab, self.require_words_in_vocab, self.bad_word_reward, reformat=self._reformat)

    def _v(self) -> VectorWordleEnv:
        if self._vec is None:
            self._vec = self.as_batched()
        return self._vec

    def reset(self, seed: Optional[int] = None, options: Optional[Dict] = None) -> TextHistory:
        return self._v().reset([seed], [options])[0]

    def step(self, text_history: TextHistory):
        return self._v().step([text_history])[0]

    def close(self) -> None:
        if self._vec is not None:
            self._vec.close()

    def copy(self):
        return type(self)(self.vocab, self.require_words_in_vocab, self.bad_word_reward)


class ReformatWordleEnvironment(WordleEnvironment):
    """LM-facing env (env.py:28-37).  Accepts either a `WordleEnvironment` (reference call shape) or
    the constructor arguments directly."""
    _reformat = True

    def __init__(self, env_or_vocab, require_words_in_vocab: bool = True, bad_word_reward: float = -1.0):
        if isinstance(env_or_vocab, WordleEnvironment):
            e = env_or_vocab
            super().__init__(e.vocab, e.require_words_in_vocab, e.bad_word_reward)
        else:
            super().__init__(env_or_vocab, require_words_in_vocab, bad_word_reward)

    def copy(self):
        return ReformatWordleEnvironment(self.vocab, self.require_words_in_vocab, self.bad_word_reward)
