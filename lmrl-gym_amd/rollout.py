"""Lock-step Wordle rollouts entirely on the device: GPT-2 policy + sampler + env + token bookkeeping.

This is the MI355X counterpart of `interact_environment(env, GPT2PPOPolicy(...), bsize=B)`
(LLM_RL/environment.py:154-207 with LLM_RL/algorithms/ppo/gpt2/interface.py:507-546): the same per-turn
sequence — policy generates an action until '\\n' / max_new_tokens, env.step, observation appended — but
for B envs at once with no host round trip, a persistent KV cache (only a turn's new tokens are forwarded)
and the observation injected as token ids.  The resulting per-env records are TokenTrajectory fields.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from .envs import wordle as W
from .gpt2 import GPT2Engine, SampleParams

def gpt2_byte_token_id(byte: int) -> int:
    """Id of the single-byte token of `byte` in GPT-2's byte-level BPE vocabulary, from the PUBLISHED construction (openai/gpt-2
    encoder.py `bytes_to_unicode` + vocabulary order): the 188 printable bytes '!'..'~', 0xA1..0xAC, 0xAE..0xFF take ids 0..187 in that
    order, the remaining 68 bytes (0x00..0x20, 0x7F..0xA0, 0xAD) ids 188..255 in increasing byte order.  Hence 'a'..'z' = 64..89,
    ':' = 25, '\n' = 198, ' ' = 220."""
    printable = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    if byte in printable:
        return printable.index(byte)
    return 188 + [b for b in range(256) if b not in printable].index(byte)


# Ids of the MERGED tokens (' a'..' z', 'Word', 'le') = 256 + their rank in merges.txt: quoted from memory of the public vocabulary and
# NOT verifiable in the build container (no tokenizer files offline) — the single-byte ids above are derived, these are
# not.  `WordleTokenTable.from_tokenizer` is the authoritative path whenever a tokenizer is available (scripts/harness.py uses it when
# transformers finds the gpt2 files); for random-init synthetic benchmarks only injectivity matters.
_GPT2_SP_LETTERS = [257, 275, 269, 288, 304, 277, 308, 289, 1312, 474, 479, 300, 285, 299, 267, 279, 10662, 374, 264, 256,
                    334, 410, 266, 2124, 331, 1976]


class _CTokens(ctypes.Structure):
    _fields_ = [("newline", ctypes.c_int32), ("pad", ctypes.c_int32), ("letter_first", ctypes.c_int32 * 26),
                ("letter_sp", ctypes.c_int32 * 26), ("sym_first", ctypes.c_int32 * 3), ("sym_sp", ctypes.c_int32 * 3),
                ("header", ctypes.c_int32 * 8), ("n_header", ctypes.c_int32)]


class _CTraj(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("tokens", "is_action", "reward", "n_tok", "gen", "gen_len", "gen_active",
                                               "env_done", "pend_newline", "n_steps", "ep_reward")]


def classify_token_string(s: str) -> int:
    """Token class word of csrc/wordle_tokens.hip for the decoded string of one token."""
    letters = [c for c in s if "a" <= c <= "z"]
    other = [c for c in s if not ("a" <= c <= "z") and not c.isspace()]
    if other or len(letters) > 5:
        return 7 << 25
    idx = [i for i, c in enumerate(s) if "a" <= c <= "z"]
    is_w = [c.isspace() and c != " " for c in s]
    if not idx:
        return (1 << 28) if any(is_w) else 0
    first, last = idx[0], idx[-1]
    if any(is_w[first:last + 1]):
        return 7 << 25
    word = 0
    for k, c in enumerate(letters):
        word |= (ord(c) - 97) << (5 * k)
    return word | (len(letters) << 25) | ((1 << 28) if any(is_w[:first]) else 0) | ((1 << 29) if any(is_w[last + 1:]) else 0)


@dataclass
class WordleTokenTable:
    newline: int
    pad: int
    letter_first: List[int]
    letter_sp: List[int]
    sym_first: List[int]   # g, y, b
    sym_sp: List[int]
    header: List[int]
    strings: dict = field(default_factory=dict)   # id -> decoded string (for the class table)

    @classmethod
    def default_gpt2(cls, pad: int = 50257) -> "WordleTokenTable":
        """pad = 50257: the `<|pad|>` the task scripts add to the GPT-2 tokenizer (train_ppo_gpt2.py:124-126) — the first id after the vocabulary."""
        lf = [gpt2_byte_token_id(97 + i) for i in range(26)]          # 'a'..'z' = 64..89 (derived, see gpt2_byte_token_id)
        ls = list(_GPT2_SP_LETTERS)
        nl, colon = gpt2_byte_token_id(10), gpt2_byte_token_id(58)    # 198, 25
        t = cls(newline=nl, pad=pad, letter_first=lf, letter_sp=ls,
                sym_first=[lf[6], lf[24], lf[1]], sym_sp=[ls[6], ls[24], ls[1]], header=[26449, 293, colon, nl])
        t.strings = {198: "\n", 26449: "Word", 293: "le", 25: ":"}
        for i in range(26):
            t.strings[lf[i]] = chr(97 + i)
            t.strings[ls[i]] = " " + chr(97 + i)
        return t

    @classmethod
    def from_tokenizer(cls, tokenizer, pad: Optional[int] = None) -> "WordleTokenTable":
        one = lambda s: (lambda ids: ids[0] if len(ids) == 1 else (_ for _ in ()).throw(
            ValueError(f"{s!r} is not a single token: {ids}")))(tokenizer.encode(s))
        lf = [one(chr(97 + i)) for i in range(26)]
        ls = [one(" " + chr(97 + i)) for i in range(26)]
        header = list(tokenizer.encode("Wordle:\n"))
        if len(header) > 8:
            raise ValueError("header longer than 8 tokens")
        t = cls(newline=one("\n"), pad=tokenizer.pad_token_id if pad is None else pad, letter_first=lf, letter_sp=ls,
                sym_first=[lf[6], lf[24], lf[1]], sym_sp=[ls[6], ls[24], ls[1]], header=header)
        vocab = len(tokenizer)
        t.strings = {i: tokenizer.decode([i]) for i in range(vocab)}
        return t

    def token_class(self, vocab: int) -> np.ndarray:
        """uint32 [vocab]; ids without a known string are 'invalid-making' (7 << 25) — a full tokenizer fills all."""
        out = np.full(vocab, 7 << 25, dtype=np.uint32)
        for i, s in self.strings.items():
            if 0 <= i < vocab:
                out[i] = classify_token_string(s)
        return out

    def c_struct(self) -> _CTokens:
        c = _CTokens()
        c.newline, c.pad = self.newline, self.pad
        for i in range(26):
            c.letter_first[i] = self.letter_first[i]
            c.letter_sp[i] = self.letter_sp[i]
        for i in range(3):
            c.sym_first[i] = self.sym_first[i]
            c.sym_sp[i] = self.sym_sp[i]
        for i, h in enumerate(self.header):
            c.header[i] = h
        c.n_header = len(self.header)
        return c

    def encode_text(self, s: str) -> List[int]:
        """Canonical ids of Wordle-alphabet text ('Wordle:\\n', 's t a r e\\n', 'g y b b y\\n', '\\n')."""
        out: List[int] = []
        for line in s.splitlines(keepends=True):
            body = line[:-1] if line.endswith("\n") else line
            if body == "Wordle:":
                out += self.header[:-1]
            elif body:
                parts = body.split(" ")
                for k, p in enumerate(parts):
                    assert len(p) == 1 and "a" <= p <= "z", f"not Wordle-alphabet text: {line!r}"
                    out.append(self.letter_first[ord(p) - 97] if k == 0 else self.letter_sp[ord(p) - 97])
            if line.endswith("\n"):
                out.append(self.newline)
        return out


class WordleRolloutEngine:
    """B lock-step Wordle episodes driven by a GPT-2 policy on one GPU."""

    def __init__(self, engine: GPT2Engine, vocab: W.Vocabulary, batch: int, tokens: Optional[WordleTokenTable] = None,
                 max_new_tokens: int = 6, require_words_in_vocab: bool = True, bad_word_reward: float = -10.0,
                 traj_cap: int = 128, share_header: bool = True, value_engine: Optional[GPT2Engine] = None, q1_head: Optional[dict] = None,
                 q2_head: Optional[dict] = None, beta: float = 0.0, session_flags: int = 0):
        """`session_flags`: lmrl_gpt2_forward variants (gpt2.FWD_*) of this engine's KV sessions (A/B hooks; results are bit-identical).
        `value_engine` + `q1_head` (+ `q2_head`) + `beta` turn the policy into the ILQL value policy
        (`GPT2ValueRLGeneration`, value_rl_base/gpt2/generation.py:97-119): `engine` is pi_beta, `value_engine` the value base
        whose last hidden state feeds the Q heads (`policies.heads_to_engine_layout` dicts); logits = pi_beta + beta * min(Q1, Q2)."""
        import torch
        t = torch
        self.eng, self.vocab, self.B = engine, vocab, batch
        self._twin_args = dict(tokens=tokens, max_new_tokens=max_new_tokens, require_words_in_vocab=require_words_in_vocab,
                               bad_word_reward=bad_word_reward, traj_cap=traj_cap, share_header=share_header, value_engine=value_engine,
                               q1_head=q1_head, q2_head=q2_head, beta=beta, session_flags=session_flags)
        self._lanes = None           # text_env_eval(concurrent=n): [(engine, stream)], this engine first
        self.episodes = 0            # eager episodes run by text_env_eval over this engine's life: part of the sampler stream key
        # the pad id is the first id AFTER the policy's vocabulary, as in the reference: the task scripts add `<|pad|>` to the tokenizer (id 50257,
        # train_ppo_gpt2.py:124-126) and the model forces every logit >= unpadded_vocab_size to -inf (ppo/gpt2/interface.py:330), so a policy can never
        # draw it.  Here the sampler draws from [0, cfg.vocab) and the engine's embedding table has zero rows from cfg.vocab on
        self.tokens = tokens or WordleTokenTable.default_gpt2(pad=engine.cfg.vocab)
        self.max_new, self.cap = max_new_tokens, traj_cap
        self.dev = engine.device
        self._L = _lib.lib()
        self.env = W.VectorWordleEnv(vocab, require_words_in_vocab, bad_word_reward)
        self.env._alloc(batch)
        self.ses = engine.session(batch, traj_cap, int(session_flags))
        # every env starts from the same header text: its K/V are computed once per episode on a 1-env session and broadcast
        self.share_header = share_header
        self.ses1 = engine.session(1, 16) if share_header else None
        self.veng, self.q1, self.q2, self.beta = value_engine, q1_head, q2_head, float(beta)
        assert (value_engine is None) == (q1_head is None), "value_engine and q1_head come together"
        self.vses = value_engine.session(batch, traj_cap, int(session_flags)) if value_engine is not None else None
        self.dual_stream = value_engine is not None          # ILQL value policy: the value base's forwards on a second HIP stream (episode_phases)
        if value_engine is not None:
            import torch as _t
            self._aux_stream = _t.cuda.Stream(device=self.dev)
            self._ev_fork, self._ev_join = _t.cuda.Event(), _t.cuda.Event()
        self.vses1 = value_engine.session(1, 16) if value_engine is not None and share_header else None
        self.qh = [t.zeros(batch, value_engine.cfg.d_model, dtype=t.bfloat16, device=self.dev) for _ in range(2)] if value_engine is not None else None
        ct = self.tokens.c_struct()
        cls = np.ascontiguousarray(self.tokens.token_class(engine.cfg.vocab))
        self._tok = self._L.lmrl_wordle_tok_create(ctypes.byref(ct), cls.ctypes.data, engine.cfg.vocab, max_new_tokens, traj_cap)
        if not self._tok:
            raise _lib.LmrlError(self._L.lmrl_last_error().decode())
        B, G, cap = batch, max_new_tokens, traj_cap
        z = lambda *shape, dt: t.zeros(*shape, dtype=dt, device=self.dev)
        self.traj = dict(tokens=z(B, cap, dt=t.int32), is_action=z(B, cap, dt=t.uint8), reward=z(B, cap, dt=t.float32),
                         n_tok=z(B, dt=t.int32), gen=z(B, G, dt=t.int32), gen_len=z(B, dt=t.int32), gen_active=z(B, dt=t.uint8),
                         env_done=z(B, dt=t.uint8), pend_newline=z(B, dt=t.uint8), n_steps=z(B, dt=t.int32),
                         ep_reward=z(B, dt=t.float32))
        self._ctraj = _CTraj(*[self.traj[n].data_ptr() for n, _ in _CTraj._fields_])
        self.chunk_tok, self.chunk_cnt = z(B * 8, dt=t.int32), z(B, dt=t.int32)
        self.next_tok, self.next_cnt = z(B, dt=t.int32), z(B, dt=t.int32)
        self.guess, self.active = z(B, dt=t.int32), z(B, dt=t.uint8)
        self.steer = z(6, B, dt=t.int32)
        self.sample_step = 0

    def close(self):
        if getattr(self, "_tok", None):
            self._L.lmrl_wordle_tok_destroy(self._tok)
            self._tok = None
        self.env.close()
        for twin, _ in (self._lanes or [])[1:]:
            twin.close()
        self._lanes = None

    def _ck(self, rc, what):
        _lib.check(rc, what)

    def run_episode(self, seeds: np.ndarray, temperature: float = 1.0, top_k: int = 0, sample_seed: int = 0,
                    scripted_guesses=None, steer_strength: float = 0.0, n_turns: int = W.N_TRIES, epoch=None, sampler: str = "philox", top_p: float = 0.0):
        """One full episode for all B envs (asynchronous: returns after enqueueing; read results after a sync).

        scripted_guesses: optional int32 device tensor [n_turns][B] of packed guesses; with steer_strength > 0 the
        sampler is steered towards spelling them (synthetic-workload hook; every logit is still computed and sampled).
        """
        for _ in self.episode_phases(seeds, temperature, top_k, sample_seed, scripted_guesses, steer_strength, n_turns, epoch, sampler, top_p):
            pass
        return self.traj

    # ---- hipGraph mode: the whole episode (~3400 kernel launches) is captured once and replayed with one host call.
    # Everything that changes between episodes lives in device memory: env seeds, scripted guesses and the sampler's
    # `epoch` word (4th Philox counter word), so replays draw fresh noise and start from fresh env seeds.
    def capture_episode(self, temperature: float = 1.0, top_k: int = 0, sample_seed: int = 0, steer_strength: float = 0.0,
                        n_turns: int = W.N_TRIES, scripted: bool = False, top_p: float = 0.0):
        import torch
        t = torch
        self.g_seeds = t.zeros(self.B, dtype=t.int64, device=self.dev)
        self.g_guesses = t.zeros((n_turns, self.B), dtype=t.int32, device=self.dev) if scripted else None
        self.g_epoch = t.zeros(1, dtype=t.int32, device=self.dev)
        kw = dict(temperature=temperature, top_k=top_k, top_p=top_p, sample_seed=sample_seed, scripted_guesses=self.g_guesses,
                  steer_strength=steer_strength, n_turns=n_turns, epoch=self.g_epoch)
        self.run_episode(self.g_seeds, **kw)            # eager warm-up: one-time attribute / table initialisation
        t.cuda.synchronize()
        self.sample_step = 0
        self.graph = t.cuda.CUDAGraph()
        # thread-local capture mode: other threads of the process (e.g. the RCCL watchdog of torch.distributed) may keep
        # calling HIP APIs while this thread captures
        with t.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.run_episode(self.g_seeds, **kw)
        return self.graph

    def replay_episode(self, seeds_d, guesses_d=None):
        """seeds_d int64 [B] and guesses_d int32 [n_turns][B] are device tensors (device-to-device copies, no host sync)."""
        self.g_seeds.copy_(seeds_d, non_blocking=True)
        if guesses_d is not None:
            self.g_guesses.copy_(guesses_d, non_blocking=True)
        self.g_epoch.add_(1)
        self.graph.replay()
        return self.traj

    def episode_phases(self, seeds: np.ndarray, temperature: float = 1.0, top_k: int = 0, sample_seed: int = 0,
                       scripted_guesses=None, steer_strength: float = 0.0, n_turns: int = W.N_TRIES, epoch=None, sampler: str = "philox",
                       top_p: float = 0.0):
        """Generator form of `run_episode`: enqueues one phase (a model forward + its sampling / env bookkeeping) per
        `next()`, so a host loop can interleave several engines on different HIP streams.

        sampler="jax": the reference's random stream instead of the package's Philox stream — a policy freshly built with
        `prng_key=jax.random.PRNGKey(sample_seed)` splits its key once per turn (one `act()` per lock-step turn,
        ppo/gpt2/interface.py:524-526) and HF-Flax `_sample` once per generated token; every token of the batch is one
        `jax.random.categorical(key, logits[B, V])` (jax_prng.py, csrc/threefry.h).  Eager launches only (keys are host-walked)."""
        from . import jax_prng
        from .gpt2 import RNG_JAX
        assert sampler in ("philox", "jax")
        pol_key = jax_prng.prng_key(sample_seed) if sampler == "jax" else None
        L, sp, tr, B = self._L, _lib.stream_ptr(), ctypes.byref(self._ctraj), self.B
        self.env.reset_device(seeds if not isinstance(seeds, np.ndarray) else np.asarray(seeds, dtype=np.uint64))
        pairs = [(self.ses, self.ses1)] + ([(self.vses, self.vses1)] if self.vses is not None else [])     # (pi_beta), (value base)

        def both(fn):
            """fn(ses, ses1) for pi_beta and, for the ILQL value policy, for the value base — the two transformers are independent between two sampling
            steps, so the value base's forward runs on a second HIP stream (fork / join by events; also inside a captured graph): two full-batch
            chains of latency-bound kernels fill each other's idle CUs (two independent 1024-env chains measured 1.24 x the throughput of one, §6b)."""
            if self.vses is None or not self.dual_stream:
                for ses, ses1 in pairs:
                    fn(ses, ses1)
                return
            import torch
            main = torch.cuda.current_stream(self.dev)
            self._ev_fork.record(main)
            self._aux_stream.wait_event(self._ev_fork)
            with torch.cuda.stream(self._aux_stream):
                fn(self.vses, self.vses1)
                self._ev_join.record(self._aux_stream)
            fn(self.ses, self.ses1)
            main.wait_event(self._ev_join)
        for ses, _ in pairs:
            ses.reset()
        self._ck(L.lmrl_wordle_tok_begin(self._tok, tr, _lib.ptr(self.chunk_tok), _lib.ptr(self.chunk_cnt), B, sp), "tok_begin")
        def header(ses, ses1):
            if self.share_header:
                ses1.reset()
                ses1.forward(self.chunk_tok[:8], self.chunk_cnt[:1], 8)      # env 0's header chunk = everybody's header chunk
                ses.broadcast_prefix_from(ses1, len(self.tokens.header))
            else:
                ses.forward(self.chunk_tok, self.chunk_cnt, 8)
        both(header)
        yield
        logits_out = None
        top_p = float(top_p) if top_p is not None and 0.0 < float(top_p) < 1.0 and temperature > 0 else 0.0
        if top_k > 0 or top_p > 0.0:
            # the warpers (TopK / TopP logits warpers of HF generate, train_ppo_gpt2.py:98-99,218-227) select on MATERIALISED logits: one buffer per engine,
            # allocated once — its address is baked into a captured episode graph
            import torch
            if getattr(self, "_warp_logits", None) is None:
                self._warp_logits = torch.empty(B, self.eng.cfg.vocab_padded, dtype=torch.float32, device=self.dev)
            logits_out = self._warp_logits
        steered = scripted_guesses is not None and steer_strength != 0.0
        for turn in range(n_turns):
            tok_keys = None
            if pol_key is not None:
                pol_key, new_key = jax_prng.split(pol_key)
                tok_keys = jax_prng.SampleKeys(new_key)
            if steered:     # one launch spells the whole scripted guess of this turn
                self._ck(L.lmrl_wordle_tok_steer(self._tok, _lib.ptr(scripted_guesses[turn]), -1, _lib.ptr(self.steer), B, sp), "tok_steer")
            for k in range(self.max_new):
                steer = self.steer[min(k, 5)] if steered else None
                if tok_keys is not None:
                    p = SampleParams(temperature, top_k, jax_prng.key_to_seed(tok_keys.next()), self.sample_step, steer_strength, self.beta,
                                     self.tokens.pad, None, top_p, RNG_JAX)
                else:
                    p = SampleParams(temperature, top_k, sample_seed, self.sample_step, steer_strength, self.beta, self.tokens.pad,
                                     _lib.ptr(epoch), top_p)
                self.sample_step += 1
                qops = [None, None]
                if self.vses is not None:     # Q heads on the value base's last hidden state: relu(dense1) here, dense2 inside the sampler
                    dv = self.veng.cfg.d_model
                    for i, head in enumerate((self.q1, self.q2)):
                        if head is not None:
                            self._ck(L.lmrl_gemm_bf16(_lib.ptr(self.vses.last_hidden), _lib.ptr(head["w1"]), _lib.ptr(head["b1"]), _lib.ptr(self.qh[i]),
                                                      B, dv, dv, dv, dv, dv, 4, sp), "q head dense1 + relu")
                            qops[i] = (self.qh[i], head["w2"], head["b2"])
                self.ses.sample(p, steer_tok=steer, active=self.traj["gen_active"], logits_out=logits_out, q1=qops[0], q2=qops[1],
                                want_logprob=False)
                self._ck(L.lmrl_wordle_tok_accept(self._tok, tr, _lib.ptr(self.ses.token), k, _lib.ptr(self.next_tok),
                                                  _lib.ptr(self.next_cnt), None, B, sp), "tok_accept")
                if k < self.max_new - 1:
                    both(lambda ses, _: ses.forward(self.next_tok, self.next_cnt, 1))
                    yield
            self._ck(L.lmrl_wordle_tok_guess(self._tok, tr, _lib.ptr(self.guess), _lib.ptr(self.active), B, sp), "tok_guess")
            self.env.step_device(self.guess, self.active)
            self._ck(L.lmrl_wordle_tok_observe(self._tok, tr, _lib.ptr(self.env.obs), _lib.ptr(self.env.reward), _lib.ptr(self.env.flags),
                                               _lib.ptr(self.chunk_tok), _lib.ptr(self.chunk_cnt), B, sp), "tok_observe")
            if turn < n_turns - 1:
                both(lambda ses, _: ses.forward(self.chunk_tok, self.chunk_cnt, 8))
            yield

    def token_trajectories(self):
        """Host copies as (tokens int32[t], is_action bool[t], reward float32[t], done bool) per env —
        the fields of LLM_RL.environment.TokenTrajectory."""
        tok = self.traj["tokens"].cpu().numpy(); ia = self.traj["is_action"].cpu().numpy().astype(bool)
        rw = self.traj["reward"].cpu().numpy(); n = self.traj["n_tok"].cpu().numpy(); dn = self.traj["env_done"].cpu().numpy().astype(bool)
        return [(tok[b, :n[b]].copy(), ia[b, :n[b]].copy(), rw[b, :n[b]].copy(), bool(dn[b])) for b in range(self.B)]

    # ---- the online-RL hand-over: the finished episode as PPO data, on the device -------------------------------------
    def load_params(self, params) -> None:
        """The trainer's fp32 parameters into the policy engine IN PLACE (`GPT2Engine.load_params`): captured episodes stay valid (they read the same
        weight buffers; the shared header's K/V are recomputed at every episode start)."""
        self.eng.load_params(params)

    def ppo_records(self, n: Optional[int] = None):
        """The episode record as `algorithms.ppo_device.PPORecords` — views of the engine's own buffers (no copy; valid until the next episode
        is enqueued): one single-trajectory chain per env (the first `n` envs), exactly the `TextTrajectoryChain(text_trajectory, None)` the task
        script builds from a rollout (llm_rl_scripts/wordle/ppo/train_ppo_gpt2.py:317-342) after `TokenTrajectory.from_text_trajectory` — reward on
        the last token of every action, `done` from the env."""
        from .algorithms.ppo_device import PPORecords
        tr, n = self.traj, self.B if n is None else int(n)
        return PPORecords(tr["tokens"][:n], tr["is_action"][:n], tr["reward"][:n], tr["n_tok"][:n], tr["env_done"][:n])

    def ppo_data(self, inference, *, gamma: float, lam: float, kl_weight: float, max_length: Optional[int] = None, n: Optional[int] = None, **kw):
        """`ppo_dataset_loader` of the task script on the episode this engine just ran (train_ppo_gpt2.py:301-353 -> ppo/base_interface.py:464-669),
        without host text, re-tokenisation or materialised logits -> (DevicePPODataset, all_kls device tensor); see `ppo_device.ppo_data_from_records`.
        The script's length rule (train_ppo_gpt2.py:323-341: an episode whose token count reaches `max_length` loses its last (action, observation)
        pairs, their discounted reward goes to the previous action, `done` becomes False; episodes left with fewer than three texts are skipped) is
        applied to the records on the device whenever `max_length` does not exceed the longest possible episode (`ppo_device.truncate_turns`); the
        dataset then has one row per KEPT episode, in env order."""
        from .algorithms.ppo_device import ppo_data_from_records, truncate_turns
        rec = self.ppo_records(n)
        longest = min(self.cap, len(self.tokens.header) + W.N_TRIES * (self.max_new + 1 + 6))   # header + 6 x (action + forced '\n' + 'g y b b y\n')
        if max_length is not None and max_length <= longest:
            rec, info = truncate_turns(rec, int(max_length), gamma)
            if kw.get("timings") is not None:
                kw["timings"].update(episodes_shortened=info["shortened"], episodes_skipped=info["skipped"])
        return ppo_data_from_records(inference, rec, gamma=gamma, lam=lam, kl_weight=kl_weight, max_length=max_length, **kw)

    def ppo_rollouts(self, inference, n_rollouts: int, seed_generator=None, *, gamma: float, lam: float, kl_weight: float, max_length: Optional[int] = None,
                     use_advantage_whitening: bool = True, temperature: float = 1.0, sample_seed: int = 0, use_graph: Optional[bool] = None,
                     scripted_guesses_fn=None, steer_strength: float = 0.0, timings: Optional[dict] = None, top_k: int = 0, top_p: float = 0.0, **kw):
        """One data-collection round of the online PPO loop on the device: `text_env_eval(n_rollouts, bsize=B)` + `ppo_dataset_loader`
        (train_ppo_gpt2.py:301-353) -> (DevicePPODataset over all rollouts, all_kls, summary).  Per episode batch: the lock-step episode, then
        its PPO data while the record is still in the engine's buffers; advantages are whitened once over the action tokens of ALL rollouts of
        the round (ppo/base_interface.py:609-615 whitens over every chain handed to the call), and `summary` has `text_env_eval`'s shape, from the
        per-env counters (no text is built)."""
        import torch
        from .algorithms.ppo_device import DevicePPODataset
        from . import dist as D
        n_batches = -(-n_rollouts // self.B)
        seeds_all = np.zeros((n_batches, self.B), dtype=np.uint64)
        for k in range(n_batches):
            n_k = min(n_rollouts - k * self.B, self.B)
            seeds_all[k, :n_k] = [next(seed_generator) for _ in range(n_k)] if seed_generator is not None else np.random.randint(0, 2 ** 31 - 1, size=n_k)
        scripted = scripted_guesses_fn is not None
        key = (float(temperature), int(sample_seed), float(steer_strength), scripted, int(top_k), float(top_p or 0.0))
        want_graph = (getattr(self, "_eval_graph_key", None) == key or n_batches >= 4) if use_graph is None else bool(use_graph)
        seeds_dev = torch.from_numpy(seeds_all.view(np.int64)).to(self.dev)
        if want_graph and getattr(self, "_eval_graph_key", None) != key:
            self.capture_episode(temperature=temperature, sample_seed=sample_seed, steer_strength=steer_strength, scripted=scripted, top_k=top_k,
                                 top_p=top_p or 0.0)
            self._eval_graph_key = key
        parts, kls, stats = [], [], []
        ev = lambda: (lambda e: (e.record(), e)[1])(torch.cuda.Event(enable_timing=True))
        t_roll = 0.0
        for k in range(n_batches):
            n_k = min(n_rollouts - k * self.B, self.B)
            g = scripted_guesses_fn(k) if scripted else None
            e0 = ev() if timings is not None else None
            if want_graph:
                self.replay_episode(seeds_dev[k], g)
            else:
                self.run_episode(seeds_all[k], temperature=temperature, sample_seed=sample_seed + (self.episodes << 20), scripted_guesses=g,
                                 steer_strength=steer_strength, top_k=top_k, top_p=top_p or 0.0)
                self.episodes += 1
            if timings is not None:
                e1 = ev()
            ds, kl = self.ppo_data(inference, gamma=gamma, lam=lam, kl_weight=kl_weight, max_length=max_length, n=n_k, use_advantage_whitening=False,
                                   timings=timings, **kw)
            if timings is not None:
                torch.cuda.synchronize()
                t_roll += e0.elapsed_time(e1)
            parts.append(ds); kls.append(kl)
            stats.append(np.stack([self.traj[name][:n_k].cpu().numpy().astype(np.float64) for name in ("ep_reward", "env_done", "n_steps")]))
        ds = DevicePPODataset.concat(parts)
        if use_advantage_whitening:
            adv = ds.old_advantages
            ds.old_advantages = D.whiten_distributed(adv.view(-1), ds.should_take_action.view(-1), shift_mean=True).view(adv.shape)
        st = np.concatenate(stats, axis=1)
        summ = lambda x: dict(mean=np.mean(x), std=np.std(x), min=np.min(x), max=np.max(x))
        summary = dict(reward=summ(st[0].astype(np.float32)), done=summ(st[1].astype(np.float32)), length=summ(st[2].astype(np.int64)))
        if timings is not None:
            timings["rollout_ms"] = timings.get("rollout_ms", 0.0) + t_roll
        return ds, (kls[0] if len(kls) == 1 else torch.cat(kls)), summary

    # ---- the script-level call on the device engine -----------------------------------------------------------------
    def interactions(self, decode=None):
        """The finished episode as `List[List[InteractionTransition]]` — what `interact_environment` returns for the same
        rollout (LLM_RL/environment.py:154-207): one transition per env step with the pre-action, post-action and
        post-transition histories, the step reward and the done flag.  `decode(ids) -> str` defaults to the token table."""
        return self._build_interactions(self.snapshot_records(), decode)

    def snapshot_records(self):
        """Enqueue device -> pinned-host copies of the episode record behind the episode's kernels and return a handle; the record buffers may
        be overwritten by the next episode as soon as this returns (stream order).  `_build_interactions(handle)` waits for the copies only —
        the next episode's launches, enqueued in between, run on the device while the host builds the Python objects of this one."""
        import torch
        names = ("tokens", "is_action", "reward", "n_tok", "env_done")
        if getattr(self, "_pinned", None) is None:
            self._pinned = [{n: torch.empty(self.traj[n].shape, dtype=self.traj[n].dtype, pin_memory=True) for n in names} for _ in range(2)]
            self._pin_i = 0
        buf = self._pinned[self._pin_i]
        self._pin_i ^= 1
        for n in names:
            buf[n].copy_(self.traj[n], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return buf, ev

    def _build_interactions(self, handle, decode=None):
        """Host lists of one episode batch.  The cyclic garbage collector is paused while the ~25 k tuples of a 1024-env batch are created: none
        of them is garbage, and generation-2 scans of everything the caller already holds cost more than the construction itself."""
        import gc
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            return self._build_interactions_nogc(handle, decode)
        finally:
            if was_enabled:
                gc.enable()

    def _build_interactions_nogc(self, handle, decode=None):
        from .environment import InteractionTransition, Text
        dec = decode or (lambda ids: "".join(self.tokens.strings.get(int(i), "") for i in ids))
        buf, ev = handle
        ev.synchronize()
        tok = buf["tokens"].numpy(); ia = buf["is_action"].numpy().astype(bool)
        rw = buf["reward"].numpy(); ntok = buf["n_tok"].numpy(); done = buf["env_done"].numpy().astype(bool)
        # run boundaries of every env in one vectorised pass: positions where is_action changes (the Text items alternate header, action,
        # observation, action, ...); the few thousand distinct token runs of a batch are decoded once each (Text is immutable: shared)
        cache = {}

        def text_of(run, is_action):
            key = (run.tobytes(), is_action)
            t = cache.get(key)
            if t is None:
                t = cache[key] = Text(dec(run), is_action)
            return t
        out = []
        for b in range(self.B):
            n = int(ntok[b])
            iab = ia[b, :n]
            cuts = np.flatnonzero(iab[1:] != iab[:-1]) + 1
            bounds = [0] + cuts.tolist() + [n]
            texts = tuple(text_of(tok[b, bounds[k]:bounds[k + 1]], bool(iab[bounds[k]])) for k in range(len(bounds) - 1) if bounds[k + 1] > bounds[k])
            nt = len(texts)
            last_action = max((k for k in range(nt) if texts[k].is_action), default=-1)
            trans = []
            for k in range(nt):
                if texts[k].is_action:
                    # an action is always followed by its observation; reward sits on the action's last token (environment.py:370)
                    post = texts[: k + 2] if k + 1 < nt else texts[: k + 1]
                    trans.append(InteractionTransition(texts[:k], texts[: k + 1], post, float(rw[b, bounds[k + 1] - 1]), bool(done[b]) and k == last_action))
            out.append(trans)
        return out

    def _eval_lanes(self, n: int, main):
        """[(engine, stream)] for `text_env_eval(concurrent=n)`: this engine on the caller's stream + n - 1 twins (same model, vocabulary, batch and
        record layout; own sessions / env state / records) on their own streams.  Built once and kept."""
        import torch
        if self._lanes is None:
            self._lanes = [(self, None)]
        while len(self._lanes) < n:
            st = torch.cuda.Stream(device=self.dev)
            with torch.cuda.stream(st):
                twin = WordleRolloutEngine(self.eng, self.vocab, self.B, **dict(self._twin_args, tokens=self.tokens))
            self._lanes.append((twin, st))
        return [(self, main)] + self._lanes[1:n]

    def text_env_eval(self, n_rollouts: int, seed_generator=None, temperature: float = 1.0, top_k: int = 0, sample_seed: int = 0,
                      interaction_callback=None, decode=None, scripted_guesses_fn=None, steer_strength: float = 0.0, use_graph: Optional[bool] = None,
                      concurrent: int = 1, top_p: float = 0.0):
        """`text_env_eval(env, policy, n_rollouts, bsize=B)` (LLM_RL/environment.py:211-267) with env, policy and the whole
        lock-step loop on the device: ceil(n / B) episodes batches, the same (interactions, summary) return value.
        top_k / top_p: HF's TopK / TopP logits warpers (the policy_top_k / policy_top_p of the task scripts) — on the graph path too (round 5: the
        warpers' logits buffer is allocated once per engine; a graph is keyed by (top_k, top_p) as well).
        use_graph (the ILQL value policy included): the episode is captured into a hipGraph once per (temperature, sample_seed, steering) and replayed per
        batch — one host call instead of ~3400 launches; every replay draws fresh noise (the sampler's epoch word advances).  A capture costs
        two extra episodes (warm-up + capture), so the default (None) uses the graph only when this engine already holds one for the same key
        or the call runs >= 4 batches; True / False force it.  The two paths draw DIFFERENT noise for the same `sample_seed`: the graph's
        stream is keyed by (sample_seed, replay epoch), the eager one by (sample_seed + (episode counter << 20)) — the counter runs across calls
        (`self.episodes`), so neither repeats noise between calls; both are reproducible for a fixed call history.
        concurrent = n > 1 (graph path, >= 2 batches): n episode batches in flight at once, each a full lock-step batch of B envs with its own
        KV cache, env state and graph on its own HIP stream over the SAME weights (`_eval_lanes`).  One lock-step batch is a dependent chain of
        ~3400 launches, many of them too small to fill 256 CUs; two independent chains fill each other's gaps.  Batches are independent in the
        reference too (a fresh env reset + its own generate calls per batch), and the interactions come back in batch order.
        `scripted_guesses_fn(batch_id) -> int32 device tensor [n_turns][B]` + `steer_strength`: synthetic workloads (bench.py)."""
        import torch
        from collections import deque
        inter, rewards, dones, lengths = [], [], [], []

        def absorb(eng, handle, actual):
            for ep in eng._build_interactions(handle, decode)[:actual]:
                inter.append(ep)
                rewards.append(sum(t.reward for t in ep)); dones.append(ep[-1].done); lengths.append(len(ep))
                if interaction_callback is not None:
                    interaction_callback(ep)
        # episode batches are pipelined: batch k's record is copied to pinned host memory behind its kernels, batch k + 1 is enqueued, and only
        # then the host turns batch k into InteractionTransition lists — the Python work overlaps the device work of the next batch
        # the env seeds of ALL batches are drawn (in the reference's order: one per rollout) and uploaded once: a per-batch host -> device copy of a
        # pageable array waits for the stream, i.e. for the batch in flight, and the next replay could not be enqueued under it
        n_batches = -(-n_rollouts // self.B)
        seeds_all = np.zeros((n_batches, self.B), dtype=np.uint64)
        for k in range(n_batches):
            n_k = min(n_rollouts - k * self.B, self.B)
            seeds_all[k, :n_k] = [next(seed_generator) for _ in range(n_k)] if seed_generator is not None else np.random.randint(0, 2 ** 31 - 1, size=n_k)
        scripted = scripted_guesses_fn is not None
        key = (float(temperature), int(sample_seed), float(steer_strength), scripted, int(top_k), float(top_p or 0.0))
        graph_ok = True
        if use_graph is None:
            want_graph = graph_ok and (getattr(self, "_eval_graph_key", None) == key or n_batches >= 4)
        else:
            want_graph = bool(use_graph) and graph_ok
        main = torch.cuda.current_stream(self.dev)
        lanes = [(self, main)]
        if concurrent > 1 and want_graph and n_batches >= 2:
            lanes = self._eval_lanes(min(int(concurrent), n_batches), main)
        # scripted batches: every batch's guesses are taken BEFORE anything is enqueued, so that one wait per lane (below) orders the lanes behind
        # their producer — a per-batch wait on the caller's stream would queue a lane behind the episode lane 0 has in flight
        gs = [scripted_guesses_fn(k) for k in range(n_batches)] if scripted else None
        seeds_dev = None
        for _, st in lanes[1:]:
            st.wait_stream(main)
        if want_graph:
            seeds_dev = torch.from_numpy(seeds_all.view(np.int64)).to(self.dev)
            for l, (e, st) in enumerate(lanes):
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    if getattr(e, "_eval_graph_key", None) != key:
                        # lanes draw from different noise streams: the same (seed, epoch) key on two lanes would repeat one batch's noise in the next
                        e.capture_episode(temperature=temperature, sample_seed=(sample_seed + l * 0x9E3779B97F4A7C15) & (2 ** 64 - 1),
                                          steer_strength=steer_strength, scripted=scripted, top_k=top_k, top_p=top_p or 0.0)
                        e._eval_graph_key = key
        batch_id, launched, pending = 0, 0, deque()
        while launched < n_rollouts:
            actual = min(n_rollouts - launched, self.B)
            g = gs[batch_id] if scripted else None
            e, st = lanes[batch_id % len(lanes)]
            with torch.cuda.stream(st):
                if want_graph:
                    e.replay_episode(seeds_dev[batch_id], g)
                else:
                    e.run_episode(seeds_all[batch_id], temperature=temperature, top_k=top_k, top_p=top_p or 0.0, sample_seed=sample_seed + (self.episodes << 20),
                                  scripted_guesses=g, steer_strength=steer_strength)
                    self.episodes += 1
                handle = e.snapshot_records()
            batch_id += 1
            launched += actual
            pending.append((e, handle, actual))
            while len(pending) > len(lanes):
                absorb(*pending.popleft())
        while pending:
            absorb(*pending.popleft())
        for _, st in lanes[1:]:
            main.wait_stream(st)
        summ = lambda x: dict(mean=np.mean(x), std=np.std(x), min=np.min(x), max=np.max(x))
        return inter, dict(reward=summ(np.asarray(rewards, dtype=np.float32)), done=summ(np.asarray(dones, dtype=np.float32)), length=summ(lengths))

