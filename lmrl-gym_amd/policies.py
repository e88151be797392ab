"""Text-in / text-out policies on the HIP rollout engine — counterparts of `GPT2PPOPolicy`
(LLM_RL/algorithms/ppo/gpt2/interface.py:468-549, also the BC policy of the task scripts) and `GPT2ValuePolicy`
(LLM_RL/algorithms/value_rl_base/gpt2/interface.py:239-330).

`act(text_history, done)` keeps the reference contract: done slots get the eos string as prompt and `None` back, prompts
are LEFT-truncated to `max_input_length` tokens, generation stops at `eos_token_id` or `max_new_tokens`,
`out_str_process` post-processes the completion, and the result is `history + (Text(completion, True),)`.
This is the general (any tokenizer, any env) path with one host sync per generated token; the all-device Wordle loop is
`lmrl_gym_amd.rollout.WordleRolloutEngine`.
"""
from __future__ import annotations

import ctypes
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import _lib
from .environment import BatchedTextPolicy, Text, TextHistory, text_history_to_str
from . import jax_prng
from .gpt2 import FWD_RAGGED_ALWAYS, RNG_JAX, GPT2Engine, SampleParams


class _Generator:
    """Prefill (8-token chunks) + token-by-token decode for a fixed batch on one or two KV sessions."""

    def __init__(self, engines: Sequence[GPT2Engine], batch: int, tmax: int):
        import torch
        self.t = torch
        self.engines, self.B, self.tmax = list(engines), batch, tmax
        # sequences of a batch end at different steps: every forward of these sessions runs on the compacted live rows
        # (bit-identical results); a per-session flag, not process state
        self.sessions = [e.session(batch, tmax, flags=FWD_RAGGED_ALWAYS) for e in self.engines]
        self.dev = self.engines[0].device
        self.cached: List[List[int]] = [[] for _ in range(batch)]     # token ids whose K/V rows sit at positions [0, n) of every session
        self.prefilled_tokens = 0                                      # tokens actually forwarded by prefill() (diagnostic)

    def prefill(self, prompts: List[List[int]], reuse: bool = True):
        """Chunked prefill, 16 tokens per env per forward (one full MFMA query tile of the chunk-attention kernel).

        reuse=True keeps, per env, the K/V rows of the longest common prefix of this prompt and the previous call's prompt and forwards only
        the rest: a text env's history grows by an action and an observation per turn, so a turn costs its NEW tokens instead of the whole
        history (the reference re-encodes and re-runs the full prompt in every `act`, ppo/gpt2/interface.py:519-546).  A prompt that was
        left-truncated, re-tokenised differently or belongs to a new episode simply has a short common prefix."""
        t, B, C = self.t, self.B, 16
        keep = np.zeros(B, dtype=np.int32)
        if reuse:
            for b, p in enumerate(prompts):
                old, n, lim = self.cached[b], 0, min(len(self.cached[b]), len(p) - 1)      # at least one token is forwarded: its hidden
                while n < lim and old[n] == p[n]:                                       # state feeds the first sample
                    n += 1
                keep[b] = n
        for s in self.sessions:
            s.set_len(keep) if reuse else s.reset()
        maxlen = max(len(p) - int(k) for p, k in zip(prompts, keep))
        for c0 in range(0, maxlen, C):
            toks = np.zeros((B, C), dtype=np.int32)
            cnt = np.zeros(B, dtype=np.int32)
            for b, p in enumerate(prompts):
                seg = p[keep[b] + c0: keep[b] + c0 + C]
                toks[b, : len(seg)] = seg
                cnt[b] = len(seg)
            self.prefilled_tokens += int(cnt.sum())
            td, cd = t.from_numpy(toks.reshape(-1)).to(self.dev), t.from_numpy(cnt).to(self.dev)
            bound = max(min(int(k) + c0 + C, len(p)) for p, k in zip(prompts, keep))      # exact: the kept prefixes differ per env
            for s in self.sessions:
                s.forward(td, cd, C, len_bound_after=bound)
        self.cached = [list(p) for p in prompts]      # generation writes behind the prompt: those rows are re-derived from the next prompt


class GPT2PPOPolicy(BatchedTextPolicy):
    def __init__(self, engine: GPT2Engine, tokenizer, max_input_length: int = 256, max_new_tokens: int = 256, do_sample: bool = True,
                 temperature: Optional[float] = None, top_k: Optional[int] = None, top_p: Optional[float] = None,
                 eos_token_id: Optional[int] = None,
                 pad_token_id: Optional[int] = None, seed: int = 0, in_str_process: Optional[Callable[[str], str]] = None,
                 out_str_process: Optional[Callable[[str], str]] = None, reuse_kv: bool = True, sampler: str = "philox"):
        """sampler: "philox" (default) — the package's counter-based stream, keyed by (seed, act() call, row, column, token position);
        "jax" — the reference's stream: `seed` is the integer of `jax.random.PRNGKey(seed)` handed to the policy, split once per act()
        (ppo/gpt2/interface.py:524-526) and once per generated token (HF-Flax `_sample`), each token drawn as
        `jax.random.categorical(key, logits[B, V])` (lmrl_gym_amd/jax_prng.py + csrc/threefry.h; restated from jax 0.4.7's published
        algorithm, unverified against jax itself)."""
        assert sampler in ("philox", "jax")
        self.engine, self.tokenizer = engine, tokenizer
        self.sampler = sampler
        self.prng_key = jax_prng.prng_key(seed)
        self.reuse_kv = reuse_kv
        self.max_input_length, self.max_new_tokens = max_input_length, max_new_tokens
        self.temperature = (temperature if temperature is not None else 1.0) if do_sample else 0.0
        self.top_k = top_k or 0
        self.top_p = float(top_p) if top_p is not None and 0.0 < top_p < 1.0 and do_sample else 0.0
        self.eos = eos_token_id if eos_token_id is not None else getattr(tokenizer, "eos_token_id", None)
        self.pad = pad_token_id if pad_token_id is not None else getattr(tokenizer, "pad_token_id", 0)
        self.seed, self.calls = seed, 0
        self.in_str_process = in_str_process or (lambda x: x)
        self.out_str_process = out_str_process or (lambda x: x)
        self._gen: Optional[_Generator] = None

    # hooks the ILQL policy overrides
    def _engines(self) -> List[GPT2Engine]:
        return [self.engine]

    def _sample(self, gen: _Generator, params: SampleParams, active_d, logits_out):
        return gen.sessions[0].sample(params, active=active_d, logits_out=logits_out, want_logprob=False)

    def act(self, text_history: List[Optional[TextHistory]], done: Optional[List[bool]] = None) -> List[Optional[TextHistory]]:
        import torch
        B = len(text_history)
        if done is None:
            done = [False] * B
        eos_str = self.tokenizer.decode([self.eos]) if self.eos is not None else ""
        raw = [eos_str if d else self.in_str_process(text_history_to_str(h)) for h, d in zip(text_history, done)]
        prompts = []
        memo = self.__dict__.setdefault("_encode_memo", {})    # prompt string -> ids: envs with few distinct observations (Maze: 130) repeat them every turn
        if len(memo) > 65536:
            memo.clear()
        for s in raw:
            ids = memo.get(s)
            if ids is None:
                ids = list(self.tokenizer.encode(s))
                if len(ids) > self.max_input_length:       # Truncation.LEFT
                    ids = ids[len(ids) - self.max_input_length:]
                ids = memo[s] = ids if ids else [self.pad]
            prompts.append(ids)
        tmax = -(-(self.max_input_length + self.max_new_tokens + 16) // 16) * 16
        if self._gen is None or self._gen.B != B or self._gen.tmax != tmax:
            self._gen = _Generator(self._engines(), B, tmax)
        gen = self._gen
        return self._generate(gen, prompts, text_history, done, B)

    def _generate(self, gen, prompts, text_history, done, B):
        import torch
        gen.prefill(prompts, reuse=self.reuse_kv)
        self.calls += 1                                     # one random stream per act() call, like the per-call key split
        keys = None
        if self.sampler == "jax":                           # self.prng_key, new_key = jax.random.split(self.prng_key)
            self.prng_key, new_key = jax_prng.split(self.prng_key)
            keys = jax_prng.SampleKeys(new_key)
        # Generation loop without a host sync per token: live flags, the generated ids and the next decode inputs stay on the
        # device (`lmrl_gen_accept`); the host only peeks at the live flags every `sync_every` tokens to stop early.
        L = _lib.lib()
        cap = self.max_new_tokens
        active_d = torch.from_numpy(np.array([not d for d in done], dtype=np.uint8)).to(gen.dev)
        out_tok = torch.zeros((B, cap), dtype=torch.int32, device=gen.dev)
        out_len = torch.zeros(B, dtype=torch.int32, device=gen.dev)
        next_tok = torch.zeros(B, dtype=torch.int32, device=gen.dev)
        next_cnt = torch.zeros(B, dtype=torch.int32, device=gen.dev)
        logits_out = None
        if self.top_k > 0 or self.top_p > 0.0:
            logits_out = torch.empty(B, self.engine.cfg.vocab_padded, dtype=torch.float32, device=gen.dev)
        sync_every = 16
        for k in range(cap):
            if k % sync_every == 0 and k > 0 and not bool(active_d.any().item()):
                break
            if keys is not None:
                p = SampleParams(self.temperature, self.top_k, jax_prng.key_to_seed(keys.next()), k, 0.0, 0.0, self.pad, None, self.top_p, RNG_JAX)
            else:
                p = SampleParams(self.temperature, self.top_k, self.seed + (self.calls << 20), k, 0.0, 0.0, self.pad, None, self.top_p)
            tok, _ = self._sample(gen, p, active_d, logits_out)
            _lib.check(L.lmrl_gen_accept(_lib.ptr(tok), _lib.ptr(active_d), _lib.ptr(out_tok), _lib.ptr(out_len), _lib.ptr(next_tok),
                                         _lib.ptr(next_cnt), -1 if self.eos is None else int(self.eos), cap, B, _lib.stream_ptr()), "lmrl_gen_accept")
            if k + 1 < cap:
                for s_ in gen.sessions:
                    s_.forward(next_tok, next_cnt, 1)
        ot, ol = out_tok.cpu().numpy(), out_len.cpu().numpy()
        out_ids: List[List[int]] = [ot[b, : ol[b]].tolist() for b in range(B)]
        results: List[Optional[TextHistory]] = []
        for h, d, ids in zip(text_history, done, out_ids):
            if d:
                results.append(None)
            else:
                results.append(tuple(h) + (Text(self.out_str_process(self._decode_generation(ids)), True),))
        return results

    def _decode_generation(self, ids: List[int]) -> str:
        """The reference decodes generations with `batch_decode(..., skip_special_tokens=True)`
        (value_rl_base/base_interface.py:126, bc/core.py:102): special tokens (eos / pad) never reach the action Text.  An eos that
        is ordinary text (Wordle: eos = the '\n' token, train_ilql_gpt2.py:393) is not special and stays."""
        try:
            return self.tokenizer.decode(ids, skip_special_tokens=True)
        except TypeError:
            # minimal tokenizers without the keyword: drop what THEY declare special (`all_special_ids`; default: only the pad id —
            # an eos id is ordinary text unless the tokenizer says otherwise, exactly as '\n' is for the GPT-2 tokenizer)
            special = getattr(self.tokenizer, "all_special_ids", None)
            if special is None:
                special = [t for t in (getattr(self.tokenizer, "pad_token_id", None),) if t is not None]
            special = set(special)
            return self.tokenizer.decode([t for t in ids if t not in special])

    def set_params(self, engine: GPT2Engine) -> None:
        """PPOPolicy.set_params (ppo/base_interface.py:821-823): swap in freshly trained weights."""
        self.engine = engine
        self._gen = None


class GPT2ValuePolicy(GPT2PPOPolicy):
    """ILQL policy: logits = pi_beta_logits + beta * min(Q1, Q2)(h_value)   (value_rl_base/gpt2/generation.py:97-119).

    q heads: dicts with bf16 device tensors `w1` [d][d] ([out][in]), `w2` [vocab_padded][d] and f32 `b1` [d], `b2` [vocab_padded]
    (build them with `heads_to_engine_layout`)."""

    def __init__(self, pi_beta: GPT2Engine, value_base: GPT2Engine, q1_head: dict, q2_head: Optional[dict], beta: float, tokenizer, **kw):
        super().__init__(pi_beta, tokenizer, **kw)
        self.value_base, self.q1, self.q2, self.beta = value_base, q1_head, q2_head, beta

    def _engines(self):
        return [self.engine, self.value_base]

    def set_params(self, policy_params) -> None:
        """ValueRLPolicy.set_params (value_rl_base/gpt2/interface.py:322-330): `(pi_beta, base, q1_head, q2_head)` — engines for
        the two transformers, `heads_to_engine_layout` dicts for the heads."""
        pi_beta, base, q1, q2 = policy_params
        self.engine, self.value_base, self.q1, self.q2 = pi_beta, base, q1, q2
        self._gen = None

    def _sample(self, gen, params, active_d, logits_out):
        import torch
        L = _lib.lib()
        pi_ses, v_ses = gen.sessions
        d, B = self.value_base.cfg.d_model, gen.B
        ops = []
        for head in (self.q1, self.q2):
            if head is None:
                ops.append(None)
                continue
            qh = torch.empty(B, d, dtype=torch.bfloat16, device=gen.dev)
            _lib.check(L.lmrl_gemm_bf16(_lib.ptr(v_ses.last_hidden), _lib.ptr(head["w1"]), _lib.ptr(head["b1"]), _lib.ptr(qh), B, d, d, d, d, d,
                                        4, _lib.stream_ptr()), "lmrl_gemm_bf16(q head dense1 + relu)")
            ops.append((qh, head["w2"], head["b2"]))
        params.beta = self.beta
        return pi_ses.sample(params, active=active_d, logits_out=logits_out, q1=ops[0], q2=ops[1], want_logprob=False)


def heads_to_engine_layout(head_params: dict, vocab_padded: int, device) -> dict:
    """MLPHead params (`dense1.kernel` [d,d], `dense1.bias`, `dense2.kernel` [d,V], `dense2.bias`) -> bf16 [out][in] tensors
    padded to `vocab_padded` rows for the fused sampler."""
    import torch
    w1 = head_params["dense1.kernel"].t().contiguous().to(device, torch.bfloat16)
    V, d = head_params["dense2.kernel"].shape[1], head_params["dense2.kernel"].shape[0]
    w2 = torch.zeros(vocab_padded, d, dtype=torch.bfloat16, device=device)
    w2[:V] = head_params["dense2.kernel"].t().to(device, torch.bfloat16)
    b2 = torch.zeros(vocab_padded, dtype=torch.float32, device=device)
    b2[:V] = head_params["dense2.bias"].to(device, torch.float32)
    return dict(w1=w1, b1=head_params["dense1.bias"].to(device, torch.float32).contiguous(), w2=w2, b2=b2)
