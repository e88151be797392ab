"""GPT-2 on the HIP rollout engine: weights in engine layout, KV-cache sessions, fused sampling.

Host-side counterpart of what the reference obtains from JaxSeq (`GPT2Inference`, `load_train_state`,
`generate_from_str`; call sites LLM_RL/algorithms/ppo/gpt2/interface.py:507-546 and
llm_rl_scripts/wordle/ilql/train_ilql_gpt2.py:190-200).  All compute goes through
`lmrl_gpt2_forward` / `lmrl_lm_head_sample` (csrc/gpt2.hip, csrc/sampler.hip); torch only owns the HBM.
"""
from __future__ import annotations

import ctypes
import math
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np

from . import _lib


@dataclass
class GPT2Config:
    n_layer: int = 12
    n_head: int = 12
    d_model: int = 768
    d_ff: int = 3072
    vocab: int = 50257
    n_pos: int = 1024
    ln_eps: float = 1e-5
    initializer_range: float = 0.02

    @property
    def vocab_padded(self) -> int:
        return (self.vocab + 127) // 128 * 128

    @classmethod
    def gpt2_small(cls, vocab: int = 50257) -> "GPT2Config":
        return cls(12, 12, 768, 3072, vocab, 1024)

    @classmethod
    def gpt2_medium(cls, vocab: int = 50257) -> "GPT2Config":
        return cls(24, 16, 1024, 4096, vocab, 1024)

    @classmethod
    def gpt2_large(cls, vocab: int = 50257) -> "GPT2Config":
        return cls(36, 20, 1280, 5120, vocab, 1024)


class _CConfig(ctypes.Structure):
    _fields_ = [("n_layer", ctypes.c_int32), ("n_head", ctypes.c_int32), ("d_model", ctypes.c_int32),
                ("d_ff", ctypes.c_int32), ("vocab", ctypes.c_int32), ("vocab_padded", ctypes.c_int32),
                ("n_pos", ctypes.c_int32), ("ln_eps", ctypes.c_float)]


class SampleParams(ctypes.Structure):
    _fields_ = [("temperature", ctypes.c_float), ("top_k", ctypes.c_int32), ("seed", ctypes.c_uint64),
                ("step", ctypes.c_uint32), ("steer_strength", ctypes.c_float), ("beta", ctypes.c_float),
                ("pad_token", ctypes.c_int32), ("epoch_d", ctypes.c_void_p), ("top_p", ctypes.c_float), ("rng", ctypes.c_int32),
                ("flags", ctypes.c_int32)]


RNG_PHILOX, RNG_JAX = 0, 1       # lmrl_sample_params.rng
SAMPLE_WANT_LOGITS = 1           # lmrl_sample_params.flags: materialise logits_out even where the fused top-k path would not


class _CKVPrefix(ctypes.Structure):      # lmrl_kv_prefix (include/lmrl_amd.h)
    _fields_ = [("kv_d", ctypes.c_void_p), ("n_rows", ctypes.c_int32), ("tmax", ctypes.c_int32), ("row_d", ctypes.c_void_p),
                ("n_d", ctypes.c_void_p), ("order_d", ctypes.c_void_p)]


def init_hf_style_state_dict(cfg: GPT2Config, seed: int = 0) -> Dict[str, "torch.Tensor"]:
    """Random-init weights with HF GPT-2 names/shapes/statistics (Conv1D kernels are [in, out];
    normal(0, initializer_range); residual projections scaled by 1/sqrt(2*n_layer); LN = 1/0)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    std = cfg.initializer_range
    n = lambda *shape, s=std: torch.randn(*shape, generator=g) * s
    sd = {"wte.weight": n(cfg.vocab, cfg.d_model), "wpe.weight": n(cfg.n_pos, cfg.d_model),
          "ln_f.weight": torch.ones(cfg.d_model), "ln_f.bias": torch.zeros(cfg.d_model)}
    ps = std / math.sqrt(2 * cfg.n_layer)
    for l in range(cfg.n_layer):
        p = f"h.{l}."
        sd[p + "ln_1.weight"] = torch.ones(cfg.d_model); sd[p + "ln_1.bias"] = torch.zeros(cfg.d_model)
        sd[p + "attn.c_attn.weight"] = n(cfg.d_model, 3 * cfg.d_model); sd[p + "attn.c_attn.bias"] = torch.zeros(3 * cfg.d_model)
        sd[p + "attn.c_proj.weight"] = n(cfg.d_model, cfg.d_model, s=ps); sd[p + "attn.c_proj.bias"] = torch.zeros(cfg.d_model)
        sd[p + "ln_2.weight"] = torch.ones(cfg.d_model); sd[p + "ln_2.bias"] = torch.zeros(cfg.d_model)
        sd[p + "mlp.c_fc.weight"] = n(cfg.d_model, cfg.d_ff); sd[p + "mlp.c_fc.bias"] = torch.zeros(cfg.d_ff)
        sd[p + "mlp.c_proj.weight"] = n(cfg.d_ff, cfg.d_model, s=ps); sd[p + "mlp.c_proj.bias"] = torch.zeros(cfg.d_model)
    return sd


class GPT2Engine:
    """GPT-2 weights resident in HBM in engine layout + the C handle."""

    def __init__(self, cfg: GPT2Config, state_dict: Dict[str, "torch.Tensor"], device=None):
        import torch
        self.cfg = cfg
        self.device = device or _lib.require_gpu()
        self._L = _lib.lib()
        sd = {k[len("transformer."):] if k.startswith("transformer.") else k: v for k, v in state_dict.items()}
        bf = lambda t: t.to(self.device, torch.float32).to(torch.bfloat16).contiguous()
        f32 = lambda t: t.to(self.device, torch.float32).contiguous()
        wte = torch.zeros(cfg.vocab_padded, cfg.d_model, dtype=torch.bfloat16, device=self.device)
        wte[: cfg.vocab] = bf(sd["wte.weight"][: cfg.vocab])
        self.wte, self.wpe = wte, bf(sd["wpe.weight"])
        self.lnf_g, self.lnf_b = f32(sd["ln_f.weight"]), f32(sd["ln_f.bias"])
        self.layers = []
        ptrs = []
        for l in range(cfg.n_layer):
            p = f"h.{l}."
            t = [f32(sd[p + "ln_1.weight"]), f32(sd[p + "ln_1.bias"]),
                 bf(sd[p + "attn.c_attn.weight"].t()), f32(sd[p + "attn.c_attn.bias"]),      # [3d][d]
                 bf(sd[p + "attn.c_proj.weight"].t()), f32(sd[p + "attn.c_proj.bias"]),      # [d][d]
                 f32(sd[p + "ln_2.weight"]), f32(sd[p + "ln_2.bias"]),
                 bf(sd[p + "mlp.c_fc.weight"].t()), f32(sd[p + "mlp.c_fc.bias"]),            # [dff][d]
                 bf(sd[p + "mlp.c_proj.weight"].t()), f32(sd[p + "mlp.c_proj.bias"])]        # [d][dff]
            self.layers.append(t)
            ptrs += [x.data_ptr() for x in t]
        self._ptrs = (ctypes.c_void_p * len(ptrs))(*ptrs)
        cc = _CConfig(cfg.n_layer, cfg.n_head, cfg.d_model, cfg.d_ff, cfg.vocab, cfg.vocab_padded, cfg.n_pos, cfg.ln_eps)
        self._h = self._L.lmrl_gpt2_create(ctypes.byref(cc), self.wte.data_ptr(), self.wpe.data_ptr(),
                                           self.lnf_g.data_ptr(), self.lnf_b.data_ptr(), self._ptrs)
        if not self._h:
            raise _lib.LmrlError(self._L.lmrl_last_error().decode())

    def load_params(self, params: Dict[str, "torch.Tensor"]) -> None:
        """Overwrite this engine's weights IN PLACE from a parameter dict of the same architecture (HF names; fp32 masters of a trainer,
        device or host) — the online loops' `policy.set_params(train_state.params)` after every round (llm_rl_scripts/wordle/ppo/
        train_ppo_gpt2.py via LLM_RL/algorithms/ppo/train.py) without a host round trip or a new engine: device-to-device copies with the
        bf16 rounding / [out][in] layout of `__init__`, then the LayerNorm-folded copies are re-derived (`lmrl_gpt2_refresh`).  Sessions, KV
        caches and captured hipGraphs of this engine stay valid (same addresses)."""
        cfg = self.cfg
        sd = {k[len("transformer."):] if k.startswith("transformer.") else k: v for k, v in params.items()}
        self.wte[: cfg.vocab].copy_(sd["wte.weight"][: cfg.vocab])
        self.wpe.copy_(sd["wpe.weight"])
        self.lnf_g.copy_(sd["ln_f.weight"]); self.lnf_b.copy_(sd["ln_f.bias"])
        names = ("ln_1.weight", "ln_1.bias", "attn.c_attn.weight", "attn.c_attn.bias", "attn.c_proj.weight", "attn.c_proj.bias",
                 "ln_2.weight", "ln_2.bias", "mlp.c_fc.weight", "mlp.c_fc.bias", "mlp.c_proj.weight", "mlp.c_proj.bias")
        for l, tensors in enumerate(self.layers):
            for name, dst in zip(names, tensors):
                src = sd[f"h.{l}.{name}"]
                dst.copy_(src.t() if (name.endswith(".weight") and src.dim() == 2) else src)      # Conv1D kernels [in][out] -> [out][in]
        _lib.check(self._L.lmrl_gpt2_refresh(self._h, _lib.stream_ptr()), "lmrl_gpt2_refresh")

    @classmethod
    def random_init(cls, cfg: GPT2Config, seed: int = 0, device=None) -> "GPT2Engine":
        return cls(cfg, init_hf_style_state_dict(cfg, seed), device)

    def n_params(self) -> int:
        c = self.cfg
        return c.vocab * c.d_model + c.n_pos * c.d_model + 2 * c.d_model + c.n_layer * (
            4 * c.d_model + 3 * c.d_model * c.d_model + 3 * c.d_model + c.d_model * c.d_model + c.d_model +
            2 * c.d_model * c.d_ff + c.d_ff + c.d_model)

    def session(self, batch: int, tmax: int, flags: int = 0) -> "KVSession":
        return KVSession(self, batch, tmax, flags)

    def __del__(self):
        try:
            if self._h:
                self._L.lmrl_gpt2_destroy(self._h)
                self._h = None
        except Exception:
            pass


# lmrl_gpt2_forward flags (include/lmrl_amd.h): per-call variants, held per SESSION — never process state
FWD_LN_STANDALONE, FWD_RAGGED_ALWAYS, FWD_RAGGED_NEVER, FWD_ATTN_VALU, FWD_KV_FROM_GEMM, FWD_FULL_LAST_LAYER = 1, 2, 4, 8, 16, 32
FWD_ATTN_ITEMS2, FWD_ATTN_ITEMS3 = 64, 128
FWD_SKINNY = 1 << 30       # decode forwards of <= 16 sequences on the skinny-M Dense kernels (csrc/skinny_gemm.h)


class KVSession:
    """Persistent per-env KV cache + workspace for `batch` lock-step sequences of at most `tmax` tokens.
    `flags` (FWD_*) are this session's forward variants; two sessions with different flags can be interleaved freely."""

    def __init__(self, eng: GPT2Engine, batch: int, tmax: int, flags: int = 0):
        import torch
        self.eng, self.B, self.tmax, self.flags = eng, batch, tmax, int(flags)
        L, dev = eng._L, eng.device
        self.kv = torch.zeros(L.lmrl_gpt2_kv_bytes(eng._h, batch, tmax), dtype=torch.uint8, device=dev)
        self.ws = {c: torch.zeros(L.lmrl_gpt2_ws_bytes(eng._h, batch, c), dtype=torch.uint8, device=dev) for c in (1, 8, 16)}
        self.len = torch.zeros(batch, dtype=torch.int32, device=dev)
        self.last_hidden = torch.zeros(batch, eng.cfg.d_model, dtype=torch.bfloat16, device=dev)
        self.sample_ws = torch.zeros(L.lmrl_sample_ws_bytes(batch, eng.cfg.vocab_padded), dtype=torch.uint8, device=dev)
        self.token = torch.zeros(batch, dtype=torch.int32, device=dev)
        self.logprob = torch.zeros(batch, dtype=torch.float32, device=dev)

    def reset(self):
        self.len.zero_()
        self._len_bound = 0
        self.shared_prefix = 0
        self._prefix = None

    def set_len(self, lens):
        """Truncate / set every env's cache length from the host (int array [B]): positions >= lens[b] are treated as free and will be
        overwritten by the next forwards — how a policy keeps the K/V rows of the longest common prefix of consecutive prompts."""
        import torch
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        assert lens.shape == (self.B,) and int(lens.max(initial=0)) <= self.tmax
        self.len.copy_(torch.from_numpy(lens))
        self._len_bound = int(lens.max(initial=0))
        self.shared_prefix = 0
        self._prefix = None

    shared_prefix = 0   # positions [0, shared_prefix) of every env's cache equal env 0's (set by broadcast_prefix_from, cleared by reset)

    _len_bound = 0   # host-side upper bound of max(self.len): forwards are enqueued without reading the device lengths back

    _prefix = None   # (_CKVPrefix, keep-alive refs) while the envs stand on rows of a prompt-prefix session (attach_prefix_from)

    def forward(self, tokens, cnt, chunk: int, all_hidden=None, len_bound_after: Optional[int] = None):
        """tokens int32 [B*chunk], cnt int32 [B]; updates the cache, self.len and self.last_hidden.  `len_bound_after`: the caller's exact
        knowledge of max(len) after this forward (ragged prefills); default: the previous bound + chunk."""
        e = self.eng
        self._len_bound = self._len_bound + chunk if len_bound_after is None else int(len_bound_after)
        if self._len_bound > self.tmax:
            raise _lib.LmrlError(f"KV cache overflow: up to {self._len_bound} positions would be written into a cache of tmax = {self.tmax} "
                                 "(size the session for prompt + generated tokens, or reset() it)")
        if self._prefix is not None:
            if chunk != 1 or all_hidden is not None:
                raise _lib.LmrlError("a session attached to a prompt-prefix cache takes single-token decode forwards only (reset() it first)")
            _lib.check(e._L.lmrl_gpt2_forward_prefixed(e._h, _lib.ptr(self.kv), self.tmax, _lib.ptr(self.ws[1]), _lib.ptr(tokens), _lib.ptr(cnt),
                                                       _lib.ptr(self.len), self.B, _lib.ptr(self.last_hidden), ctypes.byref(self._prefix[0]),
                                                       self.flags, _lib.stream_ptr()), "lmrl_gpt2_forward_prefixed")
            return self.last_hidden
        _lib.check(e._L.lmrl_gpt2_forward(e._h, _lib.ptr(self.kv), self.tmax, _lib.ptr(self.ws[chunk]), _lib.ptr(tokens),
                                          _lib.ptr(cnt), _lib.ptr(self.len), self.B, chunk, _lib.ptr(self.last_hidden),
                                          _lib.ptr(all_hidden), self.flags | ((self.shared_prefix & 0xFF) << 8), _lib.stream_ptr()), "lmrl_gpt2_forward")
        return self.last_hidden

    def broadcast_prefix_from(self, src: "KVSession", n_pos: int):
        """Every env of this session starts with the `n_pos`-token prefix held by the 1-env session `src`
        (`lmrl_gpt2_kv_broadcast`): K/V rows, cache lengths and the last hidden state."""
        assert src.B == 1 and src.eng is self.eng
        e = self.eng
        _lib.check(e._L.lmrl_gpt2_kv_broadcast(e._h, _lib.ptr(src.kv), src.tmax, _lib.ptr(self.kv), self.tmax, self.B, n_pos,
                                               _lib.ptr(src.last_hidden), _lib.ptr(self.last_hidden), _lib.ptr(self.len), _lib.stream_ptr()),
                   "lmrl_gpt2_kv_broadcast")
        self._len_bound = max(self._len_bound, n_pos)
        self.shared_prefix = min(n_pos, 255)

    def gather_prefix_from(self, src: "KVSession", idx, max_pos: int):
        """Env b starts from the prompt held by row idx[b] (int32 device tensor; < 0: empty cache) of the session `src` of the same
        engine (`lmrl_gpt2_kv_gather`): K/V rows, cache length and last hidden state.  `max_pos` bounds the copied prompt lengths."""
        assert src.eng is self.eng
        e = self.eng
        if max_pos > self.tmax:
            raise _lib.LmrlError(f"KV cache overflow: prompts of up to {max_pos} positions into a cache of tmax = {self.tmax}")
        _lib.check(e._L.lmrl_gpt2_kv_gather(e._h, _lib.ptr(src.kv), src.B, src.tmax, _lib.ptr(src.len), _lib.ptr(src.last_hidden), _lib.ptr(idx),
                                            _lib.ptr(self.kv), self.tmax, self.B, _lib.ptr(self.last_hidden), _lib.ptr(self.len), _lib.stream_ptr()),
                   "lmrl_gpt2_kv_gather")
        self._len_bound = max_pos
        self.shared_prefix = 0
        self._prefix = None

    def attach_prefix_from(self, src: "KVSession", idx, max_pos: int, group: bool = True):
        """The copy-free form of `gather_prefix_from` (`lmrl_gpt2_kv_attach` + `lmrl_gpt2_forward_prefixed`): env b's positions below the
        prompt length of row idx[b] are READ from `src`'s cache by the decode attention; only the generated tokens' rows are written to
        this session.  `idx` must stay alive and unchanged until the next attach / reset (it is the kernel's row table); `group`: launch
        the envs grouped by prefix row (rows shared by several envs are then served from L2).  Decode forwards only."""
        import torch
        assert src.eng is self.eng
        e = self.eng
        if max_pos > self.tmax:
            raise _lib.LmrlError(f"KV cache overflow: prompts of up to {max_pos} positions into a cache of tmax = {self.tmax}")
        if getattr(self, "_pfx_n", None) is None:
            self._pfx_n = torch.zeros(self.B, dtype=torch.int32, device=e.device)
            self._pfx_order = torch.zeros(self.B, dtype=torch.int32, device=e.device)
        _lib.check(e._L.lmrl_gpt2_kv_attach(e._h, src.B, _lib.ptr(src.len), _lib.ptr(src.last_hidden), _lib.ptr(idx), self.B, self.tmax,
                                            _lib.ptr(self.last_hidden), _lib.ptr(self.len), _lib.ptr(self._pfx_n),
                                            _lib.ptr(self._pfx_order) if group else None, _lib.stream_ptr()), "lmrl_gpt2_kv_attach")
        self._prefix = (_CKVPrefix(_lib.ptr(src.kv), src.B, src.tmax, _lib.ptr(idx), _lib.ptr(self._pfx_n),
                                   _lib.ptr(self._pfx_order) if group else None), src, idx)
        self._len_bound = max_pos
        self.shared_prefix = 0

    def sample(self, params: SampleParams, steer_tok=None, active=None, hidden=None, logits_out=None,
               q1=None, q2=None, want_logprob: bool = True):
        """Fused LM head + sampling of one token per env from `hidden` (default: last_hidden).
        logits_out: with 0 < top_k <= 256 and / or 0 < top_p < 1 the call keeps candidates, not logits (logits_out is then only scratch for rows
        handed back to the materialised selection); set `params.flags = SAMPLE_WANT_LOGITS` to have the logits written there as well.
        q1/q2: optional (q_hidden bf16 [B][d], w bf16 [Vp][d], bias f32 [Vp]) ILQL operands.
        want_logprob=False skips the log-sum-exp over the vocabulary (the reference's sampling step returns only the token;
        rollouts do not use the sampled token's log-probability) and returns (token, None)."""
        e = self.eng
        h = self.last_hidden if hidden is None else hidden
        qa = [None] * 6
        if q1 is not None:
            qa[0:3] = [_lib.ptr(x) for x in q1]
        if q2 is not None:
            qa[3:6] = [_lib.ptr(x) for x in q2]
        _lib.check(e._L.lmrl_lm_head_sample(_lib.ptr(h), _lib.ptr(e.wte), qa[0], qa[1], qa[2], qa[3], qa[4], qa[5], self.B,
                                            e.cfg.d_model, e.cfg.vocab, e.cfg.vocab_padded, ctypes.byref(params),
                                            _lib.ptr(steer_tok), _lib.ptr(active), _lib.ptr(self.token),
                                            _lib.ptr(self.logprob) if want_logprob else None,
                                            _lib.ptr(logits_out), _lib.ptr(self.sample_ws), _lib.stream_ptr()),
                   "lmrl_lm_head_sample")
        return self.token, (self.logprob if want_logprob else None)
