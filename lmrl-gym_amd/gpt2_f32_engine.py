"""fp32 rollout engine — the reference's DEFAULT rollout arithmetic.

`GPT2Engine` (gpt2.py) is the throughput mode: bf16 weights / activations, the reference's OPTIONAL `bf16_activations` setting.  The
reference's default is float32 (llm_rl_scripts/wordle/bc/eval_bc_gpt2.py:34,69 — `bf16_activations: bool=False` -> jnp.float32), and
BASELINE.json asks for sampled actions within fp32 tolerance.  `GPT2EngineF32` is that mode: fp32 weights, fp32 activations, an fp32
K/V cache, every matmul on the f32-input MFMA (`lmrl_sgemm`, bit-for-bit an fmaf chain), LayerNorm / gelu / embeddings from the fp32
train-step kernels, attention in `csrc/attn_cached_f32.hip`, the LM head as a plain fp32 GEMM into materialised logits followed by the
same sampler (`lmrl_sample_logits_steer`: identical random streams, top-k / top-p warpers).  `KVSessionF32` has the interface of
`KVSession` (reset / forward / sample / broadcast_prefix_from / last_hidden / token), so `WordleRolloutEngine`, the text policies and
the tests drive either engine unchanged; `tests/test_gpu_f32_engine.py` compares EVERY sampled token of a rollout with the float64
oracle.  It trades throughput for exactness (plain fp32 GEMMs at ~0.5 of the 157 TFLOP/s f32 MFMA peak; `bench.py` reports it as
`fp32_mode`).  There is no ILQL-perturbed sampling here (the Q-head operands of the fused sampler are bf16): policy rollouts only.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import numpy as np

from . import _lib
from .gpt2 import GPT2Config, SampleParams, init_hf_style_state_dict
from .train import ops


class GPT2EngineF32:
    def __init__(self, cfg: GPT2Config, state_dict: Dict[str, "torch.Tensor"], device=None, matmul: str = "f32"):
        """matmul = "f32": every product on the f32-input MFMA (exact fp32, 157 TFLOP/s peak).
        matmul = "bf16x3": every Dense / LM-head product as ONE bf16 GEMM over K' = 3 K on three-term splits of the fp32 operands
        (activations [hi | lo | hi] by `lmrl_split3_bf16`, weights [hi | hi | lo] staged here): hi.hi + lo.hi + hi.lo accumulated in fp32, i.e.
        ~16 mantissa bits per product (more than the TF32 products XLA uses by default for float32 matmuls on GPUs) at 3/16 of the f32 MFMA's
        cost.  LayerNorm, gelu, the residual stream, the K/V cache and the attention stay fp32 in both modes."""
        import torch
        assert matmul in ("f32", "bf16x3")
        self.matmul = matmul
        self.cfg = cfg
        self.device = device or _lib.require_gpu()
        self._L = _lib.lib()
        assert cfg.d_model == cfg.n_head * 64, "head dim 64 (every GPT-2 size)"
        sd = {k[len("transformer."):] if k.startswith("transformer.") else k: v for k, v in state_dict.items()}
        f32 = lambda t: t.to(self.device, torch.float32).contiguous()
        wte = torch.zeros(cfg.vocab_padded, cfg.d_model, dtype=torch.float32, device=self.device)
        wte[: cfg.vocab] = f32(sd["wte.weight"][: cfg.vocab])
        self.wte, self.wpe = wte, f32(sd["wpe.weight"])
        self.lnf_g, self.lnf_b = f32(sd["ln_f.weight"]), f32(sd["ln_f.bias"])
        names = ("ln_1.weight", "ln_1.bias", "attn.c_attn.weight", "attn.c_attn.bias", "attn.c_proj.weight", "attn.c_proj.bias",
                 "ln_2.weight", "ln_2.bias", "mlp.c_fc.weight", "mlp.c_fc.bias", "mlp.c_proj.weight", "mlp.c_proj.bias")
        self.layers = [{n: f32(sd[f"h.{l}.{n}"]) for n in names} for l in range(cfg.n_layer)]     # Conv1D kernels stay [in][out]
        if matmul == "bf16x3":
            def x3(w_out_in):        # fp32 [out][in] -> bf16 [out][3 in] = [hi | hi | lo]
                hi = w_out_in.to(torch.bfloat16)
                lo = (w_out_in - hi.float()).to(torch.bfloat16)
                return torch.cat([hi, hi, lo], dim=1).contiguous()
            for p in self.layers:
                for n in ("attn.c_attn.weight", "attn.c_proj.weight", "mlp.c_fc.weight", "mlp.c_proj.weight"):
                    p[n + ".x3"] = x3(p[n].t().contiguous())
            self.wte_x3 = x3(self.wte)

    def splitk_ws_bytes(self, rows: int) -> int:
        """Bytes of split-K partial-sum workspace ONE caller needs for this engine's products on `rows` rows (0: no product takes the split-K
        path).  The workspace belongs to the CALLER (a KV session sizes it once, before any graph capture): sessions on different HIP streams
        (`text_env_eval(concurrent=n)` lanes) share this engine and must not share partial sums, and a captured hipGraph keeps the pointer."""
        if self.matmul != "bf16x3":
            return 0
        c, need = self.cfg, 0
        for k, n in ((c.d_model, 3 * c.d_model), (c.d_model, c.d_model), (c.d_model, c.d_ff), (c.d_ff, c.d_model)):
            if n % 64 == 0 and 3 * k >= 4096:
                need = max(need, int(self._L.lmrl_gemm_bf16_splitk_ws_bytes(rows, n, 3 * k)))
        return need

    def linear(self, x, rows, k, n, w_name, p, y, scratch, gelu_split=None, splitk_ws=None):
        """y[rows][n] = x[rows][k] @ W + b for a layer's Dense `w_name` in this engine's matmul mode (`scratch`: bf16 [rows][3 k] split buffer).
        bf16x3 with `gelu_split` (a bf16 buffer of >= rows * 3 n elements): gelu_new(y) is written there as a split operand instead of y.
        `splitk_ws`: the caller's uint8 workspace of >= `splitk_ws_bytes(rows)` bytes (see there); without one the long-K products run unsplit."""
        if self.matmul == "f32":
            ops.sgemm(x, p[w_name + ".weight"], y, rows, n, k, lda=k, ldb=n, ldc=n, bias=p[w_name + ".bias"])
            return
        L = self._L
        if x is not None:          # None: `scratch` already holds the split operand (written by the producing LayerNorm / gelu launch)
            _lib.check(L.lmrl_split3_bf16(x.data_ptr(), k, rows, k, scratch.data_ptr(), 3 * k, _lib.stream_ptr()), "lmrl_split3_bf16")
        if gelu_split is not None:
            # c_fc: the epilogue applies gelu_new and writes the c_proj product's split operand [rows][3 n] itself (EPI_GELU_SPLIT3 = 13): no fp32
            # pre-activation tensor, no gelu + split pass over it
            _lib.check(L.lmrl_gemm_bf16(scratch.data_ptr(), p[w_name + ".weight.x3"].data_ptr(), p[w_name + ".bias"].data_ptr(), gelu_split.data_ptr(), rows, n,
                                        3 * k, 3 * k, 3 * n, n, 13, _lib.stream_ptr()), "lmrl_gemm_bf16 (bf16x3, gelu + split epilogue)")
            return
        if n % 64 == 0 and 3 * k >= 4096:
            # few output tiles and a long K' (the MLP's c_proj, K' = 9216: 48 tiles of 128 x 128 at decode size, 128 tiles of 256 x 192 at chunk size):
            # deterministic split-K
            nb = L.lmrl_gemm_bf16_splitk_ws_bytes(rows, n, 3 * k)
            if nb and splitk_ws is not None:
                if splitk_ws.numel() < nb:
                    raise _lib.LmrlError(f"split-K workspace of {splitk_ws.numel()} B < {nb} B needed for {rows} x {n} x {3 * k}")
                _lib.check(L.lmrl_gemm_bf16_splitk_bias(scratch.data_ptr(), p[w_name + ".weight.x3"].data_ptr(), p[w_name + ".bias"].data_ptr(), y.data_ptr(), rows, n,
                                                        3 * k, 3 * k, 3 * k, n, splitk_ws.data_ptr(), _lib.stream_ptr()), "lmrl_gemm_bf16_splitk_bias (bf16x3)")
                return
        _lib.check(L.lmrl_gemm_bf16(scratch.data_ptr(), p[w_name + ".weight.x3"].data_ptr(), p[w_name + ".bias"].data_ptr(), y.data_ptr(), rows, n, 3 * k, 3 * k,
                                    n, n, 3, _lib.stream_ptr()), "lmrl_gemm_bf16 (bf16x3)")

    @classmethod
    def random_init(cls, cfg: GPT2Config, seed: int = 0, device=None, matmul: str = "f32") -> "GPT2EngineF32":
        return cls(cfg, init_hf_style_state_dict(cfg, seed), device, matmul=matmul)

    def session(self, batch: int, tmax: int, flags: int = 0) -> "KVSessionF32":
        return KVSessionF32(self, batch, tmax)


class KVSessionF32:
    """fp32 twin of `gpt2.KVSession`: persistent fp32 K/V cache [layer][K|V][B][tmax][d] + workspaces for chunk forwards."""

    def __init__(self, eng: GPT2EngineF32, batch: int, tmax: int):
        import torch
        t = torch
        self.eng, self.B, self.tmax, self.flags = eng, batch, tmax, 0
        c, dev = eng.cfg, eng.device
        self.kv = t.zeros(c.n_layer, 2, batch, tmax, c.d_model, dtype=t.float32, device=dev)
        self.len = t.zeros(batch, dtype=t.int32, device=dev)
        self.last_hidden = t.zeros(batch, c.d_model, dtype=t.float32, device=dev)      # ln_f of each env's last token
        self._last_x = t.zeros(batch, c.d_model, dtype=t.float32, device=dev)
        self.token = t.zeros(batch, dtype=t.int32, device=dev)
        self.logprob = t.zeros(batch, dtype=t.float32, device=dev)
        self.logits = t.zeros(batch, c.vocab_padded, dtype=t.float32, device=dev)
        self._ws = {}
        self._len_bound = 0

    def _bufs(self, C):
        import torch
        t = torch
        if C not in self._ws:
            c, R, dev = self.eng.cfg, self.B * C, self.eng.device
            f = lambda *s: t.empty(*s, dtype=t.float32, device=dev)
            self._ws[C] = dict(ids=t.zeros(R, dtype=t.int32, device=dev), pos=t.zeros(R, dtype=t.int32, device=dev), x=f(R, c.d_model), h=f(R, c.d_model),
                               r=f(R, c.d_model),
                               qkv=f(R, 3 * c.d_model), att=f(R, c.d_model), ff=f(R, c.d_ff), mean=f(R), rstd=f(R),
                               split=t.zeros(R * 3 * c.d_model, dtype=t.bfloat16, device=dev) if self.eng.matmul == "bf16x3" else None,
                               split2=t.zeros(R * 3 * c.d_ff, dtype=t.bfloat16, device=dev) if self.eng.matmul == "bf16x3" else None,
                               # split-K partial sums of THIS session's long-K products: per session and per chunk size, sized once here (never
                               # reallocated: captured graphs hold the pointer; never shared: lanes on other streams own theirs)
                               splitk=t.empty(self.eng.splitk_ws_bytes(R), dtype=t.uint8, device=dev) if self.eng.splitk_ws_bytes(R) else None)
        return self._ws[C]

    def reset(self):
        self.len.zero_()
        self._len_bound = 0

    def forward(self, tokens, cnt, chunk: int, all_hidden=None, len_bound_after: Optional[int] = None):
        """tokens int32 [B*chunk], cnt int32 [B] (env b's new tokens are slots b*chunk .. b*chunk + cnt[b] - 1): appends them to the cache,
        advances self.len, leaves ln_f(hidden of each env's last new token) in self.last_hidden."""
        e, c, L, sp = self.eng, self.eng.cfg, self.eng._L, _lib.stream_ptr()
        C, B, d, R = chunk, self.B, c.d_model, self.B * chunk
        self._len_bound = self._len_bound + C if len_bound_after is None else int(len_bound_after)
        if self._len_bound > self.tmax:
            raise _lib.LmrlError(f"KV cache overflow: up to {self._len_bound} positions into a cache of tmax = {self.tmax}")
        w = self._bufs(C)
        w["ids"].copy_(tokens.view(-1)[:R])
        _lib.check(L.lmrl_chunk_begin_f32(_lib.ptr(self.len), _lib.ptr(cnt), _lib.ptr(w["ids"]), _lib.ptr(w["pos"]), B, C, c.n_pos, sp), "lmrl_chunk_begin_f32")
        x, h, qkv, att, ff = w["x"], w["h"], w["qkv"], w["att"], w["ff"]
        ops.embed_fwd(e.wte, e.wpe, w["ids"], w["pos"], x, R, d, vocab=c.vocab_padded)     # rows >= cfg.vocab of the engine table are zero; dead slots carry id 0
        r, pending = w["r"], None              # residual adds ride in the LayerNorm launch behind them (lmrl_layernorm_add_fwd: x += pending, then LN)
        x3 = e.matmul == "bf16x3" and d % 256 == 0 and d <= 1280      # bf16x3: LayerNorm / gelu write the three-term split operand themselves

        def ln(resid, g, b):
            if x3:
                _lib.check(L.lmrl_layernorm_add_fwd_split3(x.data_ptr(), _lib.ptr(resid), g.data_ptr(), b.data_ptr(), w["mean"].data_ptr(), w["rstd"].data_ptr(),
                                                           w["split"].data_ptr(), R, d, float(c.ln_eps), sp), "lmrl_layernorm_add_fwd_split3")
                return None
            ops.layernorm_add_fwd(x, resid, g, b, h, w["mean"], w["rstd"], None, 0, R, d, c.ln_eps)
            return h
        for l, p in enumerate(e.layers):
            e.linear(ln(pending, p["ln_1.weight"], p["ln_1.bias"]), R, d, 3 * d, "attn.c_attn", p, qkv, w["split"])
            # (bf16x3: the attention lanes that hold the output also write it as the projection GEMM's split operand — no split pass)
            _lib.check(L.lmrl_attn_cached_f32_split3(_lib.ptr(qkv), _lib.ptr(self.kv[l, 0]), _lib.ptr(self.kv[l, 1]), _lib.ptr(self.len), _lib.ptr(cnt),
                                                     _lib.ptr(att), _lib.ptr(w["split"]) if x3 else None, B, C, c.n_head, self.tmax, sp), "lmrl_attn_cached_f32")
            e.linear(None if x3 else att, R, d, d, "attn.c_proj", p, r, w["split"])   # r = att . Wproj + b ; x += r inside the ln_2 launch
            if x3:
                # c_fc reads the LayerNorm's split operand (w["split"], pitch 3 d) and writes gelu's split operand (w["split2"], pitch 3 d_ff)
                e.linear(ln(r, p["ln_2.weight"], p["ln_2.bias"]), R, d, c.d_ff, "mlp.c_fc", p, None, w["split"], gelu_split=w["split2"])
                e.linear(None, R, c.d_ff, d, "mlp.c_proj", p, r, w["split2"], splitk_ws=w["splitk"])
            else:
                e.linear(ln(r, p["ln_2.weight"], p["ln_2.bias"]), R, d, c.d_ff, "mlp.c_fc", p, ff, w["split"])
                ops.gelu_fwd(ff, ff)
                e.linear(ff, R, c.d_ff, d, "mlp.c_proj", p, r, w["split2"], splitk_ws=w["splitk"])         # r = mlp output ; added by the next LayerNorm launch
            pending = r
        if all_hidden is not None:
            ops.layernorm_add_fwd(x, pending, e.lnf_g, e.lnf_b, all_hidden, w["mean"], w["rstd"], None, 0, R, d, c.ln_eps)
        elif pending is not None:
            ops.axpby(1.0, pending, 1.0, x, x)
        _lib.check(L.lmrl_chunk_end_f32(_lib.ptr(x), _lib.ptr(cnt), _lib.ptr(self._last_x), _lib.ptr(self.len), B, C, d, sp), "lmrl_chunk_end_f32")
        ops.layernorm_fwd(self._last_x, e.lnf_g, e.lnf_b, self.last_hidden, w["mean"], w["rstd"], B, d, c.ln_eps)
        return self.last_hidden

    def broadcast_prefix_from(self, src: "KVSessionF32", n_pos: int):
        """Every env starts with the n_pos-token prefix held by the 1-env session `src` (device-to-device row copies)."""
        assert src.B == 1 and src.eng is self.eng
        self.kv[:, :, :, :n_pos].copy_(src.kv[:, :, :, :n_pos].expand(-1, -1, self.B, -1, -1))
        self._last_x.copy_(src._last_x.expand(self.B, -1))
        self.last_hidden.copy_(src.last_hidden.expand(self.B, -1))
        self.len.fill_(n_pos)
        self._len_bound = max(self._len_bound, n_pos)

    def lm_logits(self, hidden=None):
        """logits fp32 [B][vocab_padded] = hidden . wte^T (tied LM head) into self.logits."""
        c, e = self.eng.cfg, self.eng
        h = self.last_hidden if hidden is None else hidden
        if e.matmul == "f32":
            ops.sgemm(h, e.wte, self.logits, self.B, c.vocab_padded, c.d_model, trans_b=True, lda=c.d_model, ldb=c.d_model, ldc=c.vocab_padded)
        else:
            import torch
            if getattr(self, "_lm_split", None) is None:
                self._lm_split = torch.empty(self.B * 3 * c.d_model, dtype=torch.bfloat16, device=e.device)
            _lib.check(e._L.lmrl_split3_bf16(h.data_ptr(), c.d_model, self.B, c.d_model, self._lm_split.data_ptr(), 3 * c.d_model, _lib.stream_ptr()), "lmrl_split3_bf16")
            _lib.check(e._L.lmrl_gemm_bf16(self._lm_split.data_ptr(), e.wte_x3.data_ptr(), None, self.logits.data_ptr(), self.B, c.vocab_padded, 3 * c.d_model,
                                           3 * c.d_model, c.vocab_padded, c.vocab_padded, 3, _lib.stream_ptr()), "lmrl_gemm_bf16 (bf16x3 LM head)")
        return self.logits

    def sample(self, params: SampleParams, steer_tok=None, active=None, hidden=None, logits_out=None, q1=None, q2=None, want_logprob: bool = True):
        """One token per env from `hidden` (default: last_hidden).  matmul = "f32": fp32 LM head into materialised logits, then the sampler on them
        (same streams and warpers as the fused bf16 path); matmul = "bf16x3": the fused LM-head sampler on the split operands."""
        if q1 is not None or q2 is not None:
            raise _lib.LmrlError("GPT2EngineF32 samples from the policy logits only (the ILQL Q-head operands of the fused sampler are bf16)")
        c, e = self.eng.cfg, self.eng
        if e.matmul == "bf16x3":
            # the LM head and the sampler in ONE launch (`lm_head_sample_kernel`, the bf16 engine's): the tied-head product runs as a bf16 GEMM over
            # K' = 3 d on the split operands and the Gumbel-max / arg-max epilogue reads the fp32 accumulators — no [B][V] fp32 logits written and
            # re-read per token (2 x 206 MB at B = 1024), same random streams (Philox or LMRL_RNG_JAX: keyed by row / column / step, not by K)
            import torch
            h = self.last_hidden if hidden is None else hidden
            if getattr(self, "_lm_split", None) is None:
                self._lm_split = torch.empty(self.B * 3 * c.d_model, dtype=torch.bfloat16, device=e.device)
                self._sample_ws = torch.zeros(e._L.lmrl_sample_ws_bytes(self.B, c.vocab_padded), dtype=torch.uint8, device=e.device)   # zero-filled once (work-list heads)
            _lib.check(e._L.lmrl_split3_bf16(h.data_ptr(), c.d_model, self.B, c.d_model, self._lm_split.data_ptr(), 3 * c.d_model, _lib.stream_ptr()), "lmrl_split3_bf16")
            _lib.check(e._L.lmrl_lm_head_sample(_lib.ptr(self._lm_split), _lib.ptr(e.wte_x3), None, None, None, None, None, None, self.B, 3 * c.d_model, c.vocab,
                                                c.vocab_padded, ctypes.byref(params), _lib.ptr(steer_tok), _lib.ptr(active), _lib.ptr(self.token),
                                                _lib.ptr(self.logprob) if want_logprob else None, _lib.ptr(logits_out), _lib.ptr(self._sample_ws),
                                                _lib.stream_ptr()), "lmrl_lm_head_sample (bf16x3)")
            return self.token, (self.logprob if want_logprob else None)
        lg = self.lm_logits(hidden)
        _lib.check(self.eng._L.lmrl_sample_logits_steer(_lib.ptr(lg), c.vocab_padded, self.B, c.vocab, ctypes.byref(params), _lib.ptr(steer_tok),
                                                        _lib.ptr(active), _lib.ptr(self.token), _lib.ptr(self.logprob) if want_logprob else None,
                                                        _lib.stream_ptr()), "lmrl_sample_logits_steer")
        if logits_out is not None:
            logits_out.copy_(lg)           # (after the steer offset, as the fused path reports its combined logits)
        return self.token, (self.logprob if want_logprob else None)
