"""GPT-2 forward + backward in float32 on the HIP train-step kernels.

Mirrors what `policy_model(input_ids, attention_mask, position_ids, params, output_hidden_states=True)` +
`jax.value_and_grad` do in the reference `_step` functions (LLM_RL/algorithms/ppo/gpt2/interface.py:111-133,
LLM_RL/algorithms/ilql/gpt2/interface.py:139-177): HF GPT-2 blocks (pre-LN, gelu_new, causal attention masked by
attention_mask, learned positions, tied LM head), float32 parameters and activations.

Parameters are a dict of fp32 tensors with HF names/layouts (`h.0.attn.c_attn.weight` is [in, out]); gradients
accumulate into a dict with the same keys.  torch only allocates; the arithmetic is sgemm_f32 / train_ops kernels.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

from . import ops


class GradArena(dict):
    """Gradient dict whose tensors are views into ONE flat fp32 buffer (`.flat`), laid out in the order the backward pass finalises them
    (`.order`: name -> (offset, numel)).  The data-parallel all-reduce then runs IN PLACE on slices of `.flat` — no torch.cat / copy-back —
    and a slice can be handed to RCCL as soon as the backward pass is done with it (lmrl_gym_amd.dist.GradReducer)."""

    def __init__(self, params: Dict[str, "torch.Tensor"], order=None):
        import torch
        super().__init__()
        names = list(order) if order is not None else list(params)
        assert sorted(names) == sorted(params), "arena order must cover exactly the parameter names"
        total = sum(params[k].numel() for k in names)
        dev = next(iter(params.values())).device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.order = {}
        off = 0
        for k in names:
            n = params[k].numel()
            self[k] = self.flat[off:off + n].view(params[k].shape)
            self.order[k] = (off, n)
            off += n

    @classmethod
    def of_values(cls, params: Dict[str, "torch.Tensor"], order=None) -> "GradArena":
        """An arena holding COPIES of `params` (fp32): the model's parameter store — AdamW and the Polyak target update then run as one
        launch over `.flat` instead of one per tensor."""
        a = cls(params, order)
        for k, v in params.items():
            a[k].copy_(v)
        return a

    def same_layout(self, other) -> bool:
        return isinstance(other, GradArena) and self.order == other.order

    def zero_(self):
        self.flat.zero_()
        return self

    def span(self, names):
        """(lo, hi) element range covering `names` (contiguous by construction when they are adjacent in the order)."""
        lo = min(self.order[k][0] for k in names)
        hi = max(self.order[k][0] + self.order[k][1] for k in names)
        return lo, hi


class Workspace:
    """Caches scratch tensors by (name, shape)."""

    def __init__(self, device):
        import torch
        self.t, self.dev, self.buf = torch, device, {}

    def get(self, name, shape, dtype=None, zero=False):
        dtype = dtype or self.t.float32
        key = (name, tuple(shape), dtype)
        if key not in self.buf:
            self.buf[key] = self.t.empty(shape, dtype=dtype, device=self.dev)
        x = self.buf[key]
        if zero:
            x.zero_()
        return x


class GPT2F32:
    """GPT-2 forward / backward for the train step on fp32 parameters.  `matmul="bf16"` runs every Dense / Conv1D product (and the tied LM
    head) on the bf16 MFMA with fp32 accumulation — the reference's optional `bf16_activations` mode (train_ilql_gpt2.py:193) — see
    `ops.MatmulBF16`; the default "f32" is the reference's default arithmetic."""

    def __init__(self, params: Dict[str, "torch.Tensor"], n_head: int, ln_eps: float = 1e-5, device=None, matmul: str = "f32",
                 attention: str = "flash", gradient_checkpointing: bool = False):
        """attention="flash": online-softmax tile sweeps (csrc/flash_attn_train.hip; operands fp32 or bf16 following `matmul`), no
        [B*H, T, T] tensors; "materialized": the batched-sgemm + softmax formulation (fp32; the cross-check of the flash kernels)."""
        import torch
        assert matmul in ("f32", "bf16") and attention in ("flash", "materialized")
        self.attention = attention
        self.gradient_checkpointing = gradient_checkpointing      # the scripts' flag (train_ilql_gpt2.py:201-202): recompute blocks in backward
        self.t = torch
        self.dev = device or next(iter(params.values())).device
        p32 = {k: v.to(self.dev, torch.float32) for k, v in params.items()}
        self.n_layer = 1 + max(int(k.split(".")[1]) for k in p32 if k.startswith("h."))
        self.p = GradArena.of_values(p32, self.grad_order())     # one flat buffer, in the order the backward pass finalises the gradients
        self.n_head = n_head
        self.eps = ln_eps
        self.d = self.p["wte.weight"].shape[1]
        self.vocab = self.p["wte.weight"].shape[0]
        self.d_ff = self.p["h.0.mlp.c_fc.weight"].shape[1]
        self.ws = Workspace(self.dev)
        self._colsum_ws = torch.empty(64 * max(self.d_ff, 3 * self.d, self.vocab), dtype=torch.float32, device=self.dev)
        self.mm = ops.MatmulBF16(self.dev) if matmul == "bf16" else None
        self.ld_vocab = ops._pad(self.vocab) if self.mm is not None else ops._pad(self.vocab, 4)     # row stride of [rows, V] logits (fp32: 16-byte aligned rows)

    def _layer_forward(self, l: int, x, B: int, T: int, km, flash: bool, lse_n: int, resid_in=None, inference: bool = False):
        """One transformer block: x [B*T, d] -> (x_out, pending, cache of every intermediate the backward pass reads).  resid_in: a residual
        still to be added to x (the previous block's `x_mid`, when its second residual add was left to this block's ln_1: x += resid_in in place,
        inside the LayerNorm launch).  pending: likewise this block's x_mid if x_out is returned WITHOUT it (ops.FUSE_ADD_LN), else None.
        inference (bf16-matmul mode with the fused flash / gelu paths): nobody differentiates this forward (the ILQL target network) — what only a
        backward pass would read is not written: the fp32 c_fc pre-activation ([rows][d_ff] floats per block), the fp32 attention output, a flash
        workspace per block; same launches otherwise, bit-identical hidden states."""
        t = self.t
        R, d, H, p = B * T, self.d, self.n_head, self.p
        hd = d // H
        new = lambda *shape: t.empty(shape, dtype=t.float32, device=self.dev)
        q = f"h.{l}."
        c = dict(x_in=x)
        mm = self.mm
        stage = mm is not None and d % 64 == 0 and self.d_ff % 64 == 0     # producers write the bf16 GEMM operands themselves (no cast pass)
        # stage: LayerNorm / gelu / attention write the bf16 operand of the consuming GEMM themselves, into per-layer buffers the backward
        # reads as they are for the dW products (ops.FUSE_KMAJOR_DW; transposed where that kernel's shape conditions do not hold) — no fp32 copies of
        # h1 / h2 / g exist in this mode (att keeps one: the flash backward reads it)
        c["m1"], c["r1"] = new(R), new(R)
        h1 = h1b = None
        fuse_add = ops.FUSE_ADD_LN
        if stage:
            h1b, ldb = mm.stash(R, d)
            ops.layernorm_add_fwd(x, resid_in, p[q + "ln_1.weight"], p[q + "ln_1.bias"], None, c["m1"], c["r1"], h1b, ldb, R, d, self.eps)
        else:
            h1 = new(R, d)
            ops.layernorm_add_fwd(x, resid_in, p[q + "ln_1.weight"], p[q + "ln_1.bias"], h1, c["m1"], c["r1"], None, 0, R, d, self.eps)
        qkv_fused = stage and flash and ops.fused_ok(3 * d, ops.FUSE_QKV)      # c_attn writes the flash kernels' staged q / k / v itself: no fp32 qkv
        fws = None
        lean = inference and stage and flash and qkv_fused and ops.fused_ok(self.d_ff, ops.FUSE_GELU)
        if flash:
            # a workspace per block and per forward call: the q / k / v matrices staged here are the ones the block's backward sweeps
            fws = c["flash_ws"] = self._flash_ws[0] if lean else t.empty_like(self._flash_ws[0])
        if qkv_fused:
            qkv = None
            ops.linear_fwd_qkv_heads(mm, h1b, p[q + "attn.c_attn.weight"], p[q + "attn.c_attn.bias"], fws, R, d, B, H, T)
        else:
            qkv = new(R, 3 * d)
            ops.linear_fwd(h1, p[q + "attn.c_attn.weight"], p[q + "attn.c_attn.bias"], qkv, R, d, 3 * d, mm=self.mm, xb=h1b)
        att = None if (lean or (stage and flash and ops.D_FROM_BF16_O)) else new(R, d)       # bf16 mode: the backward takes D from the bf16 output
        attb = None
        if flash:
            P = None
            c["lse"] = new(lse_n)
            if stage:
                attb, ldb = mm.stash(R, d)
                ops.flash_attn_fwd_staged(qkv, km, att, c["lse"], fws, attb, ldb, B, H, T, True)
            else:
                ops.flash_attn_fwd(qkv, km, att, c["lse"], fws, B, H, T, self.mm is not None)
        else:
            P = new(B * H, T, T)
            # S = scale * Q K^T per (b, h); Q/K/V are column slices of qkv (row stride 3d)
            ops.sgemm(qkv, qkv, P, T, T, hd, trans_b=True, alpha=1.0 / math.sqrt(hd), lda=3 * d, ldb=3 * d, ldc=T, b_off=d,
                      batch=(B, H), sa=(T * 3 * d, hd), sb=(T * 3 * d, hd), sc=(H * T * T, T * T))
            ops.softmax_causal_fwd(P, km, P, B, H, T)
            ops.sgemm(P, qkv, att, T, hd, T, lda=T, ldb=3 * d, ldc=d, b_off=2 * d, batch=(B, H), sa=(H * T * T, T * T),
                      sb=(T * 3 * d, hd), sc=(T * d, hd))
        x_mid = new(R, d)
        # x_mid = x + proj(att): the add inside the GEMM epilogue / as one axpby (resid=x), or left to the ln_2 launch below (fuse_add)
        ops.linear_fwd(att, p[q + "attn.c_proj.weight"], p[q + "attn.c_proj.bias"], x_mid, R, d, d, mm=self.mm, xb=attb, resid=None if fuse_add else x)
        c["m2"], c["r2"] = new(R), new(R)
        h2 = h2b = None
        if stage:
            h2b, ldb = mm.stash(R, d)
            ops.layernorm_add_fwd(x_mid, x if fuse_add else None, p[q + "ln_2.weight"], p[q + "ln_2.bias"], None, c["m2"], c["r2"], h2b, ldb, R, d, self.eps)
        else:
            h2 = new(R, d)
            ops.layernorm_add_fwd(x_mid, x if fuse_add else None, p[q + "ln_2.weight"], p[q + "ln_2.bias"], h2, c["m2"], c["r2"], None, 0, R, d, self.eps)
        # the pre-activation's only reader is the gelu backward; with both fused epilogues on it is kept rounded to bf16 (ops.PRE_BF16)
        f_bf16 = stage and ops.PRE_BF16 and ops.fused_ok(self.d_ff, ops.FUSE_GELU) and ops.fused_ok(self.d_ff, ops.FUSE_GELU_BWD)
        f = None if lean else (t.empty(R, ops._pitch(self.d_ff), dtype=t.bfloat16, device=self.dev) if f_bf16 else new(R, self.d_ff))
        g = gb = None
        if stage and ops.fused_ok(self.d_ff, ops.FUSE_GELU):          # c_fc writes the pre-activation and the bf16 gelu output in one launch
            gb, ldb = mm.stash(R, self.d_ff)
            ops.linear_fwd_gelu(mm, h2b, p[q + "mlp.c_fc.weight"], p[q + "mlp.c_fc.bias"], f, gb, ldb, R, d, self.d_ff)
        elif stage:
            ops.linear_fwd(h2, p[q + "mlp.c_fc.weight"], p[q + "mlp.c_fc.bias"], f, R, d, self.d_ff, mm=self.mm, xb=h2b)
            gb, ldb = mm.stash(R, self.d_ff)
            ops.gelu_fwd_staged(f, None, gb, ldb, R, self.d_ff)
        else:
            ops.linear_fwd(h2, p[q + "mlp.c_fc.weight"], p[q + "mlp.c_fc.bias"], f, R, d, self.d_ff, mm=self.mm, xb=h2b)
            g = new(R, self.d_ff)
            ops.gelu_fwd(f, g)
        x_out = new(R, d)
        # x_out = x_mid + mlp: as above, or left to the NEXT LayerNorm (the following block's ln_1 / ln_f) through `pending`
        ops.linear_fwd(g, p[q + "mlp.c_proj.weight"], p[q + "mlp.c_proj.bias"], x_out, R, self.d_ff, d, mm=self.mm, xb=gb, resid=None if fuse_add else x_mid)
        c.update(h1b=h1b, attb=attb, h2b=h2b, gb=gb)
        c.update(h1=h1, qkv=qkv, P=P, att=att, x_mid=x_mid, h2=h2, f=f, g=g)
        return x_out, (x_mid if fuse_add else None), c

    # ------------------------------------------------------------------ forward
    def forward(self, input_ids, attention_mask, position_ids, tag: str = "fwd", inference: bool = False, restage: bool = True):
        """input_ids / position_ids int32 [B,T], attention_mask uint8 [B,T] -> (final hidden [B*T, d], cache).  inference: the cache will not be
        differentiated (`backward` must not be called on it): see `_layer_forward`.  restage=False (bf16-matmul mode): the caller knows that the fp32
        masters have not moved since this model's previous forward (the chunk forwards of one PPO data build) — the bf16 operand copies are kept."""
        t = self.t
        if self.mm is not None and restage:
            self.mm.begin_step()          # the optimizer may have moved the fp32 masters since the last forward: re-stage the bf16 copies
            self._stage_weights(True)     # ... of every block's Dense kernels, one launch (the forward operands; backward() adds the dX operands)
        B, T = input_ids.shape
        R, d, H, p = B * T, self.d, self.n_head, self.p
        hd = d // H
        ids = input_ids.reshape(-1).contiguous()
        pos = position_ids.reshape(-1).contiguous()
        km = attention_mask.to(t.uint8).contiguous()
        new = lambda *shape: t.empty(shape, dtype=t.float32, device=self.dev)
        x = new(R, d)
        ops.embed_fwd(p["wte.weight"], p["wpe.weight"], ids, pos, x, R, d, vocab=self.vocab)
        cache = dict(B=B, T=T, ids=ids, pos=pos, km=km, layers=[], inference=bool(inference))
        flash = self.attention == "flash" and hd == 64
        lse_n = 0
        if flash:
            key = (B, H, T)
            if getattr(self, "_flash_key", None) != key:
                self._flash_ws, self._flash_key = ops.flash_attn_ws(B, H, T, self.mm is not None, self.dev), key
            lse_n = self._flash_ws[1]
        cache["flash"] = flash
        cache["lse_n"] = lse_n
        pending = None
        for l in range(self.n_layer):
            x_out, nxt, c = self._layer_forward(l, x, B, T, km, flash, lse_n, resid_in=pending, inference=inference)
            # gradient_checkpointing (train_ilql_gpt2.py:201-202): keep only the block input (x: complete once the block's ln_1 launch has added the
            # pending residual in place); backward() recomputes the block — the same launches on the same inputs, i.e. bit-identical
            # intermediates — right before it differentiates it
            cache["layers"].append(dict(x_in=x) if (self.gradient_checkpointing or inference) else c)
            x, pending = x_out, nxt
        hid, cache["mf"], cache["rf"] = new(R, d), new(R), new(R)
        ops.layernorm_add_fwd(x, pending, p["ln_f.weight"], p["ln_f.bias"], hid, cache["mf"], cache["rf"], None, 0, R, d, self.eps)
        cache["x_final"] = x
        cache["hidden"] = hid
        return hid, cache

    def _stage_weights(self, transposed: bool):
        mats = [self.p[f"h.{l}.{n}"] for l in range(self.n_layer) for n in ("attn.c_attn.weight", "attn.c_proj.weight", "mlp.c_fc.weight", "mlp.c_proj.weight")]
        if getattr(self.p, "flat", None) is not None and all(m.shape[0] % 64 == 0 and m.shape[1] % 64 == 0 for m in mats):
            self.mm.stage_arena(self.p.flat, mats, transposed)

    def lm_logits(self, hidden, rows: int):
        """logits [rows, ld_vocab] (columns [0, V) valid) = hidden @ wte^T (tied head), fp32 — PPOInference.token_logprobs_from_logits casts
        to f32 too.  The ROW PITCH is `self.ld_vocab` (V padded to a multiple of 4 floats in fp32 mode — 16-byte rows —, to a multiple of 64 in
        bf16-matmul mode): only columns [0, V) are valid; view as `[B, T, ld_vocab][..., :V]`, never as `[B, T, V]`."""
        V, d, ld, mm = self.vocab, self.d, self.ld_vocab, self.mm
        logits = self.t.empty(rows, ld, dtype=self.t.float32, device=self.dev)
        if mm is None:
            ops.sgemm(hidden, self.p["wte.weight"], logits, rows, V, d, trans_b=True, lda=d, ldb=d, ldc=ld)
        else:
            hb = mm.cast("x", hidden, rows, d, d)
            wb = mm.cast(("w", self.p["wte.weight"].data_ptr()), self.p["wte.weight"], V, d, d, keep=True)           # [pad(V)][d]
            mm.gemm(hb, wb, None, logits, rows, ops._padn(V), d, ld, V)
        return logits

    def lm_ce(self, hidden, rows: int, targets):
        """What the CE / token-log-probability terms need of the tied LM head's logits -> (logits, yb, lse, logprob): fp32 logits [rows, ld_vocab]
        + a pass over them (fp32 mode: yb None), or in the bf16-matmul mode bf16 logits `yb` with lse / log-probabilities taken from the GEMM's
        fp32 accumulators (logits None; ops.FUSE_CE).  `ce_bwd_any` turns either into d(logits) for `lm_head_backward`."""
        t = self.t
        if ops.FUSE_CE and self.mm is not None and self.d % 64 == 0 and ops._padn(self.vocab) % 128 == 0:
            yb, lse, _, lp = ops.head_fwd_ce(self.mm, hidden, self.p["wte.weight"], None, rows, self.d, self.vocab, targets, w_is_nk=True)
            return None, yb, lse, lp
        logits = self.lm_logits(hidden, rows)
        lp, lse = t.empty(rows, dtype=t.float32, device=self.dev), t.empty(rows, dtype=t.float32, device=self.dev)
        ops.lse_gather(logits, self.ld_vocab, self.vocab, targets, rows, logprob=lp, lse=lse)
        return logits, None, lse, lp

    def token_logprobs(self, hidden_all, rows_idx, targets, n_rows: int, chunk: int = 8192):
        """log p(targets[i] | row rows_idx[i]) for `n_rows` rows of `hidden_all` [*, d] (final hidden states), nothing differentiated —
        `token_logprobs_from_logits` of PPOInference.forward (ppo/base_interface.py:396-403, gpt2/interface.py:264-302) on the rows that have a
        next token only, in row chunks: bf16-matmul mode takes log-sum-exp and the target logit out of the LM-head GEMM's accumulators and stores
        no logits; fp32 mode keeps one [chunk, V] fp32 logits scratch.  -> float32 [n_rows] (device)."""
        t = self.t
        out = t.empty(max(n_rows, 1), dtype=t.float32, device=self.dev)
        fused = ops.FUSE_CE and self.mm is not None and self.d % 64 == 0 and ops._padn(self.vocab) % 128 == 0
        # (bf16-matmul mode: the tied head's bf16 copy is staged on first use after the `forward` that produced `hidden_all` dropped the stale ones)
        for r0 in range(0, n_rows, chunk):
            n = min(chunk, n_rows - r0)
            hq = ops.gather_rows(hidden_all, rows_idx[r0:r0 + n], n, self.d)
            tg = targets[r0:r0 + n]
            if fused:
                _, _, _, lp = ops.head_fwd_ce(self.mm, hq, self.p["wte.weight"], None, n, self.d, self.vocab, tg, w_is_nk=True, keep_logits=False)
                out[r0:r0 + n].copy_(lp)
            else:
                logits = self.lm_logits(hq, n)
                ops.lse_gather(logits, self.ld_vocab, self.vocab, tg, n, logprob=out[r0:r0 + n])
                del logits
        return out[:n_rows]

    def ce_bwd_any(self, logits, yb, lse, targets, coef_ce, coef_gather, rows: int):
        if yb is not None:
            return None, ops.ce_bwd_inplace(yb, self.vocab, lse, targets, coef_ce, coef_gather, rows)
        return self.ce_bwd(logits, lse, targets, coef_ce, coef_gather, rows)

    # ------------------------------------------------------------------ backward
    def grad_order(self):
        """Parameter names in the order `backward` finalises their gradients: ln_f, blocks last to first, then the embeddings (wte also
        receives the tied-LM-head gradient before the transformer backward and the embedding gradient at its very end)."""
        names = ["ln_f.weight", "ln_f.bias"]
        for l in reversed(range(self.n_layer)):
            q = f"h.{l}."
            names += [q + n for n in ("mlp.c_proj.weight", "mlp.c_proj.bias", "mlp.c_fc.weight", "mlp.c_fc.bias", "ln_2.weight", "ln_2.bias",
                                      "attn.c_proj.weight", "attn.c_proj.bias", "attn.c_attn.weight", "attn.c_attn.bias", "ln_1.weight", "ln_1.bias")]
        names += ["wpe.weight", "wte.weight"]
        return names

    def zero_grads(self) -> "GradArena":
        """A zeroed gradient arena (allocated once per model, re-zeroed per step)."""
        if getattr(self, "_arena", None) is None:
            self._arena = GradArena(self.p, self.grad_order())
        return self._arena.zero_()

    def ce_bwd(self, logits, lse, targets, coef_ce, coef_gather, rows: int):
        """d loss / d logits of the CE (+ gather) terms for `lm_head_backward` -> (dlogits, dlb): fp32 mode overwrites `logits` in place
        (dlb None); bf16-matmul mode writes the bf16 operand of the backward products directly (dlogits None: no fp32 copy exists)."""
        if self.mm is None:
            ops.ce_bwd(logits, self.ld_vocab, self.vocab, lse, targets, coef_ce, coef_gather, rows)
            return logits, None
        return None, ops.ce_bwd_staged(self.mm, logits, self.ld_vocab, self.vocab, lse, targets, coef_ce, coef_gather, rows)

    def lm_head_backward(self, hidden, dlogits, rows: int, d_hidden, grads, accumulate_dh: bool, dlb=None):
        """d_hidden (+)= dlogits @ wte ; grads[wte] += dlogits^T @ hidden   (dlogits [rows, ld_vocab], or its staged bf16 operand dlb)"""
        V, d, ld, mm = self.vocab, self.d, self.ld_vocab, self.mm
        if mm is None:
            ops.sgemm(dlogits, self.p["wte.weight"], d_hidden, rows, d, V, lda=ld, ldb=d, ldc=d, beta=1.0 if accumulate_dh else 0.0)
            ops.sgemm(dlogits, hidden, grads["wte.weight"], V, d, rows, trans_a=True, lda=ld, ldb=d, ldc=d, beta=1.0)
            return
        staged = dlb is not None
        if not staged:
            dlb = mm.cast("dy", dlogits, rows, V, ld)                                                                  # [rows][pad(V)]
        wt = mm.cast(("wT", self.p["wte.weight"].data_ptr()), self.p["wte.weight"], V, d, d, transpose=True, keep=True)  # [d][pad(V)]
        mm.gemm(dlb, wt, None, d_hidden, rows, ops._pad(d), V, d, d, accumulate=accumulate_dh)
        if staged:
            dlt = mm.transpose_staged("dyT", dlb, ops._pitch(V), rows, V)                                              # [pad(V)][pad(rows)]
        else:
            dlt = mm.cast("dyT", dlogits, rows, V, ld, transpose=True)
        ht = mm.cast("xT", hidden, rows, d, d, transpose=True)                                                         # [d][pad(rows)]
        mm.gemm(dlt, ht, None, grads["wte.weight"], V, ops._pad(d), rows, d, d, accumulate=True)

    def backward(self, cache, d_hidden, grads: Dict[str, "torch.Tensor"], on_final=None):
        """d_hidden: gradient w.r.t. the final (post ln_f) hidden states [B*T, d]; accumulates into `grads`.
        `on_final(names)` (optional) is called after the launches that complete the gradients `names` have been enqueued — ln_f, then one
        call per block (last to first), then the embeddings — so a data-parallel reducer can start on them while the rest of the backward runs."""
        t = self.t
        if cache.get("inference"):
            raise ValueError("GPT2F32.backward: this cache comes from forward(inference=True) — the lean forward does not keep what a backward reads")
        B, T = cache["B"], cache["T"]
        R, d, H, p, ws = B * T, self.d, self.n_head, self.p, self._colsum_ws
        hd = d // H
        scale = 1.0 / math.sqrt(hd)
        new = lambda *shape: t.empty(shape, dtype=t.float32, device=self.dev)
        dx = new(R, d)
        mm = self.mm
        if mm is not None:
            self._stage_weights(False)
        dxb = [None]       # bf16-matmul mode: the bf16 copy of dx the LayerNorm backward leaves behind = the dy operand of the next c_proj backward
        if ops.layernorm_bwd_fused_supported(d):
            lws = new(ops.layernorm_bwd_fused_ws_floats(R, d))

            def ln_bwd(dy, x, name, mean, rstd, accumulate_dx):   # dx (+)=, gamma / beta gradients accumulated, one pass
                buf, ldb = mm.stage_dy(R, d) if mm is not None else (None, 0)
                ops.layernorm_bwd_fused(dy, x, p[name + ".weight"], mean, rstd, dx, grads[name + ".weight"], grads[name + ".bias"], R, d,
                                        accumulate_dx, True, lws, buf, ldb)
                dxb[0] = buf
        else:
            tmp = new(R, d)

            def ln_bwd(dy, x, name, mean, rstd, accumulate_dx):
                ops.layernorm_bwd(dy, x, p[name + ".weight"], mean, rstd, dx, tmp, R, d, accumulate_dx)
                ops.colsum(tmp, R, d, d, grads[name + ".weight"], True, ws)
                ops.colsum(dy, R, d, d, grads[name + ".bias"], True, ws)
        ln_bwd(d_hidden, cache["x_final"], "ln_f", cache["mf"], cache["rf"], False)
        if on_final is not None:
            on_final(["ln_f.weight", "ln_f.bias"])
        for l in reversed(range(self.n_layer)):
            q = f"h.{l}."
            c = cache["layers"][l]
            if "h1" not in c:              # checkpointed block: recompute its intermediates from the stored input
                _, _, c = self._layer_forward(l, c["x_in"], B, T, cache["km"], cache["flash"], cache["lse_n"])
            # MLP: x_out = x_mid + gelu(h2 W_fc + b) W_proj + b
            if mm is not None and dxb[0] is not None and c.get("gb") is not None and ops.fused_ok(self.d_ff, ops.FUSE_GELU_BWD):
                # dg = dx @ W_proj^T never exists: the dX product's epilogue multiplies by gelu'(f) and writes df as the bf16 operand of c_fc's backward
                df, dfb = None, ops.linear_bwd_dx_gelu(mm, dxb[0], p[q + "mlp.c_proj.weight"], c["f"], R, self.d_ff, d)
                ops.linear_bwd(None, p[q + "mlp.c_proj.weight"], dx, None, grads[q + "mlp.c_proj.weight"], grads[q + "mlp.c_proj.bias"], R, self.d_ff, d, ws,
                               mm=mm, dyb=dxb[0], xb=c["gb"])
            else:
                dg = new(R, self.d_ff)
                ops.linear_bwd(c["g"], p[q + "mlp.c_proj.weight"], dx, dg, grads[q + "mlp.c_proj.weight"], grads[q + "mlp.c_proj.bias"], R, self.d_ff, d, ws,
                               mm=mm, dyb=dxb[0], xb=c.get("gb"))
                if c["f"].dtype != t.float32:       # a bf16 pre-activation met the unfused backward (dx not staged): widen it for the elementwise kernel
                    c["f"] = c["f"][:, :self.d_ff].float().contiguous()
                if mm is not None:         # df only feeds the c_fc backward products: written as their bf16 operand, no fp32 copy
                    df, dfb = None, ops.gelu_bwd_staged(mm, dg, c["f"], R, self.d_ff)
                else:
                    df, dfb = dg, None
                    ops.gelu_bwd(dg, c["f"], df)
            dh2 = new(R, d)
            ops.linear_bwd(c["h2"], p[q + "mlp.c_fc.weight"], df, dh2, grads[q + "mlp.c_fc.weight"], grads[q + "mlp.c_fc.bias"], R, d, self.d_ff, ws, mm=mm,
                           dyb=dfb, xb=c.get("h2b"))
            ln_bwd(dh2, c["x_mid"], q + "ln_2", c["m2"], c["r2"], True)    # dx := dx_mid
            # attention projection
            datt = new(R, d)
            ops.linear_bwd(c["att"], p[q + "attn.c_proj.weight"], dx, datt, grads[q + "attn.c_proj.weight"], grads[q + "attn.c_proj.bias"], R, d, d, ws, mm=mm,
                           dyb=dxb[0], xb=c.get("attb"))
            qkv, P = c["qkv"], c["P"]
            dqkv, dqkvb = None, None
            if cache["flash"] and mm is not None:
                dqkvb = ops.flash_attn_bwd_staged(mm, qkv, cache["km"], c["att"], datt, c["lse"], c["flash_ws"], B, H, T, qkv_staged=True,
                                                  attb=c.get("attb"), ld_attb=ops._pitch(d))
            elif cache["flash"]:
                dqkv = new(R, 3 * d)
                ops.flash_attn_bwd(qkv, cache["km"], c["att"], datt, c["lse"], dqkv, c["flash_ws"], B, H, T, False, qkv_staged=True)
            else:
                dqkv = new(R, 3 * d)
                self._attention_bwd_materialized(qkv, P, datt, dqkv, B, T, H, hd, d, scale, new)
            dh1 = new(R, d)
            ops.linear_bwd(c["h1"], p[q + "attn.c_attn.weight"], dqkv, dh1, grads[q + "attn.c_attn.weight"], grads[q + "attn.c_attn.bias"], R, d, 3 * d, ws, mm=mm,
                           dyb=dqkvb, xb=c.get("h1b"))
            ln_bwd(dh1, c["x_in"], q + "ln_1", c["m1"], c["r1"], True)     # dx := dx_in
            if on_final is not None:
                on_final([q + n for n in ("mlp.c_proj.weight", "mlp.c_proj.bias", "mlp.c_fc.weight", "mlp.c_fc.bias", "ln_2.weight", "ln_2.bias",
                                          "attn.c_proj.weight", "attn.c_proj.bias", "attn.c_attn.weight", "attn.c_attn.bias", "ln_1.weight", "ln_1.bias")])
        # positions with attention_mask 0 whose NEXT position is masked too are never read by a loss term nor attended to: their dx is exactly zero —
        # skipped (ops.embed_bwd; with right padding that is every padded position, with left padding / holes the row before an attended one stays)
        ops.embed_bwd(dx, cache["ids"], cache["pos"], grads["wte.weight"], grads["wpe.weight"], R, d, live=cache["km"], vocab=self.vocab, t_row=cache["T"])
        if on_final is not None:
            on_final(["wpe.weight", "wte.weight"])
        return grads

    def _attention_bwd_materialized(self, qkv, P, datt, dqkv, B, T, H, hd, d, scale, new):
        """dqkv from the stored probabilities P [B*H, T, T] (batched sgemm + softmax backward)."""
        # dV = P^T dA
        ops.sgemm(P, datt, dqkv, T, hd, T, trans_a=True, lda=T, ldb=d, ldc=3 * d, c_off=2 * d, batch=(B, H), sa=(H * T * T, T * T),
                  sb=(T * d, hd), sc=(T * 3 * d, hd))
        # dP = dA V^T ; dS = softmax_bwd
        dP = new(B * H, T, T)
        ops.sgemm(datt, qkv, dP, T, T, hd, trans_b=True, lda=d, ldb=3 * d, ldc=T, b_off=2 * d, batch=(B, H), sa=(T * d, hd),
                  sb=(T * 3 * d, hd), sc=(H * T * T, T * T))
        ops.softmax_bwd(P, dP, B * H * T, T)
        # dQ = scale * dS K ; dK = scale * dS^T Q
        ops.sgemm(dP, qkv, dqkv, T, hd, T, alpha=scale, lda=T, ldb=3 * d, ldc=3 * d, b_off=d, batch=(B, H), sa=(H * T * T, T * T),
                  sb=(T * 3 * d, hd), sc=(T * 3 * d, hd))
        ops.sgemm(dP, qkv, dqkv, T, hd, T, trans_a=True, alpha=scale, lda=T, ldb=3 * d, ldc=3 * d, c_off=d, batch=(B, H),
                  sa=(H * T * T, T * T), sb=(T * 3 * d, hd), sc=(T * 3 * d, hd))


# ---------------------------------------------------------------------- value heads (LLM_RL/heads/{linear_head,mlp_head}.py)
class LinearHeadF32:
    """`LinearHead`: x @ kernel + bias (heads/linear_head.py:112-119).  matmul="bf16": see GPT2F32 (outputs narrower than 64 columns,
    e.g. the scalar value head, stay on the fp32 kernel: nothing to gain on the matrix core)."""

    def __init__(self, params, device, matmul: str = "f32"):
        import torch
        self.t, self.dev = torch, device
        self.p = {k: v.to(device, torch.float32).contiguous() for k, v in params.items()}   # kernel [in,out], bias [out]
        self.din, self.dout = self.p["kernel"].shape
        self._ws = torch.empty(64 * max(self.dout, self.din), dtype=torch.float32, device=device)   # colsum / matrix-vector scratch
        self.mm = ops.MatmulBF16(device) if matmul == "bf16" and self.dout >= 64 else None
        self.ld_out = ops._pad(self.dout) if self.mm is not None else self.dout

    def forward(self, x, rows):
        """-> (y [rows, ld_out] with columns [0, dout) valid, cache)"""
        y = self.t.empty(rows, self.ld_out, dtype=self.t.float32, device=self.dev)
        if self.mm is not None:
            self.mm.begin_step()
        ops.linear_fwd(x, self.p["kernel"], self.p["bias"], y, rows, self.din, self.dout, mm=self.mm, ldy=self.ld_out)
        return y, dict(x=x, rows=rows)

    def ce_bwd(self, y, lse, targets, coef_ce, coef_gather, rows):
        """d loss / d y of the CE (+ gather) terms -> (dy, dyb) for `backward` (see GPT2F32.ce_bwd)"""
        if self.mm is None:
            ops.ce_bwd(y, self.ld_out, self.dout, lse, targets, coef_ce, coef_gather, rows)
            return y, None
        return None, ops.ce_bwd_staged(self.mm, y, self.ld_out, self.dout, lse, targets, coef_ce, coef_gather, rows)

    def backward(self, cache, dy, grads, dx=None, accumulate_dx=False, dyb=None):
        ops.linear_bwd(cache["x"], self.p["kernel"], dy, dx, grads["kernel"], grads["bias"], cache["rows"], self.din, self.dout, self._ws,
                       dx_beta=1.0 if accumulate_dx else 0.0, mm=self.mm, lddy=self.ld_out, dyb=dyb)

    def zero_grads(self):
        if getattr(self, "_arena", None) is None:
            self._arena = GradArena(self.p)
        return self._arena.zero_()


class MLPHeadF32:
    """`MLPHead`: relu(x @ W1 + b1) @ W2 + b2 (heads/mlp_head.py:139-148). params: dense1.kernel/bias, dense2.kernel/bias."""

    def __init__(self, params, device, matmul: str = "f32"):
        import torch
        self.t, self.dev = torch, device
        self.p = {k: v.to(device, torch.float32).contiguous() for k, v in params.items()}
        self.din, self.dh = self.p["dense1.kernel"].shape
        self.dout = self.p["dense2.kernel"].shape[1]
        self._ws = torch.empty(64 * max(self.dout, self.dh), dtype=torch.float32, device=device)
        self.mm = ops.MatmulBF16(device) if matmul == "bf16" else None
        self.mm2 = self.mm if self.dout >= 64 else None           # a scalar output (V head) stays on the fp32 kernel
        # row pitch of the [rows, dout] outputs: whole 64-column groups in the bf16-matmul mode; fp32: 16-byte aligned rows for a vocabulary-wide
        # head (its dlogits are the K-contiguous operand of the dX product: 50 257 floats per row would force 4-byte loads)
        self.ld_out = ops._pad(self.dout) if self.mm2 is not None else (ops._pad(self.dout, 4) if self.dout >= 64 else self.dout)

    def hidden(self, x, rows):
        """relu(x @ W1 + b1) -> (a, z)"""
        t = self.t
        if self.mm is not None:
            self.mm.begin_step()          # a forward starts here: re-stage the bf16 weight copies (see GPT2F32.forward)
        z = t.empty(rows, self.dh, dtype=t.float32, device=self.dev)
        ops.linear_fwd(x, self.p["dense1.kernel"], self.p["dense1.bias"], z, rows, self.din, self.dh, mm=self.mm)
        a = t.empty_like(z)
        ops.relu_fwd(z, a)
        return a, z

    def _w2(self, refresh: bool):
        """(dense2.kernel operand, its row pitch) for the fp32 products: a vocabulary-wide kernel has 50 257-float rows, none of them 16-byte
        aligned, which forces the guarded 4-byte loads on the whole dX product (74 vs 113 TFLOP/s); a copy with the rows padded to a multiple of
        4 floats, refreshed once per forward (the optimizer moves the master), takes the 16-byte path.  Same values, same products."""
        w = self.p["dense2.kernel"]
        if self.mm2 is not None or self.dout < 64 or self.dout % 4 == 0:
            return w, None
        ld = ops._pad(self.dout, 4)
        if getattr(self, "_w2_al", None) is None:
            self._w2_al = self.t.zeros(self.dh, ld, dtype=self.t.float32, device=self.dev)
            refresh = True
        if refresh:
            self._w2_al[:, :self.dout].copy_(w)
        return self._w2_al, ld

    def forward(self, x, rows):
        """-> (y [rows, ld_out] with columns [0, dout) valid, cache)"""
        t = self.t
        a, z = self.hidden(x, rows)
        y = t.empty(rows, self.ld_out, dtype=t.float32, device=self.dev)
        w2, ldw = self._w2(True)
        ops.linear_fwd(a, w2, self.p["dense2.bias"], y, rows, self.dh, self.dout, mm=self.mm2, ldy=self.ld_out, ldw=ldw)
        return y, dict(x=x, z=z, a=a, rows=rows)

    def fused_ce_ok(self) -> bool:
        return ops.FUSE_CE and self.mm2 is not None and self.dh % 64 == 0 and ops._padn(self.dout) % 128 == 0

    def forward_ce(self, x, rows, targets):
        """The head's logits only as far as the CE / take_along_axis terms need them (bf16-matmul mode): -> (lse [rows], target logit [rows],
        target log-probability [rows], cache); the cache keeps the bf16 logits for `ce_bwd_fused`."""
        a, z = self.hidden(x, rows)
        yb, lse, tl, lp = ops.head_fwd_ce(self.mm2, a, self.p["dense2.kernel"], self.p["dense2.bias"], rows, self.dh, self.dout, targets)
        return lse, tl, lp, dict(x=x, z=z, a=a, rows=rows, yb=yb)

    def ce_bwd_fused(self, cache, lse, targets, coef_ce, coef_gather, rows):
        """d loss / d logits of the CE (+ gather) terms as the bf16 dy operand of `backward(dyb=...)`, formed in place on the cached logits"""
        return ops.ce_bwd_inplace(cache["yb"], self.dout, lse, targets, coef_ce, coef_gather, rows)

    def forward_at(self, x, rows, idx):
        """y[r, idx[r]] for every row without forming y: `take_along_axis(head(x), idx)` — how the ILQL loss reads the TARGET Q heads
        (ilql/base_interface.py:57-66).  dense2 shrinks from a [rows, dh] x [dh, V] product to one dh-long dot product per row (fp32)."""
        a, _ = self.hidden(x, rows)
        out = self.t.empty(rows, dtype=self.t.float32, device=self.dev)
        ops.gather_dot(a, self.p["dense2.kernel"], self.p["dense2.bias"], idx, out, rows, self.dh, self.dout)
        return out

    def ce_bwd(self, y, lse, targets, coef_ce, coef_gather, rows):
        """d loss / d y of the CE (+ gather) terms -> (dy, dyb) for `backward` (see GPT2F32.ce_bwd)"""
        if self.mm2 is None:
            ops.ce_bwd(y, self.ld_out, self.dout, lse, targets, coef_ce, coef_gather, rows)
            return y, None
        return None, ops.ce_bwd_staged(self.mm2, y, self.ld_out, self.dout, lse, targets, coef_ce, coef_gather, rows)

    def backward(self, cache, dy, grads, dx=None, accumulate_dx=False, dyb=None):
        t, rows = self.t, cache["rows"]
        da = t.empty(rows, self.dh, dtype=t.float32, device=self.dev)
        w2, ldw = self._w2(False)
        ops.linear_bwd(cache["a"], w2, dy, da, grads["dense2.kernel"], grads["dense2.bias"], rows, self.dh, self.dout, self._ws,
                       mm=self.mm2, lddy=self.ld_out, dyb=dyb, ldw=ldw)
        ops.relu_bwd(da, cache["z"], da)
        ops.linear_bwd(cache["x"], self.p["dense1.kernel"], da, dx, grads["dense1.kernel"], grads["dense1.bias"], rows, self.din, self.dh, self._ws,
                       dx_beta=1.0 if accumulate_dx else 0.0, mm=self.mm)

    def zero_grads(self):
        if getattr(self, "_arena", None) is None:
            self._arena = GradArena(self.p)
        return self._arena.zero_()


# ---------------------------------------------------------------------- optimizer
class AdamW:
    """optax.MultiSteps(optax.adamw(lr, b1, b2, eps, weight_decay, mask), every_k_schedule=k)
    (llm_rl_scripts/wordle/ilql/train_ilql_gpt2.py:155-186): gradients are averaged over k micro-steps, then one AdamW
    update; weight decay skips biases and LayerNorm parameters (the mask)."""

    def __init__(self, params: Dict[str, "torch.Tensor"], lr, b1=0.9, b2=0.95, eps=1e-8, weight_decay=0.0, every_k: int = 1,
                 no_decay=lambda name: name.endswith("bias") or ".ln_" in name or name.startswith("ln_f")):
        import torch
        self.t = torch
        self.params = params
        self.lr, self.b1, self.b2, self.eps, self.wd, self.k = lr, b1, b2, eps, weight_decay, every_k
        self.no_decay = no_decay
        self.arena = isinstance(params, GradArena)      # parameters in one flat buffer: one launch per update instead of one per tensor
        if self.arena:
            self.m, self.v = GradArena(params, list(params.order)), GradArena(params, list(params.order))
            self.acc = GradArena(params, list(params.order)) if every_k > 1 else None
            ends, wds = [], []
            for k, (off, n) in params.order.items():
                ends.append(off + n)
                wds.append(0.0 if no_decay(k) else weight_decay)
            dev = params.flat.device
            self._seg_end = torch.tensor(ends, dtype=torch.int64, device=dev)
            self._seg_wd = torch.tensor(wds, dtype=torch.float32, device=dev)
        else:
            self.m = {k: torch.zeros_like(v) for k, v in params.items()}
            self.v = {k: torch.zeros_like(v) for k, v in params.items()}
            self.acc = {k: torch.zeros_like(v) for k, v in params.items()} if every_k > 1 else None
        self.step_count = 0     # number of applied updates
        self.mini_step = 0

    def apply(self, grads: Dict[str, "torch.Tensor"], polyak=None) -> bool:
        """Returns True when parameters were updated on this call (MultiSteps.mini_step wrapped to 0).
        polyak = (target parameter arena of the same layout, alpha): when this call updates the parameters, the target's soft update
        target = alpha p_new + (1 - alpha) target rides in the AdamW sweep (`self.polyak_fused` says whether it did — it does on the arena path)."""
        fast = self.arena and self.params.same_layout(grads)
        self.polyak_fused = False
        if self.k > 1:
            if fast:
                ops.axpby(1.0, self.acc.flat, 1.0 / self.k, grads.flat, self.acc.flat)
            else:
                for k, g in grads.items():
                    ops.axpby(1.0, self.acc[k], 1.0 / self.k, g, self.acc[k])
            self.mini_step += 1
            if self.mini_step < self.k:
                return False
            self.mini_step = 0
            grads = self.acc
        self.step_count += 1
        if fast:
            tgt = None
            if polyak is not None and getattr(polyak[0], "flat", None) is not None and polyak[0].order == self.params.order:
                tgt, self.polyak_fused = polyak[0].flat, True
            ops.adamw_segments(self.params.flat, grads.flat, self.m.flat, self.v.flat, self._seg_end, self._seg_wd, self.lr, self.b1, self.b2,
                               self.eps, self.step_count, target=tgt, alpha=polyak[1] if tgt is not None else 0.0)
        else:
            for k, p in self.params.items():
                wd = 0.0 if self.no_decay(k) else self.wd
                ops.adamw(p, grads[k], self.m[k], self.v[k], self.lr, self.b1, self.b2, self.eps, wd, self.step_count)
        if self.k > 1:
            if self.arena:
                self.acc.flat.zero_()
            else:
                for a in self.acc.values():
                    a.zero_()
        return True
