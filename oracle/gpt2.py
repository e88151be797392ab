"""Oracle GPT-2 forward (TEST INFRASTRUCTURE): plain torch-CPU restatement of the GPT-2 architecture
(pre-LN blocks, gelu_new, learned positions, tied LM head, LN eps 1e-5) in float32/float64.

The reference's transformer lives in JaxSeq / HF-Flax (third party, not in /root/reference, not installed:
SURVEY.md §2.3) -> "parity unpinned" against the JAX path.  What IS pinned: tests/test_oracle_gpt2.py checks
this restatement against the installed HF PyTorch `GPT2LMHeadModel` (same architecture family the reference's
HF-Flax model implements) on random-init weights.
State-dict names follow HF GPT-2 (`wte.weight`, `h.0.attn.c_attn.weight` [in, out], ...).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch


def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def layer_norm(x, g, b, eps):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def forward(sd: Dict[str, torch.Tensor], input_ids: torch.Tensor, n_head: int, eps: float = 1e-5,
            attention_mask: Optional[torch.Tensor] = None, position_ids: Optional[torch.Tensor] = None,
            dtype=torch.float64, return_hidden: bool = False):
    """input_ids [B, T] -> logits [B, T, V] (and final-LN hidden states [B, T, d])."""
    sd = {k: v.to(dtype) for k, v in sd.items()}
    B, T = input_ids.shape
    if position_ids is None:
        position_ids = torch.arange(T).unsqueeze(0).expand(B, T)
    x = sd["wte.weight"][input_ids] + sd["wpe.weight"][position_ids]
    d = x.shape[-1]
    hd = d // n_head
    causal = torch.tril(torch.ones(T, T, dtype=torch.bool))
    n_layer = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("h."))
    for l in range(n_layer):
        p = f"h.{l}."
        h = layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], eps)
        qkv = h @ sd[p + "attn.c_attn.weight"] + sd[p + "attn.c_attn.bias"]
        q, k, v = (t.reshape(B, T, n_head, hd).transpose(1, 2) for t in qkv.split(d, dim=-1))
        att = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
        mask = causal[None, None]
        if attention_mask is not None:
            mask = mask & attention_mask[:, None, None, :].bool()
        att = att.masked_fill(~mask, float("-inf")).softmax(-1)
        a = (att @ v).transpose(1, 2).reshape(B, T, d)
        x = x + a @ sd[p + "attn.c_proj.weight"] + sd[p + "attn.c_proj.bias"]
        h = layer_norm(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], eps)
        x = x + gelu_new(h @ sd[p + "mlp.c_fc.weight"] + sd[p + "mlp.c_fc.bias"]) @ sd[p + "mlp.c_proj.weight"] + sd[p + "mlp.c_proj.bias"]
    hid = layer_norm(x, sd["ln_f.weight"], sd["ln_f.bias"], eps)
    logits = hid @ sd["wte.weight"].t()
    return (logits, hid) if return_hidden else logits


def round_weights_to_bf16(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The engine stores matrices in bf16 (LN params / biases stay f32): round the same way so that the oracle
    and the engine see identical parameters."""
    out = {}
    for k, v in sd.items():
        is_mat = k.endswith("wte.weight") or k.endswith("wpe.weight") or (k.endswith(".weight") and v.dim() == 2)
        out[k] = v.float().to(torch.bfloat16).float() if is_mat else v.float()
    return out


# ---- Philox4x32-R + Gumbel, restating csrc/sampler.hip's documented random stream (numpy, uint64 math); the sampler
# uses R = 7 rounds, the Random123 known-answer vectors (tests/test_oracle_gpt2.py) pin the R = 10 function
def philox4x32_10(c0, c1, c2, c3, k0, k1, rounds: int = 10):
    import numpy as np
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint64) for x in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0), np.uint64(k1)
    M = np.uint64(0xFFFFFFFF)
    for _ in range(rounds):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & M
        n1 = p1 & M
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & M
        n3 = p0 & M
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & M
        k1 = (k1 + np.uint64(0xBB67AE85)) & M
    return c0, c1, c2, c3


def gumbel_noise(rows: int, vocab: int, seed: int, step: int, epoch: int = 0):
    """[rows, vocab] float32 Gumbel noise exactly as the kernels draw it: counter (row, col//4, step, epoch)."""
    import numpy as np
    ncol4 = (vocab + 3) // 4
    r = np.repeat(np.arange(rows, dtype=np.uint64), ncol4)
    c = np.tile(np.arange(ncol4, dtype=np.uint64), rows)
    o = philox4x32_10(r, c, np.full_like(r, step), np.full_like(r, epoch), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, rounds=7)
    bits = np.stack(o, axis=1).reshape(rows, ncol4 * 4)[:, :vocab]
    u = ((bits >> np.uint64(9)).astype(np.float32) + np.float32(0.5)) * np.float32(1.1920928955078125e-07)
    return -np.log(-np.log(u))
