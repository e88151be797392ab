"""Behaviour cloning — counterpart of LLM_RL/algorithms/bc/{interface,data}.py: `bc_loss` (masked next-token CE with a
weight on non-action tokens), `block_token_histories`, `filter_items`, and a fp32 `GPT2BCTrain.step`."""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Optional, Tuple

import numpy as np

from .. import dist as D
from ..train import ops
from ..train.gpt2_f32 import AdamW, GPT2F32
from .common import BlockingStrategy, Padding, Truncation, block_sequences, initialize_attn_mask_pos_ids, masked_rows
from .ppo import _t


def block_token_histories(token_histories, max_len: Optional[int], pad_token_id: int) -> Tuple[np.ndarray, np.ndarray]:
    """bc/data.py:10-28."""
    bs = BlockingStrategy(padding=Padding.RIGHT, truncation=Truncation.RIGHT, max_length=max_len)
    return (block_sequences([th.tokens for th in token_histories], pad_token_id, np.int32, bs),
            block_sequences([th.is_action for th in token_histories], False, np.bool_, bs))


def filter_items(score_fn: Callable[[Any], float], items: List[Any], take_top: Optional[float] = None,
                 threshold: Optional[float] = None) -> List[Any]:
    """%BC filtering (bc/data.py:32-47)."""
    assert (take_top is None) != (threshold is None)
    scores = np.array([score_fn(item) for item in items])
    if take_top is not None:
        threshold = np.percentile(scores, 100 - take_top)
    return [item for item, s in zip(items, scores) if s >= threshold]


def bc_weights(attention_mask: np.ndarray, is_action: np.ndarray, non_action_weight: float) -> Tuple[np.ndarray, float]:
    """Per-token CE weights of bc/interface.py:39-42: attn * (is_action + (1-is_action)*w) on the shifted grid, and the
    normaliser sum(attn[:, 1:])."""
    am = np.asarray(attention_mask, dtype=np.float32)[:, 1:]
    ia = np.asarray(is_action, dtype=np.float32)[:, 1:]
    return am * (ia + (1.0 - ia) * np.float32(non_action_weight)), float(am.sum())


def masked_ce_forward_backward(m: GPT2F32, ids: np.ndarray, am: np.ndarray, pos: np.ndarray, w: np.ndarray, denom: float,
                               grads=None, grad_scale: float = 1.0, compact_rows: bool = True) -> float:
    """loss = sum(w * CE(logits[:, :-1], ids[:, 1:])) / denom with the masked sums all-reduced across ranks; when `grads`
    is given, grad_scale * d loss / d params is ACCUMULATED into it (shared by the BC trainer and the PPO BC term)."""
    import torch
    B, T = ids.shape
    R, dev = B * T, m.dev
    ids_d = _t(ids, np.int32)
    hid, cache = m.forward(ids_d, _t(am, np.uint8), _t(pos, np.int32))
    if D.is_distributed():
        denom = float(D.allreduce_sum_(torch.tensor([denom], dtype=torch.float64, device=dev)).item())
    wn = np.asarray(w, dtype=np.float32) / np.float32(denom)                     # [B, T-1] CE weights
    # the LM head runs on the rows with a non-zero weight only (with non_action_weight = 0: the action tokens) — rows of weight 0
    # contribute exact zeros to the loss and to every gradient
    rows_h = masked_rows(wn != 0, T)
    Ra = int(rows_h.size)
    compact = compact_rows and 0 < Ra < R
    if compact:
        idx = _t(rows_h, np.int32)
        hq, Rq = ops.gather_rows(hid, idx, Ra, m.d), Ra
        tgt = _t(ids[:, 1:][wn != 0].astype(np.int32), np.int32)
        coef = _t(wn[wn != 0].astype(np.float32), np.float32)
    else:
        hq, Rq = hid, R
        tgt = torch.zeros(R, dtype=torch.int32, device=dev)
        tgt.view(B, T)[:, :-1] = ids_d[:, 1:]
        wfull = np.zeros((B, T), dtype=np.float32)
        wfull[:, :-1] = wn
        coef = _t(wfull.reshape(-1), np.float32)
    logits, logits_b, lse, lp = m.lm_ce(hq, Rq, tgt)
    # loss = sum(coef * CE) = -sum(coef * logprob): a dot product done as a 1 x 1 x R GEMM on the matrix core
    out = torch.zeros(1, dtype=torch.float32, device=dev)
    ops.sgemm(coef, lp, out, 1, 1, Rq, alpha=-1.0, lda=Rq, ldb=1, ldc=1)
    if D.is_distributed():
        D.allreduce_sum_(out)
    loss = float(out.item())
    if grads is not None:
        if grad_scale != 1.0:
            ops.axpby(grad_scale, coef, 0.0, None, coef)
        dlogits, dlb = m.ce_bwd_any(logits, logits_b, lse, tgt, coef, None, Rq)
        if compact:
            dhq = torch.empty(Ra, m.d, dtype=torch.float32, device=dev)
            m.lm_head_backward(hq, dlogits, Ra, dhq, grads, accumulate_dh=False, dlb=dlb)
            d_hidden = torch.zeros(R, m.d, dtype=torch.float32, device=dev)
            ops.scatter_rows(dhq, idx, d_hidden, Ra, m.d, False)
        else:
            d_hidden = torch.empty(R, m.d, dtype=torch.float32, device=dev)
            m.lm_head_backward(hid, dlogits, R, d_hidden, grads, accumulate_dh=False, dlb=dlb)
        m.backward(cache, d_hidden, grads)
    return loss


class GPT2BCTrain:
    """fp32 BC trainer: loss = sum(w * CE) / sum(attn[:, 1:])  (bc/interface.py:28-43)."""

    def __init__(self, model: GPT2F32, pad_token_id: int, non_action_weight: float = 0.0, lr: float = 1e-4, weight_decay: float = 0.0,
                 grad_accum_steps: int = 1):
        self.model, self.pad, self.w = model, pad_token_id, non_action_weight
        self.opt = AdamW(model.p, lr, weight_decay=weight_decay, every_k=grad_accum_steps)
        self.last_grads = None

    def step(self, input_ids, is_action, attention_mask=None, position_ids=None, train: bool = True):
        ids = np.asarray(input_ids, dtype=np.int32)
        am, pos = initialize_attn_mask_pos_ids(ids, self.pad, attention_mask, position_ids)
        w, denom = bc_weights(am, is_action, self.w)
        grads = self.model.zero_grads() if train else None
        loss = masked_ce_forward_backward(self.model, ids, am, pos, w, denom, grads)
        if not train:
            return self, loss, {"loss": np.float32(loss)}
        self.last_grads = grads
        D.allreduce_grads([grads])
        self.opt.apply(grads)
        return self, loss, {"loss": np.float32(loss)}
