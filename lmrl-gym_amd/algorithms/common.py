"""Shared host-side pieces: blocking of ragged sequences (JaxSeq.utils.block_sequences semantics as used at
LLM_RL/algorithms/ppo/data.py:18-60), attention-mask / position-id initialisation
(JaxSeq.models.base_interface.initialize_attn_mask_pos_ids, call sites ppo/base_interface.py:190-195), and the
masked tensor statistics of LLM_RL/utils.py:12-21 finalised from device partial sums."""
from __future__ import annotations

import enum
import math
from typing import Dict, NamedTuple, Optional, Sequence

import numpy as np


class Padding(enum.Enum):
    LEFT = "left"
    RIGHT = "right"


class Truncation(enum.Enum):
    LEFT = "left"
    RIGHT = "right"


class BlockingStrategy(NamedTuple):
    padding: Padding
    truncation: Truncation
    max_length: Optional[int]


def block_sequences(sequences: Sequence[Sequence], pad_value, dtype, blocking_strategy: BlockingStrategy) -> np.ndarray:
    max_len = blocking_strategy.max_length
    if max_len is None:
        max_len = max((len(s) for s in sequences), default=0)
    out = np.full((len(sequences), max_len), pad_value, dtype=dtype)
    for i, s in enumerate(sequences):
        s = np.asarray(s)
        if len(s) > max_len:
            s = s[:max_len] if blocking_strategy.truncation == Truncation.RIGHT else s[len(s) - max_len:]
        if blocking_strategy.padding == Padding.RIGHT:
            out[i, : len(s)] = s
        else:
            out[i, max_len - len(s):] = s
    return out


def initialize_attn_mask_pos_ids(input_ids: np.ndarray, pad_token_id: Optional[int], attention_mask=None, position_ids=None):
    """attention_mask = (ids != pad); position_ids = clip(cumsum(mask) - 1, 0)."""
    if attention_mask is None:
        attention_mask = (input_ids != pad_token_id) if pad_token_id is not None else np.ones_like(input_ids, dtype=bool)
    attention_mask = np.asarray(attention_mask).astype(np.int32)
    if position_ids is None:
        position_ids = np.maximum(np.cumsum(attention_mask, axis=1) - 1, 0)
    return attention_mask, np.asarray(position_ids, dtype=np.int32)


def masked_rows(mask_bt1: np.ndarray, T: int) -> np.ndarray:
    """Flat row indices r = b*T + t (int32, increasing) of the set entries of a [B, T-1] mask on the shifted grid — the rows of the
    [B*T, d] hidden-state matrix whose vocabulary-wide head outputs a masked loss actually reads (the train steps run those heads on the
    gathered rows only: every term of ppo_loss_fn / ilql_loss / mc_loss / the BC loss carries such a mask)."""
    m = np.asarray(mask_bt1, dtype=bool)
    B, T1 = m.shape
    assert T1 == T - 1
    return (np.arange(B, dtype=np.int64)[:, None] * T + np.arange(T1, dtype=np.int64)[None, :])[m].astype(np.int32)


def stats_from_sums(sum_w, sum_b, sumsq_b, mn, mx, count_b, n) -> Dict[str, np.float32]:
    """get_tensor_stats: mean = sum(x*mask)/n ; min/max/std over mask.astype(bool) (population std)."""
    f = np.float32
    if count_b <= 0:
        return dict(mean=f(sum_w / n) if n else f("nan"), min=f("inf"), max=f("-inf"), std=f("nan"))
    mb = sum_b / count_b
    var = max(sumsq_b / count_b - mb * mb, 0.0)
    return dict(mean=f(sum_w / n), min=f(mn), max=f(mx), std=f(math.sqrt(var)))
