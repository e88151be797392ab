"""Oracle logits warpers (TEST INFRASTRUCTURE): numpy float64 restatement of the sampling pipeline the task scripts configure —
`GenerationConfig(do_sample, temperature, top_k, top_p)` (llm_rl_scripts/wordle/ppo/train_ppo_gpt2.py:98-99, 218-227) run by HF-Flax `generate`:
FlaxTemperatureLogitsWarper -> FlaxTopKLogitsWarper -> FlaxTopPLogitsWarper -> jax.random.categorical.

The Flax warpers are third party (transformers' Flax side needs jax: not installed).  What IS pinned: tests/test_oracle_warpers.py checks the kept
sets of this restatement against the installed `transformers` PyTorch twins (`TemperatureLogitsWarper`, `TopKLogitsWarper`, `TopPLogitsWarper`: the
same library's implementation of the same three processors; for top-p the PyTorch form — ascending sort, drop while the cumulative mass stays
<= 1 - top_p — and the Flax form — descending sort, keep while the mass BEFORE a token is < top_p — select the same set; of two EXACTLY tied
tokens at the crossing the processors keep whichever their sort puts first — the HIP kernels keep both: every test skips rows whose crossing is a tie).
"""
from __future__ import annotations

import numpy as np


def warp(logits: np.ndarray, temperature: float = 1.0, top_k: int = 0, top_p: float = 0.0):
    """logits [..., V] -> (scores = logits / temperature as float64, keep mask [..., V]).
    top_k (0 / >= V: off): keep every score >= the k-th largest (ties kept: `scores < kth` is what the warper removes).
    top_p (0 / >= 1: off), applied to what top-k kept: in descending order, keep a token while the renormalised probability mass of the tokens
    BEFORE it is < top_p (the first token always; the token that crosses top_p is kept)."""
    z = np.asarray(logits, dtype=np.float64) / float(temperature)
    V = z.shape[-1]
    keep = np.ones(z.shape, dtype=bool)
    if 0 < top_k < V:
        kth = np.sort(z, axis=-1)[..., V - top_k][..., None]
        keep &= z >= kth
    if 0.0 < top_p < 1.0:
        zk = np.where(keep, z, -np.inf)
        order = np.argsort(-zk, axis=-1, kind="stable")
        zs = np.take_along_axis(zk, order, axis=-1)
        p = np.exp(zs - zs[..., :1])
        p = p / p.sum(axis=-1, keepdims=True)
        before = np.cumsum(p, axis=-1) - p
        ks = before < top_p
        ks[..., 0] = True
        kp = np.zeros_like(keep)
        np.put_along_axis(kp, order, ks, axis=-1)
        keep &= kp
    return z, keep


def log_probs(z: np.ndarray, keep: np.ndarray) -> np.ndarray:
    """log of the renormalised sampling distribution over the kept set (-inf outside)."""
    zk = np.where(keep, z, -np.inf)
    m = zk.max(axis=-1, keepdims=True)
    return zk - (m + np.log(np.exp(zk - m).sum(axis=-1, keepdims=True)))
