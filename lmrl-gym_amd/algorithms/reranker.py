"""Reranker policies and log-prob / advantage score functions — counterparts of
LLM_RL/algorithms/ppo/reranker_policy.py:5-34, LLM_RL/algorithms/ppo/score_fn.py:10-126 and
LLM_RL/algorithms/ilql/gpt2/score_fn.py:11-68."""
from __future__ import annotations

from typing import Callable, List, Optional

import numpy as np

from ..environment import TextHistory, TextPolicy, TokenHistory
from ..train import ops
from .ppo import _t


class ReRankerSamplePolicy(TextPolicy):
    def __init__(self, proposal_fn, score_fn: Callable[[List[TextHistory]], List[float]]):
        self.proposal_fn, self.score_fn = proposal_fn, score_fn

    def act(self, text_history: TextHistory) -> TextHistory:
        proposals = self.proposal_fn(text_history)
        scores = np.asarray(self.score_fn(proposals), dtype=np.float32)
        p = np.exp(scores) / np.exp(scores).sum()
        return proposals[np.random.choice(len(p), p=p)]


class ReRankerPolicy(TextPolicy):
    def __init__(self, proposal_fn: Callable[[TextHistory], List[TextHistory]], score_fn: Callable[[List[TextHistory]], List[float]]):
        self.proposal_fn, self.score_fn = proposal_fn, score_fn

    def act(self, text_history: TextHistory) -> TextHistory:
        proposals = self.proposal_fn(text_history)
        return proposals[np.argmax(np.asarray(self.score_fn(proposals), dtype=np.float32)).item()]


def _tokens_for_scoring(text_histories, tokenizer, max_length):
    assert all(th[-1].is_action for th in text_histories)
    prev = [TokenHistory.from_text_history(th[:-1], tokenizer) for th in text_histories]
    full = [TokenHistory.from_text_history(th, tokenizer) for th in text_histories]
    tokens = np.stack([np.concatenate((t.tokens[-max_length:], np.full((max_length - min(t.tokens.shape[0], max_length),), tokenizer.pad_token_id)))
                       for t in full]).astype(np.int32)
    return tokens, np.array([p.tokens.shape[0] for p in prev], dtype=np.int64), full


def build_logprob_score_fn(model, tokenizer, max_length: int, bsize: int):
    """build_ppo_score_fn / build_bc_score_fn: sum of the last action's token log-probs under `model` (a GPT2F32)."""

    def score_fn(text_histories: List[TextHistory]) -> List[float]:
        import torch
        tokens, prefix_len, _ = _tokens_for_scoring(text_histories, tokenizer, max_length)
        out: List[float] = []
        for i in range(0, len(text_histories), bsize):
            tb = tokens[i:i + bsize]
            am = (tb != tokenizer.pad_token_id)
            pos = np.maximum(np.cumsum(am, axis=1) - 1, 0).astype(np.int32)
            B, T = tb.shape
            ids_d = _t(tb, np.int32)
            hid, _ = model.forward(ids_d, _t(am, np.uint8), _t(pos, np.int32))
            logits = model.lm_logits(hid, B * T)
            tgt = torch.zeros(B * T, dtype=torch.int32, device=model.dev)
            tgt.view(B, T)[:, :-1] = ids_d[:, 1:]
            lp = torch.empty(B * T, dtype=torch.float32, device=model.dev)
            ops.lse_gather(logits, model.ld_vocab, model.vocab, tgt, B * T, logprob=lp)
            lpn = lp.view(B, T)[:, :-1].cpu().numpy() * am[:, 1:]
            for x in range(B):
                out.append(float(lpn[x][(prefix_len[i + x] - 1):].sum()))
        return out

    return score_fn


def build_ppo_score_fn(inference, tokenizer, max_length: int, bsize: int):
    """ppo/score_fn.py:10-66 signature: `inference` is a `GPT2PPOInference` (its policy is scored) or a bare `GPT2F32`."""
    return build_logprob_score_fn(getattr(inference, "policy", inference), tokenizer, max_length, bsize)


def build_bc_score_fn(inference, tokenizer, max_length: int, bsize: int):
    """ppo/score_fn.py:69-126 signature: `inference` is a `GPT2F32` (or any object with a `.model` / `.policy` GPT2F32)."""
    return build_logprob_score_fn(getattr(inference, "model", getattr(inference, "policy", inference)), tokenizer, max_length, bsize)


def build_ilql_score_fn(base, q1_head, q2_head, v_head, tokenizer, max_length: int, bsize: int, value_weight: float = 1.0,
                        pi_beta=None, logit_weight: Optional[float] = None):
    """ilql/gpt2/score_fn.py:22-66: sum over the last action's tokens of value_weight * (min(Q1,Q2)(s,a) - V(s))
    (+ logit_weight * log pi_beta)."""

    def score_fn(text_histories: List[TextHistory]) -> List[float]:
        import torch
        tokens, prefix_len, _ = _tokens_for_scoring(text_histories, tokenizer, max_length)
        out: List[float] = []
        for i in range(0, len(text_histories), bsize):
            tb = tokens[i:i + bsize]
            am = (tb != tokenizer.pad_token_id)
            pos = np.maximum(np.cumsum(am, axis=1) - 1, 0).astype(np.int32)
            B, T = tb.shape
            R = B * T
            ids_d = _t(tb, np.int32)
            hid, _ = base.forward(ids_d, _t(am, np.uint8), _t(pos, np.int32))
            tgt = torch.zeros(R, dtype=torch.int32, device=base.dev)
            tgt.view(B, T)[:, :-1] = ids_d[:, 1:]
            qs = []
            for head in (q1_head, q2_head):
                qo, _ = head.forward(hid, R)
                qsa = torch.empty(R, dtype=torch.float32, device=base.dev)
                ops.lse_gather(qo, head.ld_out, head.dout, tgt, R, target_logit=qsa)
                qs.append(qsa.view(B, T)[:, :-1].cpu().numpy())
            vo, _ = v_head.forward(hid, R)
            v = vo.view(B, T)[:, :-1].cpu().numpy()
            adv = (np.minimum(qs[0], qs[1]) - v) * value_weight
            if pi_beta is not None and logit_weight is not None:
                phid, _ = pi_beta.forward(ids_d, _t(am, np.uint8), _t(pos, np.int32))
                lp = torch.empty(R, dtype=torch.float32, device=base.dev)
                ops.lse_gather(pi_beta.lm_logits(phid, R), pi_beta.ld_vocab, pi_beta.vocab, tgt, R, logprob=lp)
                adv = adv + logit_weight * lp.view(B, T)[:, :-1].cpu().numpy()
            adv = adv * am[:, 1:]
            for x in range(B):
                out.append(float(adv[x][(prefix_len[i + x] - 1):].sum()))
        return out

    return score_fn
