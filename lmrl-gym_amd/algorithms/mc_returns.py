"""MC-returns baseline pieces — counterpart of LLM_RL/algorithms/mc_returns/{data,base_interface}.py:
`get_rtg` (reward-to-go over action tokens, `lmrl_rtg`), `MCData`, `mc_loss` (`lmrl_mc_loss`)."""
from __future__ import annotations

from typing import Dict, List, NamedTuple

import numpy as np

from .. import _lib
from .. import dist as D
from ..train import ops
from .common import BlockingStrategy, block_sequences, stats_from_sums
from .ppo import _t


def get_rtg(rewards: np.ndarray, gamma: float) -> np.ndarray:
    """mc_returns/data.py:10-14 for a 1-d array of action rewards: rtg_i = sum_{j>=i} gamma^(j-i) r_j."""
    import torch
    r = np.asarray(rewards, dtype=np.float32).reshape(1, -1)
    n = r.shape[1]
    if n == 0:
        return np.zeros((0,), dtype=np.float32)
    rd = _t(r, np.float32)
    sta = torch.ones((1, n), dtype=torch.uint8, device=rd.device)
    out = torch.empty((1, n), dtype=torch.float32, device=rd.device)
    _lib.check(_lib.lib().lmrl_rtg(rd.data_ptr(), sta.data_ptr(), None, out.data_ptr(), 1, n, float(gamma), _lib.stream_ptr()))
    return out.cpu().numpy()[0]


def rtg_from_chains(rewards: np.ndarray, should_take_action: np.ndarray, lens: np.ndarray, gamma: float) -> np.ndarray:
    """Batched form: rewards / should_take_action [B, L] (whole chains concatenated per row) -> rtg scattered to positions."""
    import torch
    B, L = rewards.shape
    rd, sd, ld = _t(rewards, np.float32), _t(should_take_action, np.uint8), _t(lens, np.int32)
    out = torch.empty((B, L), dtype=torch.float32, device=rd.device)
    _lib.check(_lib.lib().lmrl_rtg(rd.data_ptr(), sd.data_ptr(), ld.data_ptr(), out.data_ptr(), B, L, float(gamma), _lib.stream_ptr()))
    return out.cpu().numpy()


class MCData(NamedTuple):
    input_ids: np.ndarray           # [t]
    should_take_action: np.ndarray  # [t-1]
    returns: np.ndarray             # [t-1]

    @staticmethod
    def block(data: List["MCData"], blocking_strategy: BlockingStrategy, tokenizer) -> Dict[str, np.ndarray]:
        sm = blocking_strategy._replace(max_length=blocking_strategy.max_length - 1)
        col = lambda name: [getattr(x, name) for x in data]
        return dict(input_ids=block_sequences(col("input_ids"), tokenizer.pad_token_id, np.int32, blocking_strategy),
                    should_take_action=block_sequences(col("should_take_action"), False, np.bool_, sm),
                    returns=block_sequences(col("returns"), 0.0, np.float32, sm))

    @classmethod
    def from_token_trajectory_chain(cls, token_trajectory_chain, gamma: float) -> "MCData":
        """mc_returns/data.py:49-74: reward-to-go over the action tokens of the WHOLE chain, written onto the first chunk."""
        filt = []
        for tt in token_trajectory_chain.to_list():
            sta = tt.is_action[1:]
            filt.append(tt.reward[1:][sta])
        rtgs = get_rtg(np.concatenate(filt, axis=0), gamma)
        sta0 = token_trajectory_chain.token_trajectory.is_action[1:]
        returns = np.zeros(sta0.shape, dtype=np.float32)
        returns[sta0] = rtgs[: sta0.sum()]
        return cls(input_ids=token_trajectory_chain.token_trajectory.tokens, should_take_action=sta0, returns=returns)


def mc_loss_device(q, ce, attn, sta, returns, *, cql_weight):
    """Device tensors [B, T-1]; returns (loss, logs, dq, coef_ce)."""
    import torch
    L = _lib.lib()
    n_el, dev = q.numel(), q.device
    n_d = torch.zeros(1, dtype=torch.float64, device=dev)
    ops.mask_sum(sta, attn, n_el, n_d)
    D.allreduce_sum_(n_d)     # data parallel: divide by the GLOBAL token count, as PPO / ILQL do (dist.py convention)
    nb, ns = L.lmrl_mc_loss_blocks(n_el), L.lmrl_mc_loss_nstats()
    part = torch.empty((nb, ns), dtype=torch.float64, device=dev)
    dq, coef = torch.empty_like(q), torch.empty_like(q)
    _lib.check(L.lmrl_mc_loss(q.data_ptr(), ce.data_ptr(), attn.data_ptr(), sta.data_ptr(), returns.data_ptr(), n_el, float(cql_weight),
                              n_d.data_ptr(), part.data_ptr(), dq.data_ptr(), coef.data_ptr(), _lib.stream_ptr()), "lmrl_mc_loss")
    P = part.cpu().numpy()
    s = P.sum(axis=0)
    s[5], s[6], s[9], s[10] = P[:, 5].min(), P[:, 6].max(), P[:, 9].min(), P[:, 10].max()
    if D.is_distributed():    # logged sums / mins / maxes of all ranks -> the log dict equals the single-process one
        mn_i, mx_i = [5, 9], [6, 10]
        add_i = [k for k in range(len(s)) if k not in mn_i + mx_i]
        a, mn, mx = D.reduce_stat_partials(s[add_i], s[mn_i], s[mx_i])
        s[add_i], s[mn_i], s[mx_i] = a, mn, mx
    n = float(n_d.item())
    f = np.float32
    q_loss, cql = s[0] / n, s[1] / n
    loss = q_loss + cql_weight * cql
    logs = dict(losses=dict(total_loss=f(loss), q_loss=f(q_loss), q_cql_loss=f(cql)),
                q=stats_from_sums(s[3], s[3], s[4], s[5], s[6], s[2], n), returns=stats_from_sums(s[7], s[7], s[8], s[9], s[10], s[2], n))
    return float(loss), logs, dq, coef


def mc_loss(q, q_logits, token_ids, attention_mask, should_take_action, returns, *, cql_weight):
    """numpy face with the reference signature (mc_returns/base_interface.py:19-60); returns (loss, logs)."""
    import torch
    B, T1, V = np.asarray(q_logits).shape
    lg = _t(np.asarray(q_logits).reshape(B * T1, V), np.float32)
    tok = _t(np.asarray(token_ids).reshape(-1), np.int32)
    lp = torch.empty(B * T1, dtype=torch.float32, device=lg.device)
    ops.lse_gather(lg, V, V, tok, B * T1, logprob=lp)
    ce = torch.empty_like(lp)
    ops.axpby(-1.0, lp, 0.0, None, ce)
    f32 = lambda x: _t(x, np.float32)
    loss, logs, _, _ = mc_loss_device(f32(q), ce.view(B, T1), f32(attention_mask), _t(should_take_action, np.uint8), f32(returns),
                                      cql_weight=cql_weight)
    return loss, logs


class GPT2MCTrain:
    """fp32 MC-returns trainer: GPT-2 base + Q `MLPHead` (d -> d -> V), loss = `mc_loss`
    (LLM_RL/algorithms/mc_returns/gpt2/interface.py:57-170; step signature of mc_returns/base_interface.py:89-137).
    `step` returns `(self, loss, logs)` and updates the trainer in place."""

    def __init__(self, base, q_head, pad_token_id: int, loss_kwargs, lr: float = 3e-5, weight_decay: float = 0.0,
                 grad_accum_steps: int = 1, detach_q: bool = False):
        from ..train.gpt2_f32 import AdamW
        self.base, self.q_head, self.pad, self.loss_kwargs, self.detach_q = base, q_head, pad_token_id, dict(loss_kwargs), detach_q
        self.base_opt = AdamW(base.p, lr, weight_decay=weight_decay, every_k=grad_accum_steps)
        self.q_opt = AdamW(q_head.p, lr, weight_decay=weight_decay, every_k=grad_accum_steps, no_decay=lambda n: n.endswith("bias"))
        self.last_grads = None

    compact_q_rows = True

    def step(self, input_ids, should_take_action, returns, prng_key=None, attention_mask=None, position_ids=None, train: bool = True):
        import torch
        from .. import dist as D
        from .common import initialize_attn_mask_pos_ids
        ids = np.asarray(input_ids, dtype=np.int32)
        am, pos = initialize_attn_mask_pos_ids(ids, self.pad, attention_mask, position_ids)
        B, T = ids.shape
        R, base, dev, V = B * T, self.base, self.base.dev, self.q_head.dout
        ids_d = _t(ids, np.int32)
        hid, cache = base.forward(ids_d, _t(am, np.uint8), _t(pos, np.int32))
        # the Q head runs on the rows the loss reads: should_take_action (the L2 term selects by it alone — q_query_indicators / a_mask,
        # mc_returns/base_interface.py:32-41; the CQL term and n also multiply by the attention mask, which the loss kernel applies) — see
        # GPT2ILQLTrain.compact_q_rows
        from .common import masked_rows
        q_mask = np.asarray(should_take_action, dtype=bool)
        rows_h = masked_rows(q_mask, T)
        Ra = int(rows_h.size)
        compact = self.compact_q_rows and 0 < Ra < R
        if compact:
            idx = _t(rows_h, np.int32)
            hq, Rq, tgt = ops.gather_rows(hid, idx, Ra, base.d), Ra, _t(ids[:, 1:][q_mask].astype(np.int32), np.int32)
        else:
            idx, hq, Rq = None, hid, R
            tgt = torch.zeros(R, dtype=torch.int32, device=dev)
            tgt.view(B, T)[:, :-1] = ids_d[:, 1:]
        newq = lambda: torch.empty(Rq, dtype=torch.float32, device=dev)
        fuse_ce = self.q_head.fused_ce_ok()              # bf16-matmul mode: no fp32 [rows][V] logits (ops.FUSE_CE)
        if fuse_ce:
            qo = None
            lse, qsa_q, lp, qc = self.q_head.forward_ce(hq, Rq, tgt)
            ce_q = newq()
        else:
            qo, qc = self.q_head.forward(hq, Rq)
            qsa_q, lse, lp, ce_q = newq(), newq(), newq(), newq()
            ld = self.q_head.ld_out
            ops.lse_gather(qo, ld, V, tgt, Rq, logprob=lp, lse=lse, target_logit=qsa_q)
        ops.axpby(-1.0, lp, 0.0, None, ce_q)
        if compact:
            qsa, ce = torch.zeros(R, dtype=torch.float32, device=dev), torch.zeros(R, dtype=torch.float32, device=dev)
            ops.scatter_rows(qsa_q, idx, qsa, Ra, 1, False)
            ops.scatter_rows(ce_q, idx, ce, Ra, 1, False)
        else:
            qsa, ce = qsa_q, ce_q
        sl = lambda x: x.view(B, T)[:, :-1].contiguous()
        f32 = lambda x: _t(x, np.float32)
        loss, logs, dq, coef = mc_loss_device(sl(qsa), sl(ce), f32(am[:, 1:]), _t(should_take_action, np.uint8), f32(returns), **self.loss_kwargs)
        if not train:
            return self, loss, logs
        full = lambda g: (lambda z: (z.view(B, T)[:, :-1].copy_(g), z)[1])(torch.zeros(R, dtype=torch.float32, device=dev))
        coef_r, dq_r = full(coef), full(dq)
        if compact:
            coef_r, dq_r = (ops.gather_rows(x.view(R, 1), idx, Ra, 1).view(Ra) for x in (coef_r, dq_r))
        if fuse_ce:
            dqo, dqb = None, self.q_head.ce_bwd_fused(qc, lse, tgt, coef_r, dq_r, Rq)
        else:
            dqo, dqb = self.q_head.ce_bwd(qo, lse, tgt, coef_r, dq_r, Rq)          # d loss / d q logits
        bgrads, qgrads = base.zero_grads(), self.q_head.zero_grads()
        if compact:
            dhq = torch.empty(Ra, base.d, dtype=torch.float32, device=dev)
            self.q_head.backward(qc, dqo, qgrads, dx=dhq, accumulate_dx=False, dyb=dqb)
            d_hidden = torch.zeros(R, base.d, dtype=torch.float32, device=dev)
            if not self.detach_q:
                ops.scatter_rows(dhq, idx, d_hidden, Ra, base.d, False)
        else:
            d_hidden = torch.empty(R, base.d, dtype=torch.float32, device=dev)
            self.q_head.backward(qc, dqo, qgrads, dx=d_hidden, accumulate_dx=False, dyb=dqb)
            if self.detach_q:
                d_hidden.zero_()
        red = D.GradReducer()                        # gradient all-reduce overlapped with the base backward (data parallel)
        late = red.early([qgrads])                   # the head's gradients are final already
        base.backward(cache, d_hidden, bgrads, on_final=red.ready(bgrads))
        self.last_grads = (bgrads, qgrads)
        red.finish(late)
        self.base_opt.apply(bgrads)
        self.q_opt.apply(qgrads)
        return self, loss, logs
