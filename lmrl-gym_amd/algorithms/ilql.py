"""ILQL on the MI355X engine — counterpart of LLM_RL/algorithms/ilql/{base_interface,data,gpt2/interface}.py.

`ilql_loss` (numpy face, reference signature), `ILQLData` / `ILQLDataset`, and `GPT2ILQLTrain.step(...)` which restates
`GPT2ILQLTrain._step` (ilql/gpt2/interface.py:88-367): base forward (+ frozen target base), q1/q2/v MLP heads, target q
heads, Q(s,a) gathers, `v_final`, the loss, gradients w.r.t. (base, q1, q2, v), four AdamW updates and the Polyak / hard
target updates — all arithmetic in HIP kernels (sgemm_f32, train_ops, `lmrl_ilql_loss`).
"""
from __future__ import annotations

import math
from typing import Any, Dict, List, NamedTuple, Optional

import numpy as np

from .. import _lib
from .. import dist as D
from ..train import ops
from ..train.gpt2_f32 import AdamW, GPT2F32, MLPHeadF32
from .common import BlockingStrategy, Padding, Truncation, block_sequences, initialize_attn_mask_pos_ids, stats_from_sums, masked_rows
from .ppo import _t


# ----------------------------------------------------------------------------- data (ilql/data.py:10-132)
class ILQLData(NamedTuple):
    input_ids: np.ndarray            # [t]
    should_take_action: np.ndarray   # [t-1]
    rewards: np.ndarray              # [t-1]
    done: np.ndarray                 # []
    next_token_ids: Optional[np.ndarray]
    next_done: Optional[np.ndarray]

    @staticmethod
    def block(data: List["ILQLData"], blocking_strategy: BlockingStrategy, tokenizer) -> Dict[str, np.ndarray]:
        has_next = any(x.next_token_ids is not None for x in data)
        assert all(x.next_token_ids is None for x in data) or has_next
        sm = blocking_strategy._replace(max_length=blocking_strategy.max_length - 1)
        col = lambda name: [getattr(x, name) for x in data]
        return dict(
            input_ids=block_sequences(col("input_ids"), tokenizer.pad_token_id, np.int32, blocking_strategy),
            should_take_action=block_sequences(col("should_take_action"), False, np.bool_, sm),
            rewards=block_sequences(col("rewards"), 0.0, np.float32, sm),
            dones=np.asarray(col("done"), dtype=np.bool_),
            next_token_ids=block_sequences(col("next_token_ids"), tokenizer.pad_token_id, np.int32, blocking_strategy) if has_next else None,
            next_dones=np.asarray(col("next_done"), dtype=np.bool_) if has_next else None,
        )

    @classmethod
    def from_token_trajectory_chain(cls, chain) -> "ILQLData":
        """ilql/data.py:58-79: next chunk's tokens up to (excluding) its first action, for bootstrapping."""
        nxt = chain.next
        if nxt is not None:
            ia = nxt.token_trajectory.is_action
            if ia[1:].sum() > 0:
                first = int(np.argmax(ia[1:], axis=0)) + 1
                next_token_ids, next_done = nxt.token_trajectory.tokens[:first], False
            else:
                next_token_ids, next_done = nxt.token_trajectory.tokens, nxt.token_trajectory.done
        else:
            next_token_ids, next_done = None, None
        tt = chain.token_trajectory
        return cls(input_ids=tt.tokens, should_take_action=tt.is_action[1:], rewards=tt.reward[1:], done=tt.done,
                   next_token_ids=next_token_ids, next_done=next_done)


class ILQLDataset:
    def __init__(self, input_ids, should_take_action, rewards, dones, next_token_ids, next_dones):
        assert input_ids.shape[1] == should_take_action.shape[1] + 1 == rewards.shape[1] + 1
        assert input_ids.shape[0] == should_take_action.shape[0] == rewards.shape[0] == dones.shape[0]
        self.input_ids, self.should_take_action, self.rewards, self.dones = input_ids, should_take_action, rewards, dones
        self.next_token_ids, self.next_dones = next_token_ids, next_dones

    def __getitem__(self, index):
        return dict(input_ids=np.asarray(self.input_ids[index], dtype=np.int32),
                    should_take_action=np.asarray(self.should_take_action[index], dtype=np.bool_),
                    rewards=np.asarray(self.rewards[index], dtype=np.float32),
                    dones=np.asarray(self.dones[index], dtype=np.float32),
                    next_token_ids=None if self.next_token_ids is None else np.asarray(self.next_token_ids[index], dtype=np.int32),
                    next_dones=None if self.next_dones is None else np.asarray(self.next_dones[index], dtype=np.float32))

    def __len__(self):
        return self.input_ids.shape[0]

    @classmethod
    def from_ilql_data_list(cls, ilql_data_list, tokenizer, blocking_strategy) -> "ILQLDataset":
        return cls(**ILQLData.block(ilql_data_list, blocking_strategy, tokenizer))


# ----------------------------------------------------------------------------- loss (ilql/base_interface.py:29-119)
def _finalize_ilql_logs(P: np.ndarray, n: float, v_final: np.ndarray, cql_weight: float):
    f = np.float32
    s = P.sum(axis=0)
    for t in range(7):
        s[9 + 4 * t] = P[:, 9 + 4 * t].min(); s[10 + 4 * t] = P[:, 10 + 4 * t].max()
    s[38], s[39] = P[:, 38].min(), P[:, 39].max()
    if D.is_distributed():
        mn_i = [9 + 4 * t for t in range(7)] + [38]
        mx_i = [10 + 4 * t for t in range(7)] + [39]
        add_i = [k for k in range(len(s)) if k not in mn_i + mx_i]
        a, mn, mx = D.reduce_stat_partials(s[add_i], s[mn_i], s[mx_i])
        s[add_i], s[mn_i], s[mx_i] = a, mn, mx
    q1_loss, q2_loss, v_loss, c1, c2 = (s[k] / n for k in range(5))
    loss = q1_loss + q2_loss + v_loss + cql_weight * (c1 + c2)
    st = lambda o, cnt: stats_from_sums(s[o], s[o], s[o + 1], s[o + 2], s[o + 3], cnt, n)
    vf = np.asarray(v_final, dtype=np.float64)
    vf_sum, vf_sq, vf_n, vf_min, vf_max = vf.sum(), (vf * vf).sum(), float(len(vf)), vf.min(), vf.max()
    if D.is_distributed():   # v_final statistics are over the GLOBAL batch (get_tensor_stats with an all-ones mask, base_interface.py:117)
        (vf_sum, vf_sq, vf_n), (vf_min,), (vf_max,) = D.reduce_stat_partials([vf_sum, vf_sq, vf_n], [vf_min], [vf_max])
    vf_mean = vf_sum / vf_n
    vf_std = np.sqrt(max(vf_sq / vf_n - vf_mean * vf_mean, 0.0))
    logs = dict(
        losses=dict(total_loss=f(loss), q1_loss=f(q1_loss), q2_loss=f(q2_loss), v_loss=f(v_loss), q1_cql_loss=f(c1), q2_cql_loss=f(c2)),
        q1=st(7, s[5]), q2=st(11, s[5]), v=st(15, s[5]), target_q=st(19, s[5]), target_q1=st(23, s[5]), target_q2=st(27, s[5]),
        vns=st(31, s[6]),
        v_final=dict(mean=f(vf_mean), min=f(vf_min), max=f(vf_max), std=f(vf_std)),
        rewards=stats_from_sums(s[35], s[36], s[37], s[38], s[39], s[40], n),
    )
    return float(loss), logs


def ilql_loss_device(q1, q2, v, v_final, tq1, tq2, ce1, ce2, attn, sta, rewards, *, gamma, tau, cql_weight):
    """Device tensors [B, T-1] (v_final [B]); ce1/ce2 = per-token CQL cross-entropies of the two Q heads.
    Returns (loss, logs, dq1, dq2, dv, coef_ce)."""
    import torch
    L = _lib.lib()
    B, T1 = q1.shape
    dev = q1.device
    n_d = torch.zeros(1, dtype=torch.float64, device=dev)
    ops.mask_sum(sta, attn, B * T1, n_d)
    D.allreduce_sum_(n_d)     # data parallel: global token count
    ns = L.lmrl_ilql_loss_nstats()
    part = torch.empty((B, ns), dtype=torch.float64, device=dev)
    dq1, dq2, dv, coef = (torch.empty_like(q1) for _ in range(4))
    _lib.check(L.lmrl_ilql_loss(q1.data_ptr(), q2.data_ptr(), v.data_ptr(), v_final.data_ptr(), tq1.data_ptr(), tq2.data_ptr(),
                                ce1.data_ptr(), ce2.data_ptr(), attn.data_ptr(), sta.data_ptr(), rewards.data_ptr(), B, T1, float(gamma),
                                float(tau), float(cql_weight), n_d.data_ptr(), part.data_ptr(), dq1.data_ptr(), dq2.data_ptr(),
                                dv.data_ptr(), coef.data_ptr(), _lib.stream_ptr()), "lmrl_ilql_loss")
    loss, logs = _finalize_ilql_logs(part.cpu().numpy(), float(n_d.item()), v_final.cpu().numpy(), cql_weight)
    return loss, logs, dq1, dq2, dv, coef


def ilql_loss(q1, q2, v, v_final, target_q1, target_q2, q1_logits, q2_logits, token_ids, attention_mask, should_take_action, rewards, *,
              gamma, tau, cql_weight):
    """numpy face with the reference signature (q*_logits [B, T-1, V]); returns (loss, logs)."""
    import torch
    f32 = lambda x: _t(x, np.float32)
    B, T1, V = np.asarray(q1_logits).shape
    tok = _t(np.asarray(token_ids).reshape(-1), np.int32)
    ces = []
    for lg in (q1_logits, q2_logits):
        lgd = f32(np.asarray(lg).reshape(B * T1, V))
        lp = torch.empty(B * T1, dtype=torch.float32, device=lgd.device)
        ops.lse_gather(lgd, V, V, tok, B * T1, logprob=lp)
        ce = torch.empty_like(lp)
        ops.axpby(-1.0, lp, 0.0, None, ce)
        ces.append(ce.view(B, T1))
    loss, logs, *_ = ilql_loss_device(f32(q1), f32(q2), f32(v), f32(v_final), f32(target_q1), f32(target_q2), ces[0], ces[1],
                                      f32(attention_mask), _t(should_take_action, np.uint8), f32(rewards), gamma=gamma, tau=tau,
                                      cql_weight=cql_weight)
    return loss, logs


# ----------------------------------------------------------------------------- train step (ilql/gpt2/interface.py:88-367)
class GPT2ILQLTrain:
    compact_q_rows = True        # class default (GPT2ILQLInference.eval_loss builds a loss-only view without __init__)

    def __init__(self, base: GPT2F32, q1_head: MLPHeadF32, q2_head: MLPHeadF32, v_head: MLPHeadF32, pad_token_id: int,
                 loss_kwargs: Dict[str, float], target_base: Optional[GPT2F32] = None, lr: float = 3e-5, weight_decay: float = 0.0,
                 grad_accum_steps: int = 1, polyak_alpha: float = 0.005, hard_update_every: Optional[int] = None,
                 detach_q1: bool = False, detach_q2: bool = False, detach_v: bool = False, compact_q_rows: bool = True):
        """compact_q_rows: run the two Q heads (forward and backward: six [rows, d] x [d, V] products) only on the rows the loss reads —
        the `should_take_action` rows: the L2 terms of `ilql_loss` select by should_take_action alone, n and the CQL terms by
        should_take_action x attention_mask (ilql/base_interface.py:22-119), so that row set covers every Q term — instead of on all B*T
        rows; same loss, logs and gradients (rows outside the set contribute exact zeros), fewer flops in proportion to the mask density."""
        import torch
        self.compact_q_rows = compact_q_rows
        self.base, self.q1, self.q2, self.v = base, q1_head, q2_head, v_head
        self.target_base = target_base
        dev = base.dev
        # the target heads compute in the arithmetic mode of the heads they track (their hidden layer was fp32 in the bf16-matmul mode: 2 x 190 us)
        clone = lambda head: MLPHeadF32({k: t.clone() for k, t in head.p.items()}, dev, matmul="bf16" if head.mm is not None else "f32")
        self.q1_target, self.q2_target = clone(q1_head), clone(q2_head)
        self.pad, self.loss_kwargs = pad_token_id, dict(loss_kwargs)
        self.alpha, self.hard_every = polyak_alpha, hard_update_every
        self.detach_q1, self.detach_q2, self.detach_v = detach_q1, detach_q2, detach_v
        hd = lambda n: n.endswith("bias")
        self.base_opt = AdamW(base.p, lr, weight_decay=weight_decay, every_k=grad_accum_steps)
        self.q1_opt = AdamW(q1_head.p, lr, weight_decay=weight_decay, every_k=grad_accum_steps, no_decay=hd)
        self.q2_opt = AdamW(q2_head.p, lr, weight_decay=weight_decay, every_k=grad_accum_steps, no_decay=hd)
        self.v_opt = AdamW(v_head.p, lr, weight_decay=weight_decay, every_k=grad_accum_steps, no_decay=hd)
        self.last_grads = None
        self.calls = 0

    def _update_targets(self, online: Dict[str, Any], target: Dict[str, Any], step: int):
        """optax.incremental_update + optional optax.periodic_update (interface.py:327-365).  `step` is the reference's TrainState.step:
        it counts every apply_gradients call, MultiSteps micro-steps included (`self.calls`), not the applied optimizer updates."""
        hard = self.hard_every is not None and step % self.hard_every == 0
        if getattr(online, "flat", None) is not None and getattr(target, "flat", None) is not None and online.order == target.order:
            if hard:                                     # parameter arenas: the whole model in one launch
                ops.axpby(1.0, online.flat, 0.0, None, target.flat)
            else:
                ops.axpby(self.alpha, online.flat, 1.0 - self.alpha, target.flat, target.flat)
            return
        for k, tp in target.items():
            if hard:
                ops.axpby(1.0, online[k], 0.0, None, tp)
            else:
                ops.axpby(self.alpha, online[k], 1.0 - self.alpha, tp, tp)

    def step(self, input_ids, should_take_action, rewards, dones, next_token_ids=None, next_dones=None, prng_key=None,
             attention_mask=None, position_ids=None, next_tokens_attention_mask=None, next_tokens_position_ids=None, train: bool = True):
        import torch
        ids = np.asarray(input_ids, dtype=np.int32)
        am, pos = initialize_attn_mask_pos_ids(ids, self.pad, attention_mask, position_ids)
        B, T = ids.shape
        R, T1 = B * T, T - 1
        base = self.base
        dev = base.dev
        V = self.q1.dout
        ids_d, pos_d, am_d = _t(ids, np.int32), _t(pos, np.int32), _t(am, np.uint8)
        hid, cache = base.forward(ids_d, am_d, pos_d)
        thid = self.target_base.forward(ids_d, am_d, pos_d, inference=True)[0] if self.target_base is not None else hid      # never differentiated
        vo, vc = self.v.forward(hid, R)
        tgt = torch.zeros(R, dtype=torch.int32, device=dev)
        tgt.view(B, T)[:, :-1] = ids_d[:, 1:]
        # the target heads are only read at the taken token (take_along_axis, base_interface.py:57-66): one column per row, not [R, V]
        tq1sa = self.q1_target.forward_at(thid, R, tgt)
        tq2sa = self.q2_target.forward_at(thid, R, tgt)
        new = lambda: torch.empty(R, dtype=torch.float32, device=dev)
        sl = lambda x: x.view(B, T)[:, :-1].contiguous()
        sta = np.asarray(should_take_action, dtype=bool)
        # rows the Q terms of the loss read: should_take_action (host-known).  The q1 / q2 / v L2 terms and target_q select by
        # should_take_action ALONE (qv_query_indicators / sa_mask, base_interface.py:57-83); only n and the CQL terms also multiply by the
        # attention mask (:48-49, :91-95) — the loss kernel applies that second mask itself, so the compacted row set must be the superset.
        # The Q heads run on those rows only (gathered hidden states -> logits -> lse / gather; backward: dlogits -> head backward ->
        # scatter-add into d_hidden)
        q_mask = sta
        q_rows = masked_rows(q_mask, T)
        Ra = int(q_rows.size)
        compact = self.compact_q_rows and 0 < Ra < R
        if compact:
            idx = _t(q_rows, np.int32)
            hq = ops.gather_rows(hid, idx, Ra, base.d)
            tgt_q = _t(ids[:, 1:][q_mask].astype(np.int32), np.int32)
            Rq = Ra
        else:
            idx, hq, tgt_q, Rq = None, hid, tgt, R
        newq = lambda: torch.empty(Rq, dtype=torch.float32, device=dev)
        fuse_ce = self.q1.fused_ce_ok() and self.q2.fused_ce_ok()      # bf16-matmul mode: no fp32 [rows][V] logits (ops.FUSE_CE)
        if fuse_ce:
            q1o = q2o = None
            lse1, q1sa_q, lp1, q1c = self.q1.forward_ce(hq, Rq, tgt_q)
            lse2, q2sa_q, lp2, q2c = self.q2.forward_ce(hq, Rq, tgt_q)
        else:
            q1o, q1c = self.q1.forward(hq, Rq)
            q2o, q2c = self.q2.forward(hq, Rq)
            q1sa_q, lse1, lp1 = newq(), newq(), newq()
            q2sa_q, lse2, lp2 = newq(), newq(), newq()
            ld = self.q1.ld_out
            ops.lse_gather(q1o, ld, V, tgt_q, Rq, logprob=lp1, lse=lse1, target_logit=q1sa_q)
            ops.lse_gather(q2o, ld, V, tgt_q, Rq, logprob=lp2, lse=lse2, target_logit=q2sa_q)
        ce1_q, ce2_q = newq(), newq()
        ops.axpby(-1.0, lp1, 0.0, None, ce1_q)
        ops.axpby(-1.0, lp2, 0.0, None, ce2_q)
        if compact:      # back to [R] vectors for the loss kernel (zeros off the mask: those entries are multiplied by a zero mask there)
            def spread(vq):
                full_v = torch.zeros(R, dtype=torch.float32, device=dev)
                ops.scatter_rows(vq, idx, full_v, Ra, 1, False)
                return full_v
            q1sa, q2sa, ce1, ce2 = spread(q1sa_q), spread(q2sa_q), spread(ce1_q), spread(ce2_q)
        else:
            q1sa, q2sa, ce1, ce2 = q1sa_q, q2sa_q, ce1_q, ce2_q
        v_full = vo.view(B, T)
        # v_final (interface.py:253-273)
        d = np.asarray(dones, dtype=np.float32)
        if next_token_ids is not None:
            nids = np.asarray(next_token_ids, dtype=np.int32)
            nam, npos = initialize_attn_mask_pos_ids(nids, self.pad, next_tokens_attention_mask, next_tokens_position_ids)
            nhid, ncache = base.forward(_t(nids, np.int32), _t(nam, np.uint8), _t(npos, np.int32))
            last_idx = (nam.shape[1] - 1) - np.argmax(np.flip(nam, axis=1).astype(np.int32), axis=1)
            rows = torch.from_numpy((np.arange(B) * nids.shape[1] + last_idx).astype(np.int64)).to(dev)
            final_h = nhid.index_select(0, rows).contiguous()
            nv, _ = self.v.forward(final_h, B)
            v_final = torch.empty(B, dtype=torch.float32, device=dev)
            ops.axpby(1.0, nv.view(B), 0.0, None, v_final)
            v_final = v_final * _t(1.0 - np.asarray(next_dones, dtype=np.float32), np.float32)   # (1 - next_dones): host-known 0/1 mask
        else:
            last_action = (T1 - 1) - np.argmax(np.flip(sta, axis=1).astype(np.int32), axis=1) + 1
            last_token = (T - 1) - np.argmax(np.flip(am, axis=1).astype(np.int32), axis=1)
            final_idx = ((1 - d) * last_action + d * last_token).astype(np.int64)
            rows = torch.from_numpy(np.arange(B) * T + final_idx).to(dev)
            v_final = vo.view(R).index_select(0, rows) * _t(1.0 - d, np.float32)
        f32 = lambda x: _t(x, np.float32)
        loss, logs, dq1, dq2, dv, coef = ilql_loss_device(sl(q1sa), sl(q2sa), sl(v_full.reshape(R)), v_final.contiguous(), sl(tq1sa), sl(tq2sa),
                                                          sl(ce1), sl(ce2), f32(am[:, 1:]), _t(sta, np.uint8), f32(rewards), **self.loss_kwargs)
        if not train:
            return self, loss, logs
        # ---- backward
        full = lambda g: (lambda z: (z.view(B, T)[:, :-1].copy_(g), z)[1])(torch.zeros(R, dtype=torch.float32, device=dev))
        coef_r, dq1_r, dq2_r, dv_r = full(coef), full(dq1), full(dq2), full(dv)
        # d loss / d q logits: in place (fp32) or straight into the bf16 operand of the head's backward products (bf16-matmul mode)
        if compact:
            coef_q, dq1_q, dq2_q = (ops.gather_rows(x.view(R, 1), idx, Ra, 1).view(Ra) for x in (coef_r, dq1_r, dq2_r))
        else:
            coef_q, dq1_q, dq2_q = coef_r, dq1_r, dq2_r
        if fuse_ce:
            dq1o, dq1b = None, self.q1.ce_bwd_fused(q1c, lse1, tgt_q, coef_q, dq1_q, Rq)
            dq2o, dq2b = None, self.q2.ce_bwd_fused(q2c, lse2, tgt_q, coef_q, dq2_q, Rq)
        else:
            dq1o, dq1b = self.q1.ce_bwd(q1o, lse1, tgt_q, coef_q, dq1_q, Rq)
            dq2o, dq2b = self.q2.ce_bwd(q2o, lse2, tgt_q, coef_q, dq2_q, Rq)
        bgrads, g1, g2, gv = base.zero_grads(), self.q1.zero_grads(), self.q2.zero_grads(), self.v.zero_grads()
        d_hidden = torch.empty(R, base.d, dtype=torch.float32, device=dev)
        # detach_q1 / detach_q2 / detach_v (interface.py:120-139: stop_gradient on the hidden states fed to that head): the head still trains,
        # its gradient does not reach the transformer
        scratch = torch.empty_like(d_hidden) if (self.detach_q1 or self.detach_q2 or self.detach_v) else None
        d_hidden.zero_()
        if compact:
            dhq = torch.zeros(Ra, base.d, dtype=torch.float32, device=dev)          # d loss / d (gathered hidden rows), both Q heads
            scr_q = torch.empty_like(dhq) if (self.detach_q1 or self.detach_q2) else None
            self.q1.backward(q1c, dq1o, g1, dx=scr_q if self.detach_q1 else dhq, accumulate_dx=not self.detach_q1, dyb=dq1b)
            self.q2.backward(q2c, dq2o, g2, dx=scr_q if self.detach_q2 else dhq, accumulate_dx=not self.detach_q2, dyb=dq2b)
            if not (self.detach_q1 and self.detach_q2):
                ops.scatter_rows(dhq, idx, d_hidden, Ra, base.d, True)
        else:
            self.q1.backward(q1c, dq1o, g1, dx=scratch if self.detach_q1 else d_hidden, accumulate_dx=not self.detach_q1, dyb=dq1b)
            self.q2.backward(q2c, dq2o, g2, dx=scratch if self.detach_q2 else d_hidden, accumulate_dx=not self.detach_q2, dyb=dq2b)
        self.v.backward(vc, dv_r.view(R, 1), gv, dx=scratch if self.detach_v else d_hidden, accumulate_dx=not self.detach_v)
        # data parallel: the head gradients (final already) and the base gradients are all-reduced while the base backward runs — arena
        # slices go to RCCL as blocks finish (dist.GradReducer); the one data-path collective of an ILQL step (~815 MB fp32, GPT-2-small)
        red = D.GradReducer()
        late = red.early([g1, g2, gv])                 # the heads' gradients are final: reduced under the base backward, not after it
        base.backward(cache, d_hidden, bgrads, on_final=red.ready(bgrads))
        self.last_grads = (bgrads, g1, g2, gv)
        red.finish(late)
        self.calls += 1                                 # TrainState.step of the reference: one per apply_gradients call
        # targets move only when MultiSteps.mini_step == 0 (interface.py:343-347); their Polyak step rides in the AdamW sweep over the same arena
        # (one pass over parameters + optimizer state + target instead of an update pass and a second pass that re-reads the parameters)
        pairs = [(self.base_opt, base.p, self.target_base.p if self.target_base is not None else None, bgrads),
                 (self.q1_opt, self.q1.p, self.q1_target.p, g1), (self.q2_opt, self.q2.p, self.q2_target.p, g2)]
        hard = self.hard_every is not None and self.calls % self.hard_every == 0
        upd = False
        for opt, online, target, grads in pairs:
            upd = opt.apply(grads, polyak=None if (target is None or hard) else (target, self.alpha))
            if upd and target is not None and not opt.polyak_fused:
                self._update_targets(online, target, self.calls)
        self.v_opt.apply(gv)
        return self, loss, logs


# ----------------------------------------------------------------------------- inference face
class ValueRLForwardOutput(NamedTuple):
    """value_rl_base/base_interface.py:20-24 (base_raw_output reduced to the LM logits)."""
    base_logits: np.ndarray            # [B, T, V]
    q1: np.ndarray                     # [B, T, V]
    q2: Optional[np.ndarray]
    v: Optional[np.ndarray]            # [B, T]


class GPT2ILQLInference:
    """Counterpart of `ILQLInference` / `ValueRLInference` (ilql/base_interface.py:235-439, value_rl_base/base_interface.py:26-181)
    on the fp32 train kernels: `forward`, `forward_from_str`, `eval_loss`; generation is `policies.GPT2ValuePolicy`
    (`generate_from_str` below wraps it)."""

    def __init__(self, base: GPT2F32, q1_head: MLPHeadF32, q2_head: Optional[MLPHeadF32], v_head: Optional[MLPHeadF32], pad_token_id: int,
                 tokenizer=None, loss_kwargs: Optional[Dict[str, float]] = None, target_base: Optional[GPT2F32] = None,
                 q1_target_head: Optional[MLPHeadF32] = None, q2_target_head: Optional[MLPHeadF32] = None, policy=None):
        self.base, self.q1, self.q2, self.v, self.pad, self.tokenizer = base, q1_head, q2_head, v_head, pad_token_id, tokenizer
        self.loss_kwargs, self.target_base = dict(loss_kwargs or {}), target_base
        self.q1_target, self.q2_target, self.policy = q1_target_head, q2_target_head, policy

    def forward(self, input_ids, attention_mask=None, position_ids=None) -> ValueRLForwardOutput:
        ids = np.asarray(input_ids, dtype=np.int32)
        am, pos = initialize_attn_mask_pos_ids(ids, self.pad, attention_mask, position_ids)
        B, T = ids.shape
        R = B * T
        hid, _ = self.base.forward(_t(ids, np.int32), _t(am, np.uint8), _t(pos, np.int32))
        logits = self.base.lm_logits(hid, R).view(B, T, -1)[:, :, :self.base.vocab].cpu().numpy()
        q1 = self.q1.forward(hid, R)[0].view(B, T, -1)[:, :, :self.q1.dout].cpu().numpy()
        q2 = self.q2.forward(hid, R)[0].view(B, T, -1)[:, :, :self.q2.dout].cpu().numpy() if self.q2 is not None else None
        v = self.v.forward(hid, R)[0].view(B, T).cpu().numpy() if self.v is not None else None
        return ValueRLForwardOutput(logits, q1, q2, v)

    def forward_from_str(self, input_strs: List[str], blocking_strategy: BlockingStrategy = BlockingStrategy(Padding.RIGHT, Truncation.RIGHT, None),
                         token_process=None) -> ValueRLForwardOutput:
        tp = token_process or (lambda x: x)
        return self.forward(block_sequences([tp(list(self.tokenizer.encode(s))) for s in input_strs], self.pad, np.int32, blocking_strategy))

    def eval_loss(self, input_ids, should_take_action, rewards, dones, next_token_ids=None, next_dones=None, attention_mask=None,
                  position_ids=None, next_tokens_attention_mask=None, next_tokens_position_ids=None, prng_key=None, train: bool = False):
        """ilql/base_interface.py:383-439: loss and log dict of the train step without an update."""
        ev = GPT2ILQLTrain.__new__(GPT2ILQLTrain)          # loss-only view on the same weights: no optimizer state is created
        ev.base, ev.q1, ev.q2, ev.v, ev.pad, ev.loss_kwargs = self.base, self.q1, self.q2, self.v, self.pad, self.loss_kwargs
        ev.target_base = self.target_base
        ev.q1_target, ev.q2_target = self.q1_target or self.q1, self.q2_target or self.q2
        _, loss, logs = ev.step(input_ids, should_take_action, rewards, dones, next_token_ids=next_token_ids, next_dones=next_dones,
                                attention_mask=attention_mask, position_ids=position_ids, next_tokens_attention_mask=next_tokens_attention_mask,
                                next_tokens_position_ids=next_tokens_position_ids, train=False)
        return loss, logs

    def generate_from_str(self, input_strs: List[str]) -> List[str]:
        """Completions of `input_strs` under the ILQL-perturbed policy given at construction (a `policies.GPT2ValuePolicy`)."""
        from ..environment import Text
        assert self.policy is not None, "pass policy=GPT2ValuePolicy(...) to generate"
        out = self.policy.act([(Text(s, False),) for s in input_strs], [False] * len(input_strs))
        return [h[-1].text for h in out]
