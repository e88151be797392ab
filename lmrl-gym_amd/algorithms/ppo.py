"""PPO on the MI355X engine — counterpart of LLM_RL/algorithms/ppo/{base_interface,data,gpt2/interface}.py.

Same names and call shapes as the reference where one exists: `AdaptiveKLController`, `FixedKLController`,
`ppo_loss_fn`, `get_action_state_next_state_idxs`, `whiten`, `get_advantages_and_returns`, `PPOData`, `PPODataset`,
`GPT2PPOTrain.step(...) -> (trainer, loss, logs)`.  Arrays in / out are numpy (as in the reference's host code);
the arithmetic runs in the HIP kernels (`lmrl_gae`, `lmrl_whiten_*`, `lmrl_ppo_loss`, sgemm/train_ops for the model).
"""
from __future__ import annotations

import ctypes
from typing import Any, Dict, List, NamedTuple, Optional, Tuple

import numpy as np

from .. import _lib
from .. import dist as D
from ..train import ops
from ..train.gpt2_f32 import AdamW, GPT2F32, LinearHeadF32
from .common import BlockingStrategy, block_sequences, initialize_attn_mask_pos_ids, masked_rows, stats_from_sums


# ----------------------------------------------------------------------------- KL controllers (base_interface.py:38-69)
class AdaptiveKLController:
    def __init__(self, init_kl_coef: float, target: float, horizon: int):
        self.value, self.target, self.horizon = init_kl_coef, target, horizon

    def update(self, current: float, n_steps: int):
        proportional_error = np.clip(current / self.target - 1, -0.2, 0.2)
        self.value *= 1 + proportional_error * n_steps / self.horizon


class FixedKLController:
    def __init__(self, kl_coef):
        self.value = kl_coef

    def update(self, current: float, n_steps: int):
        pass


# ----------------------------------------------------------------------------- device helpers
def _dev():
    return _lib.require_gpu()


_TORCH_OF = {np.float32: "float32", np.int32: "int32", np.uint8: "uint8", np.int64: "int64", np.float64: "float64"}


def _on_device(x) -> bool:
    import torch
    return isinstance(x, torch.Tensor) and x.is_cuda


def _t(x, dtype):
    """numpy (the reference's host batches) -> device tensor; a tensor already in HBM (a `DevicePPODataset` batch) passes through."""
    import torch
    if _on_device(x):
        want = getattr(torch, _TORCH_OF[dtype])
        return (x if x.dtype == want else x.to(want)).contiguous()
    return torch.from_numpy(np.ascontiguousarray(x, dtype=dtype)).to(_dev())


def get_action_state_next_state_idxs(should_take_action: np.ndarray):
    """base_interface.py:230-243 (host index arithmetic; the device GAE kernel derives the same pairing itself)."""
    sta = np.asarray(should_take_action, dtype=bool)
    action_idxs = np.where(sta)[0]
    is_next = sta.copy()
    if sta.size:
        is_next[np.argmax(is_next.astype(np.int32))] = False
    is_next = np.concatenate((is_next, np.array([sta.sum() > 0])))
    return action_idxs, action_idxs.copy(), np.where(is_next)[0]


def whiten(xs: np.ndarray, shift_mean: bool = True) -> np.ndarray:
    """base_interface.py:245-251 over ALL elements of xs."""
    import torch
    x = _t(np.asarray(xs).reshape(-1), np.float32)
    mom = torch.zeros(3, dtype=torch.float64, device=x.device)
    y = torch.empty_like(x)
    L = _lib.lib()
    _lib.check(L.lmrl_whiten_moments(x.data_ptr(), None, mom.data_ptr(), x.numel(), _lib.stream_ptr()))
    _lib.check(L.lmrl_whiten_apply(x.data_ptr(), None, mom.data_ptr(), y.data_ptr(), x.numel(), int(shift_mean), _lib.stream_ptr()))
    return y.cpu().numpy().reshape(np.asarray(xs).shape)


def get_advantages_and_returns(state_values, next_state_values, action_rewards, *, gamma, lam, use_whitening: bool = True):
    """base_interface.py:253-293 on already-compacted [b, n] arrays (every column is an action)."""
    import torch
    v, nv, r = (np.asarray(a, dtype=np.float32) for a in (state_values, next_state_values, action_rewards))
    b, n = v.shape
    # lmrl_gae consumes per-chain value rows with the next-state value of the LAST action in the bootstrap slot;
    # for pre-compacted inputs next_state_values[:, :-1] == state_values[:, 1:] is not guaranteed, so run it per column
    # pairing explicitly: values row = [v_0 .. v_{n-1}, nv_{n-1}] only matches when chains are contiguous.  The general
    # (compacted) form is evaluated as a single-chain scan with explicit deltas instead.
    delta = r + np.float32(gamma) * nv - v
    vals = np.zeros((b, n + 1), dtype=np.float32)          # values == 0 -> adv recurrence on delta alone
    dv, dr = _t(vals, np.float32), _t(delta, np.float32)
    sta = torch.ones((b, n), dtype=torch.uint8, device=dv.device)
    adv = torch.empty((b, n), dtype=torch.float32, device=dv.device)
    ret = torch.empty_like(adv)
    # with V == 0 the kernel's delta_t is r_t and its recurrence A_t = r_t + (gamma*lam) A_{t+1} when called with gamma'=gamma*lam, lam'=1
    _lib.check(_lib.lib().lmrl_gae(dv.data_ptr(), dr.data_ptr(), sta.data_ptr(), None, adv.data_ptr(), ret.data_ptr(), b, n,
                                   float(gamma * lam), 1.0, _lib.stream_ptr()))
    advantages = adv.cpu().numpy()
    returns = advantages + v
    if use_whitening:
        advantages = whiten(advantages)
    return advantages, returns


def gae_from_chains(values: np.ndarray, rewards: np.ndarray, should_take_action: np.ndarray, lens: np.ndarray, gamma: float, lam: float):
    """Chain form used by the PPO data pipeline (base_interface.py:586-606, 635-645): values [B, L+1] with the bootstrap
    slot at index len, rewards / should_take_action [B, L]; returns advantages / returns scattered to token positions."""
    import torch
    B, L = rewards.shape
    dv, dr = _t(values, np.float32), _t(rewards, np.float32)
    ds, dl = _t(should_take_action, np.uint8), _t(lens, np.int32)
    adv = torch.empty((B, L), dtype=torch.float32, device=dv.device)
    ret = torch.empty_like(adv)
    _lib.check(_lib.lib().lmrl_gae(dv.data_ptr(), dr.data_ptr(), ds.data_ptr(), dl.data_ptr(), adv.data_ptr(), ret.data_ptr(), B, L,
                                   float(gamma), float(lam), _lib.stream_ptr()))
    return adv.cpu().numpy(), ret.cpu().numpy()


# ----------------------------------------------------------------------------- loss (base_interface.py:72-142)
def ppo_loss_device(attn, logprobs, values, sta, old_logprobs, old_values, old_advantages, old_returns, *, cliprange_value,
                    cliprange, value_loss_coef):
    """All inputs are device tensors of identical shape ([B, T-1]; sta uint8, the rest float32).
    Returns (loss float, logs dict, d_logprobs, d_values)."""
    import torch
    L = _lib.lib()
    n_el = logprobs.numel()
    dev = logprobs.device
    n_d = torch.zeros(1, dtype=torch.float64, device=dev)
    ops.mask_sum(sta, attn, n_el, n_d)
    D.allreduce_sum_(n_d)     # data parallel: divide by the GLOBAL token count (ppo/base_interface.py:92 under a dp-sharded batch)
    nb, ns = L.lmrl_ppo_loss_blocks(n_el), L.lmrl_ppo_loss_nstats()
    part = torch.empty((nb, ns), dtype=torch.float64, device=dev)
    dlp, dv = torch.empty_like(logprobs), torch.empty_like(values)
    _lib.check(L.lmrl_ppo_loss(attn.data_ptr(), logprobs.data_ptr(), values.data_ptr(), sta.data_ptr(), old_logprobs.data_ptr(),
                               old_values.data_ptr(), old_advantages.data_ptr(), old_returns.data_ptr(), n_el, float(cliprange_value),
                               float(cliprange), float(value_loss_coef), n_d.data_ptr(), part.data_ptr(), dlp.data_ptr(), dv.data_ptr(),
                               _lib.stream_ptr()), "lmrl_ppo_loss")
    P = part.cpu().numpy()
    s = P.sum(axis=0)
    for k in (12, 17, 22):
        s[k] = P[:, k].min()
    for k in (13, 18, 23):
        s[k] = P[:, k].max()
    if D.is_distributed():
        mn_i, mx_i = [12, 17, 22], [13, 18, 23]
        add_i = [k for k in range(len(s)) if k not in mn_i + mx_i]
        a, mn, mx = D.reduce_stat_partials(s[add_i], s[mn_i], s[mx_i])
        s[add_i], s[mn_i], s[mx_i] = a, mn, mx
        n_el = n_el * D.world()[1]
    n = s[0]
    f = np.float32
    vf_loss, pg_loss = 0.5 * s[1] / n, s[4] / n
    loss = pg_loss + value_loss_coef * vf_loss
    st = lambda o: stats_from_sums(s[o], s[o + 1], s[o + 2], s[o + 3], s[o + 4], s[8], n)
    logs = dict(
        losses=dict(total_loss=f(loss), policy_loss=f(pg_loss), value_loss=f(vf_loss)),
        values=dict(st(9), values_error=f(s[6] / n), clipfrac=f(s[2] / n)),
        old_values=st(14), returns=st(19),
        policy=dict(approx_kl=f(s[3] / n), clipfrac=f(s[5] / n)),
        ratio=f(s[7] / n), padding_percentage=f(n / n_el),
    )
    return float(loss), logs, dlp, dv


def ppo_loss_fn(attention_mask, logprobs, values, should_take_action, old_logprobs, old_values, old_advantages, old_returns, *,
                cliprange_value, cliprange, value_loss_coef):
    """numpy-in / numpy-out face with the reference signature; returns (loss, logs)."""
    a = lambda x: _t(x, np.float32)
    loss, logs, _, _ = ppo_loss_device(a(attention_mask), a(logprobs), a(values), _t(should_take_action, np.uint8), a(old_logprobs),
                                       a(old_values), a(old_advantages), a(old_returns), cliprange_value=cliprange_value,
                                       cliprange=cliprange, value_loss_coef=value_loss_coef)
    return loss, logs


# ----------------------------------------------------------------------------- data (ppo/data.py:9-114)
def _masked_lm_term(policy, pad, input_ids, attention_mask, position_ids, training_mask, grads, grad_scale):
    """The BC auxiliary of PPO: JaxSeq `loss_fn_mask` as wired in llm_rl_scripts/wordle/ppo/train_ppo_gpt2.py —
    sum(CE * training_mask[:, 1:]) / sum(training_mask[:, 1:]) on a second batch (third-party arithmetic: parity unpinned)."""
    from .bc import masked_ce_forward_backward
    ids = np.asarray(input_ids, dtype=np.int32)
    am, pos = initialize_attn_mask_pos_ids(ids, pad, attention_mask, position_ids)
    tm = np.asarray(training_mask if training_mask is not None else am, dtype=np.float32)[:, 1:]
    return masked_ce_forward_backward(policy, ids, am, pos, tm, float(tm.sum()), grads, grad_scale)


class PPOData(NamedTuple):
    input_ids: np.ndarray           # [t]
    should_take_action: np.ndarray  # [t-1]
    old_logprobs: np.ndarray        # [t-1]
    old_values: np.ndarray          # [t-1]
    old_advantages: np.ndarray      # [t-1]
    old_returns: np.ndarray         # [t-1]

    @staticmethod
    def block(data: List["PPOData"], blocking_strategy: BlockingStrategy, tokenizer) -> Dict[str, np.ndarray]:
        sm = blocking_strategy._replace(max_length=blocking_strategy.max_length - 1)
        col = lambda name: [getattr(x, name) for x in data]
        return dict(
            input_ids=block_sequences(col("input_ids"), tokenizer.pad_token_id, np.int32, blocking_strategy),
            should_take_action=block_sequences(col("should_take_action"), False, np.bool_, sm),
            old_logprobs=block_sequences(col("old_logprobs"), 0.0, np.float32, sm),
            old_values=block_sequences(col("old_values"), 0.0, np.float32, sm),
            old_advantages=block_sequences(col("old_advantages"), 0.0, np.float32, sm),
            old_returns=block_sequences(col("old_returns"), 0.0, np.float32, sm),
        )


class PPODataset:
    def __init__(self, input_ids, should_take_action, old_logprobs, old_values, old_advantages, old_returns):
        for a in (should_take_action, old_logprobs, old_values, old_advantages, old_returns):
            assert input_ids.shape[1] == a.shape[1] + 1 and input_ids.shape[0] == a.shape[0]
        self.input_ids, self.should_take_action = input_ids, should_take_action
        self.old_logprobs, self.old_values = old_logprobs, old_values
        self.old_advantages, self.old_returns = old_advantages, old_returns

    def __getitem__(self, index):
        return dict(input_ids=np.asarray(self.input_ids[index], dtype=np.int32),
                    should_take_action=np.asarray(self.should_take_action[index], dtype=np.bool_),
                    old_logprobs=np.asarray(self.old_logprobs[index], dtype=np.float32),
                    old_values=np.asarray(self.old_values[index], dtype=np.float32),
                    old_advantages=np.asarray(self.old_advantages[index], dtype=np.float32),
                    old_returns=np.asarray(self.old_returns[index], dtype=np.float32))

    def __len__(self):
        return self.input_ids.shape[0]

    @classmethod
    def from_ppo_data_list(cls, ppo_data_list: List[PPOData], tokenizer, blocking_strategy: BlockingStrategy) -> "PPODataset":
        return cls(**PPOData.block(ppo_data_list, blocking_strategy, tokenizer))


# ----------------------------------------------------------------------------- train step (ppo/gpt2/interface.py:72-211)
class GPT2PPOTrain:
    """fp32 PPO trainer: GPT-2 policy + LinearHead value head, two AdamW states.

    `compact_rows` (default on): the tied LM head (forward and both backward products, [rows, d] x [d, V]) runs only on the rows whose
    log-probability the loss reads — `should_take_action x attention_mask[:, 1:]` masks every policy term of `ppo_loss_fn`
    (ppo/base_interface.py:72-142) — same loss / logs / gradients, head flops in proportion to the mask density.

    `step(...)` has the reference signature (base_interface.py:172-228) and returns `(self, loss, logs)`; the trainer is
    updated in place (the reference donates the old buffers)."""

    def __init__(self, policy: GPT2F32, value_head: LinearHeadF32, pad_token_id: int, loss_kwargs: Dict[str, float],
                 lr: float = 1e-5, weight_decay: float = 0.0, grad_accum_steps: int = 1, bc_loss_weight: float = 0.0,
                 bc_non_action_weight: float = 0.0):
        self.policy, self.value_head, self.pad = policy, value_head, pad_token_id
        self.loss_kwargs = dict(loss_kwargs)
        self.policy_opt = AdamW(policy.p, lr, weight_decay=weight_decay, every_k=grad_accum_steps)
        self.head_opt = AdamW(value_head.p, lr, weight_decay=weight_decay, every_k=grad_accum_steps, no_decay=lambda n: n == "bias")
        self.bc_loss_weight = bc_loss_weight
        self.last_grads = None

    compact_rows = True

    def _bc_term(self, ids, am, pos, tmask, grads):
        return _masked_lm_term(self.policy, self.pad, ids, am, pos, tmask, grads, self.bc_loss_weight)

    def step(self, input_ids, should_take_action, old_logprobs, old_values, old_advantages, old_returns, prng_key=None,
             attention_mask=None, position_ids=None, bc_data_input_ids=None, bc_data_input_attention_mask=None,
             bc_data_input_position_ids=None, bc_data_input_training_mask=None, train: bool = True):
        import torch
        pol, head = self.policy, self.value_head
        dev = pol.dev
        f32 = lambda x: _t(x, np.float32)
        if _on_device(input_ids):
            # a batch that never left the device (`ppo_device.DevicePPODataset.batch`): masks, positions, the masked row list and its
            # next-token targets come from csrc/ppo_data.hip; the host learns one integer (the row count that sizes the LM-head product)
            from .ppo_device import mask_pos_device, masked_rows_device
            ids_d = _t(input_ids, np.int32)
            B, T = ids_d.shape
            R = B * T
            if attention_mask is None and position_ids is None:
                am_d, pos_d, attn_s = mask_pos_device(ids_d, self.pad, shifted=True)
            else:
                am0, pos0 = mask_pos_device(ids_d, self.pad)
                am_d = _t(attention_mask, np.uint8) if attention_mask is not None else am0
                pos_d = _t(position_ids, np.int32) if position_ids is not None else pos0
                attn_s = f32(am_d[:, 1:])
            sta_d = _t(should_take_action, np.uint8)
            idx, tgt_rows, Ra = masked_rows_device(sta_d, am_d, ids_d, T)
        else:
            ids = np.asarray(input_ids, dtype=np.int32)
            am, pos = initialize_attn_mask_pos_ids(ids, self.pad, attention_mask, position_ids)
            B, T = ids.shape
            R = B * T
            ids_d, pos_d, am_d = _t(ids, np.int32), _t(pos, np.int32), _t(am, np.uint8)
            # logprobs[b, t] = log p(ids[b, t+1] | ids[b, :t+1]) for t < T-1 : row r = b*T + t, target ids[b, t+1] — needed on the masked rows only
            p_mask = np.asarray(should_take_action, dtype=bool) & (np.asarray(am)[:, 1:] != 0)
            rows_h = masked_rows(p_mask, T)
            Ra = int(rows_h.size)
            idx, tgt_rows = _t(rows_h, np.int32), _t(ids[:, 1:][p_mask].astype(np.int32), np.int32)
            attn_s, sta_d = f32(am[:, 1:]), _t(should_take_action, np.uint8)
        hid, cache = pol.forward(ids_d, am_d, pos_d)
        values_full, hcache = head.forward(hid, R)                               # [R, 1]
        compact = self.compact_rows and 0 < Ra < R
        if compact:
            hq, tgt_q, Rq = ops.gather_rows(hid, idx, Ra, pol.d), tgt_rows, Ra
        else:
            tgt = torch.zeros(R, dtype=torch.int32, device=dev)
            tgt.view(B, T)[:, :-1] = ids_d[:, 1:]
            idx, hq, tgt_q, Rq = None, hid, tgt, R
        logits, logits_b, lse, lp_q = pol.lm_ce(hq, Rq, tgt_q)                    # fp32 [Rq, V] logits, or (bf16-matmul mode) bf16 ones + fp32-exact lse
        if compact:
            logprob_all = torch.zeros(R, dtype=torch.float32, device=dev)          # zeros off the mask (multiplied by the zero mask in the loss)
            ops.scatter_rows(lp_q, idx, logprob_all, Ra, 1, False)
        else:
            logprob_all = lp_q
        sl = lambda x: x.view(B, T)[:, :-1].contiguous()
        loss, logs, dlp, dv = ppo_loss_device(attn_s, sl(logprob_all), sl(values_full.view(R)), sta_d,
                                              f32(old_logprobs), f32(old_values), f32(old_advantages), f32(old_returns), **self.loss_kwargs)
        use_bc = bc_data_input_ids is not None
        if use_bc and not train:
            bc_loss = self._bc_term(bc_data_input_ids, bc_data_input_attention_mask, bc_data_input_position_ids, bc_data_input_training_mask, None)
            total = loss + bc_loss * self.bc_loss_weight
            return self, total, {"ppo": logs, "bc": {"loss": np.float32(bc_loss)}, "total_loss": np.float32(total)}
        if not train:
            return self, loss, logs
        # ---- backward: d loss / d logprob[r] -> logits via CE backward (logprob = -CE -> coef_ce = -dlp) ; values -> head
        coef = torch.zeros(R, dtype=torch.float32, device=dev)
        neg = torch.empty_like(dlp)
        ops.axpby(-1.0, dlp, 0.0, None, neg)
        coef.view(B, T)[:, :-1] = neg
        coef_q = ops.gather_rows(coef.view(R, 1), idx, Ra, 1).view(Ra) if compact else coef
        dlogits, dlb = pol.ce_bwd_any(logits, logits_b, lse, tgt_q, coef_q, None, Rq)   # in place (fp32 / bf16 logits)
        pgrads, hgrads = pol.zero_grads(), head.zero_grads()
        if compact:
            dhq = torch.empty(Ra, pol.d, dtype=torch.float32, device=dev)
            pol.lm_head_backward(hq, dlogits, Ra, dhq, pgrads, accumulate_dh=False, dlb=dlb)
            d_hidden = torch.zeros(R, pol.d, dtype=torch.float32, device=dev)
            ops.scatter_rows(dhq, idx, d_hidden, Ra, pol.d, False)
        else:
            d_hidden = torch.empty(R, pol.d, dtype=torch.float32, device=dev)
            pol.lm_head_backward(hid, dlogits, R, d_hidden, pgrads, accumulate_dh=False, dlb=dlb)
        dvals = torch.zeros(R, 1, dtype=torch.float32, device=dev)
        dvals.view(B, T)[:, :-1] = dv
        head.backward(hcache, dvals, hgrads, dx=d_hidden, accumulate_dx=True)
        # data parallel: the policy gradient all-reduce overlaps this backward (arena slices are handed to RCCL as blocks finish) unless a
        # BC auxiliary batch still has to accumulate into the same gradients afterwards
        red = D.GradReducer()
        pol.backward(cache, d_hidden, pgrads, on_final=None if use_bc else red.ready(pgrads))
        if use_bc:   # policy_grads += bc_loss_weight * bc_grads ; loss += bc_loss_weight * bc_loss  (gpt2/interface.py:180-203)
            del cache, logits, d_hidden
            bc_loss = self._bc_term(bc_data_input_ids, bc_data_input_attention_mask, bc_data_input_position_ids, bc_data_input_training_mask, pgrads)
            total = loss + bc_loss * self.bc_loss_weight
            loss, logs = total, {"ppo": logs, "bc": {"loss": np.float32(bc_loss)}, "total_loss": np.float32(total)}
        self.last_grads = (pgrads, hgrads)
        red.finish([pgrads, hgrads] if use_bc else [hgrads])      # the one data-path collective of a PPO step (RCCL over xGMI)
        self.policy_opt.apply(pgrads)
        self.head_opt.apply(hgrads)
        return self, loss, logs
