"""Oracle for the per-token RL math (TEST INFRASTRUCTURE).

CPU restatements (numpy / torch-CPU, float64 unless the reference itself computes in
float32 numpy) of the reference's PPO / ILQL / MC / BC arithmetic.  Each function cites
the reference lines it follows (paths relative to /root/reference/LLM_RL).

Pinning status
  * get_action_state_next_state_idxs, gae, AdaptiveKLController, chain/data shaping:
    PINNED against tests/golden/rl_helpers.json (outputs of the reference's own numpy code).
  * ppo_loss, ilql_loss, get_query_indicators, mc_loss, bc_loss, whiten, get_rtg, token_logprobs_from_logits,
    tensor_stats: PINNED against tests/golden/rl_losses.json — outputs of the reference's own, unmodified JAX
    functions executed under a numpy-backed `jax.numpy` / `optax` shim (tests/golden/_jnp_shim.py,
    make_loss_fixtures.py): loss, the complete log dict, and directional derivatives taken by complex-step
    differentiation THROUGH the reference code (so its stop_gradient placement is pinned too);
    tests/test_oracle_losses_pinned.py.
  * still "parity unpinned": the flax head modules (linear_head / mlp_head: flax is not installable; they are a
    Dense / Dense-relu-Dense restatement), ilql_gather_qv (lives inside a pjit closure of the reference), the
    GPT-2 forward vs the JAX model and the sampling stream (oracle/gpt2.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

F64 = torch.float64


# ----------------------------------------------------------------------------- index helpers / scans
def get_action_state_next_state_idxs(should_take_action: np.ndarray):
    """algorithms/ppo/base_interface.py:230-243."""
    sta = np.asarray(should_take_action, dtype=bool)
    action_idxs = np.where(sta)[0]
    state_idxs = np.where(sta)[0]
    is_next = sta.copy()
    if is_next.size:
        is_next[np.argmax(is_next.astype(np.int32))] = False
    is_next = np.concatenate((is_next, np.array([sta.sum() > 0])))
    next_state_idxs = np.where(is_next)[0]
    return action_idxs, state_idxs, next_state_idxs


def gae(state_values, next_state_values, action_rewards, gamma: float, lam: float, dtype=np.float32):
    """get_advantages_and_returns without whitening (ppo/base_interface.py:253-293); [b, n] arrays.
    The reference runs this in numpy on float32 inputs with Python-float gamma/lam -> float32 arithmetic."""
    v = np.asarray(state_values, dtype=dtype); nv = np.asarray(next_state_values, dtype=dtype)
    r = np.asarray(action_rewards, dtype=dtype)
    n = v.shape[1]
    last = 0
    adv_rev = []
    for t in reversed(range(n)):
        delta = r[:, t] + gamma * nv[:, t] - v[:, t]
        last = delta + gamma * lam * last
        adv_rev.append(last)
    adv = np.stack(adv_rev[::-1], axis=1)
    return adv, adv + v


def whiten(xs, shift_mean: bool = True):
    """ppo/base_interface.py:245-251 (population variance, eps 1e-8 inside the sqrt)."""
    xs = np.asarray(xs, dtype=np.float64)
    mean, var = xs.mean(), xs.var()
    out = (xs - mean) * (1.0 / np.sqrt(var + 1e-8))
    if not shift_mean:
        out = out + mean
    return out


def get_rtg(rewards, gamma: float, dtype=np.float64):
    """mc_returns/data.py:10-14: sum_j>=i (g^(j+1) / g^(i+1)) r_j via a cumprod ratio (not g^(j-i))."""
    r = np.asarray(rewards, dtype=dtype)
    row = np.cumprod(np.full((r.shape[0],), gamma, dtype=dtype))
    tens = np.triu(row[None, :] / row[:, None])
    return (tens * r[None, :]).sum(axis=1)


class AdaptiveKLController:
    """ppo/base_interface.py:38-56."""

    def __init__(self, init_kl_coef, target, horizon):
        self.value, self.target, self.horizon = init_kl_coef, target, horizon

    def update(self, current, n_steps):
        err = np.clip(current / self.target - 1, -0.2, 0.2)
        self.value *= 1 + err * n_steps / self.horizon


def tensor_stats(xs: torch.Tensor, mask: torch.Tensor, n) -> Dict[str, torch.Tensor]:
    """utils.py:12-21: mean = sum(xs*mask)/n ; min/max/std over mask==True (population std)."""
    xs = xs.to(F64); maskf = mask.to(F64)
    mean = (xs * maskf).sum() / n
    mb = mask.bool()
    sel = xs[mb]
    if sel.numel() == 0:
        return dict(mean=mean, min=torch.tensor(float("inf"), dtype=F64), max=torch.tensor(float("-inf"), dtype=F64),
                    std=torch.tensor(float("nan"), dtype=F64))
    return dict(mean=mean, min=sel.min(), max=sel.max(), std=sel.std(unbiased=False))


# ----------------------------------------------------------------------------- losses (torch f64, autograd-able)
def token_logprobs_from_logits(logits: torch.Tensor, input_ids: torch.Tensor) -> torch.Tensor:
    """ppo/base_interface.py:396-403: -CE(logits[:, :-1], ids[:, 1:])."""
    lp = torch.log_softmax(logits[:, :-1].to(F64), dim=-1)
    return lp.gather(-1, input_ids[:, 1:].long().unsqueeze(-1)).squeeze(-1)


def ppo_loss(attention_mask, logprobs, values, should_take_action, old_logprobs, old_values, old_advantages,
             old_returns, *, cliprange_value, cliprange, value_loss_coef):
    """ppo/base_interface.py:72-142."""
    mask = should_take_action.to(F64) * attention_mask.to(F64)
    n = mask.sum()
    values_clipped = torch.maximum(torch.minimum(values, old_values + cliprange_value), old_values - cliprange_value)
    vf1 = (values - old_returns) ** 2
    vf2 = (values_clipped - old_returns) ** 2
    vf_loss = 0.5 * (torch.maximum(vf1, vf2) * mask).sum() / n
    vf_clipfrac = ((vf2 > vf1).to(F64) * mask).sum() / n
    log_ratio = (logprobs - old_logprobs) * mask
    ratio = torch.exp(log_ratio)
    approx_kl = ((ratio - 1) - log_ratio).sum() / n
    pg1 = -old_advantages * ratio
    pg2 = -old_advantages * torch.clamp(ratio, 1.0 - cliprange, 1.0 + cliprange)
    pg_loss = (torch.maximum(pg1, pg2) * mask).sum() / n
    pg_clipfrac = ((pg2 > pg1).to(F64) * mask).sum() / n
    loss = pg_loss + value_loss_coef * vf_loss
    logs = dict(
        losses=dict(total_loss=loss, policy_loss=pg_loss, value_loss=vf_loss),
        values=dict(tensor_stats(values, mask, n), values_error=(((values - old_returns) * mask) ** 2).sum() / n,
                    clipfrac=vf_clipfrac),
        old_values=tensor_stats(old_values, mask, n),
        returns=tensor_stats(old_returns, mask, n),
        policy=dict(approx_kl=approx_kl, clipfrac=pg_clipfrac),
        ratio=(ratio * mask).sum() / n,
        padding_percentage=n / mask.numel(),
    )
    return loss, logs


def get_query_indicators(flat_mask: torch.Tensor) -> torch.Tensor:
    """ilql/base_interface.py:22-27: row k is the one-hot of the k-th True position (all-zero rows after)."""
    N = flat_mask.shape[0]
    idxs = torch.nonzero(flat_mask.bool())[:, 0]
    out = torch.zeros((N, N), dtype=F64)
    out[torch.arange(idxs.shape[0]), idxs] = 1.0
    return out


def _l2(pred, target):
    return 0.5 * (pred - target) ** 2  # optax.l2_loss


def ilql_loss(q1, q2, v, v_final, target_q1, target_q2, q1_logits, q2_logits, token_ids, attention_mask,
              should_take_action, rewards, *, gamma, tau, cql_weight):
    """ilql/base_interface.py:29-119 — literal restatement (one-hot query indicators and all)."""
    sta = should_take_action.bool()
    mask = sta.to(F64) * attention_mask.to(F64)
    n = mask.sum()
    B = sta.shape[0]
    vns_flat = torch.cat((v, v_final[..., None]), dim=1).reshape(-1)
    qv_ind = get_query_indicators(sta.reshape(-1))
    is_next = sta.clone()
    first = torch.argmax(is_next.to(torch.int32), dim=1)
    is_next[torch.arange(B), first] = False
    is_next = torch.cat((is_next, (sta.sum(dim=1) > 0)[..., None]), dim=1)
    vns_ind = get_query_indicators(is_next.reshape(-1))[: qv_ind.shape[0], :]

    sel = lambda ind, x: (ind * x.reshape(-1).to(F64)).sum(dim=1)
    q1s, q2s, vs = sel(qv_ind, q1), sel(qv_ind, q2), sel(qv_ind, v)
    tq1s, tq2s = sel(qv_ind, target_q1), sel(qv_ind, target_q2)
    vnss = (vns_ind * vns_flat.to(F64)).sum(dim=1)
    rs = sel(qv_ind, rewards)
    sa_mask = (qv_ind.sum(dim=1) > 0).to(F64)
    ns_mask = (vns_ind.sum(dim=1) > 0).to(F64)

    tgt = (rs + gamma * vnss).detach()
    q1_loss = (_l2(q1s, tgt) * sa_mask).sum() / n
    q2_loss = (_l2(q2s, tgt) * sa_mask).sum() / n
    tq = torch.minimum(tq1s, tq2s)
    ind = (tq >= vs).to(F64)
    w = (ind * tau + (1 - ind) * (1 - tau)).detach()
    v_loss = (_l2(vs, tq.detach()) * w * sa_mask).sum() / n
    ce = lambda lg: -torch.log_softmax(lg.to(F64), dim=-1).gather(-1, token_ids.long().unsqueeze(-1)).squeeze(-1)
    q1_cql = (mask * ce(q1_logits)).sum() / n
    q2_cql = (mask * ce(q2_logits)).sum() / n
    loss = q1_loss + q2_loss + v_loss + cql_weight * (q1_cql + q2_cql)
    logs = dict(
        losses=dict(total_loss=loss, q1_loss=q1_loss, q2_loss=q2_loss, v_loss=v_loss, q1_cql_loss=q1_cql, q2_cql_loss=q2_cql),
        q1=tensor_stats(q1s, sa_mask, n), q2=tensor_stats(q2s, sa_mask, n), v=tensor_stats(vs, sa_mask, n),
        target_q=tensor_stats(tq, sa_mask, n), target_q1=tensor_stats(tq1s, sa_mask, n), target_q2=tensor_stats(tq2s, sa_mask, n),
        vns=tensor_stats(vnss, ns_mask, n),
        v_final=tensor_stats(v_final, torch.ones_like(v_final), v_final.shape[0]),
        rewards=tensor_stats(rewards, mask, n),
    )
    return loss, logs


def ilql_gather_qv(q1_head_out, q2_head_out, v_head_out, tq1_head_out, tq2_head_out, input_ids, attention_mask,
                   should_take_action, dones, next_v_head_out=None, next_attention_mask=None, next_dones=None):
    """ilql/gpt2/interface.py:241-273: Q(s,a) gathers, v, v_full, v_final.  With `next_v_head_out` (the V head applied to
    the hidden states of `next_token_ids`, [B, T', 1]) v_final follows the next-token branch (:252-264)."""
    ids = input_ids[:, 1:].long().unsqueeze(-1)
    take = lambda h: h[:, :-1].gather(2, ids).squeeze(2)
    q1, q2 = take(q1_head_out), take(q2_head_out)
    tq1, tq2 = take(tq1_head_out).detach(), take(tq2_head_out).detach()
    v_full = v_head_out.squeeze(2)
    v = v_full[:, :-1]
    T1 = should_take_action.shape[1]
    last_action = (T1 - 1) - torch.argmax(torch.flip(should_take_action.to(torch.int32), dims=[1]), dim=1) + 1
    last_token = (attention_mask.shape[1] - 1) - torch.argmax(torch.flip(attention_mask.to(torch.int32), dims=[1]), dim=1)
    d = dones.to(F64)
    final_idx = ((1 - d) * last_action + d * last_token).to(torch.int64)
    v_final = v_full[torch.arange(v_full.shape[0]), final_idx] * (1 - d)
    if next_v_head_out is not None:
        nam = next_attention_mask
        last_next = (nam.shape[1] - 1) - torch.argmax(torch.flip(nam.to(torch.int32), dims=[1]), dim=1)
        v_final = next_v_head_out.squeeze(2)[torch.arange(nam.shape[0]), last_next] * (1 - next_dones.to(F64))
    return q1, q2, v, v_final.detach(), tq1, tq2


def mc_loss(q, q_logits, token_ids, attention_mask, should_take_action, returns, *, cql_weight):
    """mc_returns/base_interface.py:19-60."""
    sta = should_take_action.bool()
    mask = sta.to(F64) * attention_mask.to(F64)
    n = mask.sum()
    ind = get_query_indicators(sta.reshape(-1))
    qs = (ind * q.reshape(-1).to(F64)).sum(dim=1)
    rs = (ind * returns.reshape(-1).to(F64)).sum(dim=1)
    a_mask = (ind.sum(dim=1) > 0).to(F64)
    q_loss = (_l2(qs, rs.detach()) * a_mask).sum() / n
    ce = -torch.log_softmax(q_logits.to(F64), dim=-1).gather(-1, token_ids.long().unsqueeze(-1)).squeeze(-1)
    q_cql = (mask * ce).sum() / n
    loss = q_loss + cql_weight * q_cql
    return loss, dict(losses=dict(total_loss=loss, q_loss=q_loss, q_cql_loss=q_cql),
                      q=tensor_stats(qs, a_mask, n), returns=tensor_stats(rs, a_mask, n))


def bc_loss(logits, input_ids, attention_mask, is_action, *, non_action_weight):
    """bc/interface.py:28-43."""
    am = attention_mask[:, 1:].to(F64)
    ce = -torch.log_softmax(logits[:, :-1].to(F64), dim=-1).gather(-1, input_ids[:, 1:].long().unsqueeze(-1)).squeeze(-1)
    tl = ce * am
    ia = is_action[:, 1:].to(F64)
    tl = ia * tl + (1 - ia) * tl * non_action_weight
    return tl.sum() / am.sum()


# ----------------------------------------------------------------------------- heads
def linear_head(x, W, b):
    """heads/linear_head.py:112-119 (flax Dense: x @ kernel + bias, kernel [in, out])."""
    return x.to(F64) @ W.to(F64) + b.to(F64)


def mlp_head(x, W1, b1, W2, b2):
    """heads/mlp_head.py:139-148."""
    return torch.relu(x.to(F64) @ W1.to(F64) + b1.to(F64)) @ W2.to(F64) + b2.to(F64)


def value_rl_logits(pi_beta_logits, q1_logits, q2_logits, beta):
    """value_rl_base/gpt2/generation.py:112-119."""
    q = q1_logits if q2_logits is None else torch.minimum(q1_logits, q2_logits)
    return beta * q if pi_beta_logits is None else pi_beta_logits + beta * q


# ----------------------------------------------------------------------------- data shaping
def ilql_data_from_chain(chain: List[dict]) -> dict:
    """ilql/data.py:58-79.  `chain` = list of token trajectories (dict tokens/is_action/reward/done), head first."""
    cur = chain[0]
    if len(chain) > 1:
        nxt = chain[1]
        ia = np.asarray(nxt["is_action"], dtype=bool)
        if ia[1:].sum() > 0:
            first = int(np.argmax(ia[1:])) + 1
            next_token_ids, next_done = list(nxt["tokens"][:first]), False
        else:
            next_token_ids, next_done = list(nxt["tokens"]), bool(nxt["done"])
    else:
        next_token_ids, next_done = None, None
    return dict(input_ids=list(cur["tokens"]), should_take_action=[int(x) for x in cur["is_action"][1:]],
                rewards=list(cur["reward"][1:]), done=bool(cur["done"]), next_token_ids=next_token_ids, next_done=next_done)


def truncate_turns_record(rec: dict, max_length: int, gamma: float) -> Optional[dict]:
    """The task scripts' length rule (llm_rl_scripts/wordle/ppo/train_ppo_gpt2.py:323-341) on ONE token record
    {tokens, is_action, reward, done} of a single-trajectory chain, where a "text" is a maximal run of equal is_action flags and a text's
    reward sits on its last token (environment.py:349-380).  Returns the kept record, or None when the script skips the episode.
    PINNED against tests/golden/ppo_truncation.json (the reference's own loop executed, tests/test_oracle_rl.py)."""
    tokens, ia = list(rec["tokens"]), [bool(x) for x in rec["is_action"]]
    starts = [0] + [t for t in range(1, len(tokens)) if ia[t] != ia[t - 1]] if tokens else []
    ends = starts[1:] + [len(tokens)]
    reward = [float(np.float32(rec["reward"][e - 1])) for e in ends]          # one Python float per text, :319
    done = bool(rec["done"])
    n_items = len(starts)
    while n_items > 3:                                                         # :323
        if ends[n_items - 1] >= max_length:                                    # tokens of the (current) trajectory, :324
            new_reward = reward[:n_items][:-2]
            new_reward[-2] += sum(reward[:n_items][-2:]) * gamma               # :325-326
            reward[:n_items - 2] = new_reward
            n_items -= 2
            done = False                                                       # :327-332
        else:
            break
    if n_items < 3:                                                            # :336-337
        return None
    n = ends[n_items - 1]
    if n >= max_length:                                                        # :338-339
        return None
    out_r = [0.0] * n
    for k in range(n_items):
        out_r[ends[k] - 1] = float(np.float32(reward[k]))                      # np.array(reward, dtype=np.float32), environment.py:377
    return dict(tokens=tokens[:n], is_action=[int(x) for x in ia[:n]], reward=out_r, done=done)


def combined_chain(chain: List[dict], max_length: Optional[int] = None) -> dict:
    """CombinedTokenTrajectoryChain.from_token_trajectory_chain (ppo/base_interface.py:303-336)."""
    if max_length is None:
        max_length = max(len(t["tokens"]) for t in chain) + 1
    assert not any(t["done"] for t in chain[:-1]), "done can only be true at the end of the chain"
    for i, t in enumerate(chain):
        ia = np.asarray(t["is_action"], dtype=bool)
        no_trunc = (len(t["tokens"]) - 1) <= max_length
        ends_with_state = not np.any(ia[1:][max_length:])
        next_starts_with_action = i < len(chain) - 1 and bool(chain[i + 1]["is_action"][0])
        assert not (ends_with_state and next_starts_with_action), "trajectory truncation error"
        assert no_trunc or ends_with_state, "trajectory truncation error"
    cat = lambda key, sl: [x for t in chain for x in sl(list(t[key]))[:max_length]]
    return dict(
        input_tokens=cat("tokens", lambda l: l[:-1]), output_tokens=cat("tokens", lambda l: l[1:]),
        rewards=cat("reward", lambda l: l[1:]), should_take_action=[int(x) for x in cat("is_action", lambda l: l[1:])],
        done=bool(chain[-1]["done"]), chunk_lens=[min(len(t["tokens"]) - 1, max_length) for t in chain])


def mc_data_from_chain(chain: List[dict], gamma: float) -> dict:
    """mc_returns/data.py:49-74."""
    filt, _ = [], []
    for t in chain:
        ia = np.asarray(t["is_action"], dtype=bool)[1:]
        filt.append(np.asarray(t["reward"], dtype=np.float32)[1:][ia])
    rtg = get_rtg(np.concatenate(filt), gamma, dtype=np.float32)
    sta = np.asarray(chain[0]["is_action"], dtype=bool)[1:]
    ret = np.zeros(sta.shape, dtype=np.float32)
    ret[sta] = rtg[: sta.sum()]
    return dict(input_ids=list(chain[0]["tokens"]), should_take_action=sta.astype(int).tolist(), returns=ret)


def block_sequences(seqs, pad_value, dtype, max_length: Optional[int], padding="right", truncation="right"):
    """JaxSeq.utils.block_sequences semantics as used at ppo/data.py:24-59 (3rd-party, unverified):
    truncate then pad every sequence to `max_length` (or the longest)."""
    if max_length is None:
        max_length = max(len(s) for s in seqs)
    out = np.full((len(seqs), max_length), pad_value, dtype=dtype)
    for i, s in enumerate(seqs):
        s = list(s)
        if len(s) > max_length:
            s = s[:max_length] if truncation == "right" else s[-max_length:]
        if padding == "right":
            out[i, : len(s)] = s
        else:
            out[i, max_length - len(s):] = s
    return out


def ppo_data_from_chains(chains: List[List[dict]], policy_logprobs: List[np.ndarray], init_logprobs: List[np.ndarray],
                         values: List[np.ndarray], *, gamma, lam, kl_weight, use_advantage_whitening=True):
    """Post-forward half of get_ppo_data_from_token_trajectory_chain (ppo/base_interface.py:538-669).

    chains[i]   : list of token-trajectory dicts of chain i
    *_logprobs[i]: per-chunk token logprobs, already un-padded and concatenated over the chain ([sum chunk_lens])
    values[i]   : per-chunk values[:-1] concatenated + the bootstrap slot (last value * (1-done)) appended
                  ([sum chunk_lens + 1]) — i.e. `values_chains[i]` at :566-570.
    Returns (per-chain dict of old_logprobs/old_values/old_advantages/old_returns over the concatenated chain, all_kls).
    """
    combos = [combined_chain(c) for c in chains]
    log_ratio = [(np.asarray(p) - np.asarray(q)) * np.asarray(c["should_take_action"], dtype=np.float32)
                 for p, q, c in zip(policy_logprobs, init_logprobs, combos)]
    valid = np.argwhere(np.concatenate([np.asarray(c["should_take_action"], dtype=np.float32) for c in combos]))[:, 0]
    all_lr = np.concatenate([x.reshape(-1) for x in log_ratio])[valid]
    all_kls = np.exp(all_lr) - 1 - all_lr
    rewards = [np.asarray(c["rewards"], dtype=np.float32) - kl_weight * lr for c, lr in zip(combos, log_ratio)]
    advs, rets, idxs = [], [], []
    for i, c in enumerate(combos):
        a_idx, s_idx, n_idx = get_action_state_next_state_idxs(np.asarray(c["should_take_action"], dtype=bool))
        v = np.asarray(values[i])
        adv, ret = gae(v[s_idx][None], v[n_idx][None], rewards[i][a_idx][None], gamma, lam, dtype=v.dtype)
        advs.append(adv[0]); rets.append(ret[0]); idxs.append(a_idx)
    if use_advantage_whitening:
        w = whiten(np.concatenate(advs), shift_mean=True)
        pos = 0
        for i in range(len(advs)):
            advs[i] = w[pos: pos + len(advs[i])]; pos += len(advs[i])
    out = []
    for i, c in enumerate(combos):
        L = len(values[i]) - 1
        a = np.zeros((L,), dtype=np.float32); a[idxs[i]] = advs[i]
        r = np.zeros((L,), dtype=np.float32); r[idxs[i]] = rets[i]
        out.append(dict(old_logprobs=np.asarray(policy_logprobs[i]), old_values=np.asarray(values[i])[:-1],
                        old_advantages=a, old_returns=r, should_take_action=np.asarray(c["should_take_action"]),
                        chunk_lens=c["chunk_lens"]))
    return out, all_kls
