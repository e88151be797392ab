"""Rollout records -> PPO data -> PPO batches without leaving the device.

The reference's online loop is `text_env_eval -> TextTrajectoryChain -> get_ppo_data_from_text_trajectory_chain -> PPODataset -> train`
(llm_rl_scripts/wordle/ppo/train_ppo_gpt2.py:301-353, LLM_RL/algorithms/ppo/base_interface.py:464-669, LLM_RL/algorithms/ppo/data.py:9-114): Python
text, a re-tokenisation, per-chain numpy loops.  The rollout engines of this package already leave `TokenTrajectory` fields in HBM
(`tokens / is_action / reward / n_tok / done`), so here the same function runs on those records where they lie:

    records --lmrl_ppo_count / lmrl_ppo_block--> ids, mask, positions, LM-head row list        (csrc/ppo_data.hip)
            --policy, initial policy (inference forwards), value head--> log-probs on the listed rows, values
            --lmrl_ppo_shape--> KL terms, KL-shaped rewards, chain rows --lmrl_gae, lmrl_whiten_*--> --lmrl_ppo_unroll--> DevicePPODataset

and `GPT2PPOTrain.step` takes `DevicePPODataset.batch(...)` (device tensors) as it takes the reference's numpy batches.  No `[B, T, V]` logits
exist at any point (bf16-matmul mode: log-sum-exp from the LM-head GEMM's accumulators; fp32: one row-chunk scratch).  The host reads back
(2 n + 10) integers per call (row offsets + the totals and the reference's truncation checks) to size the launches.

`get_ppo_data_from_token_trajectory_chain` in ppo_inference.py stays as the host-array form (the reference's own signature); both are tested
against the reference function's fixture and against each other (tests/test_gpu_ppo_device.py).
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .. import _lib
from .. import dist as D
from ..train import ops
from .ppo import PPODataset


class _CRecords(ctypes.Structure):            # lmrl_ppo_records (include/lmrl_amd.h)
    _fields_ = [(n, ctypes.c_void_p) for n in ("tokens", "is_action", "reward", "n_tok", "done", "chain", "pos", "last")] + \
               [("n", ctypes.c_int32), ("cap", ctypes.c_int32), ("n_chains", ctypes.c_int32)]


class PPORecords:
    """n token trajectories in HBM, grouped into chains: what a rollout engine hands over (`WordleRolloutEngine.ppo_records()`), or what
    `from_token_trajectory_chains` uploads for chains built on the host."""

    def __init__(self, tokens, is_action, reward, n_tok, done, chain=None, pos=None, last=None, n_chains: Optional[int] = None,
                 chain_len_bound: Optional[int] = None):
        """tokens int32 [n, cap], is_action uint8 [n, cap], reward float32 [n, cap], n_tok int32 [n], done uint8 [n_chains] — device tensors.
        chain / pos (int32 [n]) / last (uint8 [n]): see include/lmrl_amd.h::lmrl_ppo_records; None = every trajectory is its own chain.
        chain_len_bound: an upper bound of the longest chain's concatenated length (multi-trajectory chains only)."""
        self.tokens, self.is_action, self.reward, self.n_tok, self.done = tokens, is_action, reward, n_tok, done
        self.chain, self.pos, self.last = chain, pos, last
        self.n, self.cap = int(tokens.shape[0]), int(tokens.shape[1])
        self.n_chains = self.n if n_chains is None else int(n_chains)
        self.chain_len_bound = chain_len_bound
        assert (chain is None) == (pos is None) == (last is None)
        assert chain is not None or self.n_chains == self.n
        for x in (tokens, is_action, reward):
            assert x.is_contiguous() and tuple(x.shape) == (self.n, self.cap)

    def c_struct(self) -> _CRecords:
        p = _lib.ptr
        return _CRecords(p(self.tokens), p(self.is_action), p(self.reward), p(self.n_tok), p(self.done), p(self.chain), p(self.pos), p(self.last),
                         self.n, self.cap, self.n_chains)

    @classmethod
    def from_token_trajectory_chains(cls, chains, max_length: Optional[int] = None, device=None) -> "PPORecords":
        """Upload host `TokenTrajectoryChain`s (LLM_RL/environment.py:383-420): one row per trajectory, chains concatenated in order."""
        import torch
        dev = device or _lib.require_gpu()
        tts, chain, pos, last, done = [], [], [], [], []
        for c, ch in enumerate(chains):
            lst = ch.to_list()
            assert not any(bool(tt.done) for tt in lst[:-1]), "done can only be true at the end of the chain"
            p = 0
            for i, tt in enumerate(lst):
                n = int(tt.tokens.shape[0]) if max_length is None else min(int(tt.tokens.shape[0]), int(max_length))
                tts.append(tt); chain.append(c); pos.append(p); last.append(i == len(lst) - 1)
                p += max(n - 1, 0)
            done.append(bool(lst[-1].done))
        cap = max(int(tt.tokens.shape[0]) for tt in tts)
        tok = np.zeros((len(tts), cap), np.int32); ia = np.zeros((len(tts), cap), np.uint8); rw = np.zeros((len(tts), cap), np.float32)
        for k, tt in enumerate(tts):
            n = int(tt.tokens.shape[0])
            tok[k, :n], ia[k, :n], rw[k, :n] = tt.tokens, np.asarray(tt.is_action, dtype=np.uint8), tt.reward
        up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
        bound = max(p + max((int(tt.tokens.shape[0]) if max_length is None else min(int(tt.tokens.shape[0]), int(max_length))) - 1, 0)
                    for p, tt in zip(pos, tts))
        return cls(up(tok, np.int32), up(ia, np.uint8), up(rw, np.float32), up([int(tt.tokens.shape[0]) for tt in tts], np.int32), up(done, np.uint8),
                   up(chain, np.int32), up(pos, np.int32), up(last, np.uint8), n_chains=len(done), chain_len_bound=bound)


class DevicePPODataset:
    """`PPODataset` (LLM_RL/algorithms/ppo/data.py:63-114) with its six blocked arrays resident in HBM: input_ids int32 [N, T],
    should_take_action uint8 [N, T-1], old_logprobs / old_values / old_advantages / old_returns float32 [N, T-1]."""

    FIELDS = ("input_ids", "should_take_action", "old_logprobs", "old_values", "old_advantages", "old_returns")

    def __init__(self, longest: Optional[int] = None, lengths=None, **arrays):
        """longest: number of tokens of the longest trajectory in the dataset (known from the build's one readback), for `batch(width=...)`.
        lengths (int32 [N], device; optional): every row's trajectory length as the data build saw it.  With it a batch carries the build's OWN
        `attention_mask` / `position_ids` (right-padded rows: 1 below the length), so the train step masks exactly what the log-probs / values were
        computed under even when a token equal to the pad id sits inside a trajectory (a tokenizer whose pad id the policy can sample); without
        it the step derives the masks from `ids != pad` as the reference does (ppo/base_interface.py:190-195)."""
        for k in self.FIELDS:
            setattr(self, k, arrays[k])
        n, t = self.input_ids.shape
        for k in self.FIELDS[1:]:
            assert tuple(getattr(self, k).shape) == (n, t - 1)
        self.longest = int(longest) if longest is not None else int(t)
        self.lengths = lengths
        assert lengths is None or tuple(lengths.shape) == (n,)

    @classmethod
    def concat(cls, parts: Sequence["DevicePPODataset"]) -> "DevicePPODataset":
        """The datasets of several episode batches of one round as one (rows concatenated; the parts share the blocking width)."""
        import torch
        if len(parts) == 1:
            return parts[0]
        T = parts[0].input_ids.shape[1]
        assert all(p.input_ids.shape[1] == T for p in parts), "batches of one round share the blocking width (pass max_length)"
        lengths = torch.cat([p.lengths for p in parts]) if all(p.lengths is not None for p in parts) else None
        return cls(longest=max(p.longest for p in parts), lengths=lengths, **{k: torch.cat([getattr(p, k) for p in parts]) for k in cls.FIELDS})

    def trimmed_width(self, multiple: int = 64) -> int:
        """The narrowest batch width (a multiple of `multiple`) that still holds every trajectory."""
        return min(int(self.input_ids.shape[1]), -(-self.longest // multiple) * multiple)

    def __len__(self) -> int:
        return int(self.input_ids.shape[0])

    def batch(self, index, width: Optional[int] = None) -> Dict[str, "torch.Tensor"]:
        """Rows `index` (a device int32 tensor, or anything numpy can turn into indices) of every array: a shuffled batch of the dataloader
        (`lmrl_gather_rows_bytes`), keyword-compatible with `GPT2PPOTrain.step(**batch)`.
        width (>= `longest`, default: the dataset's blocking width): the batch keeps only the first `width` columns.  The reference blocks every
        batch to max_input_length + max_output_length because XLA wants one static shape; a right-padded causal model computes the same loss,
        logs (but `padding_percentage`) and gradients on `trimmed_width()` columns — at 70-token Wordle episodes 8 x fewer rows than 1024."""
        import torch
        if not isinstance(index, torch.Tensor):
            index = torch.from_numpy(np.ascontiguousarray(index, dtype=np.int32)).to(self.input_ids.device)
        index = index.to(torch.int32).contiguous()
        n = int(index.numel())
        T = int(self.input_ids.shape[1])
        w = T if width is None else int(width)
        if not (self.longest <= w <= T):
            raise ValueError(f"batch width {w} must lie between the longest trajectory ({self.longest}) and the dataset width ({T})")
        out = {}
        for k in self.FIELDS:
            src = getattr(self, k)
            cols = w if k == "input_ids" else w - 1
            dst = torch.empty((n, src.shape[1]), dtype=src.dtype, device=src.device)
            _lib.check(_lib.lib().lmrl_gather_rows_bytes(src.data_ptr(), index.data_ptr(), dst.data_ptr(), n, src.shape[1] * src.element_size(),
                                                         _lib.stream_ptr()), "lmrl_gather_rows_bytes")
            out[k] = dst if cols == src.shape[1] else dst[:, :cols].contiguous()
        if self.lengths is not None:
            ln = torch.empty(n, dtype=torch.int32, device=index.device)
            _lib.check(_lib.lib().lmrl_gather_rows_bytes(self.lengths.data_ptr(), index.data_ptr(), ln.data_ptr(), n, 4, _lib.stream_ptr()), "lmrl_gather_rows_bytes")
            am = torch.empty(n, w, dtype=torch.uint8, device=index.device)
            pos = torch.empty(n, w, dtype=torch.int32, device=index.device)
            _lib.check(_lib.lib().lmrl_len_mask_pos(ln.data_ptr(), n, w, am.data_ptr(), pos.data_ptr(), _lib.stream_ptr()), "lmrl_len_mask_pos")
            out["attention_mask"], out["position_ids"] = am, pos
        return out

    def batches(self, rng, bsize: int, truncate: bool = True, width: Optional[int] = None):
        """One epoch of shuffled batches (`datasets.dataloader` / ppo/train.py:264 on the device dataset)."""
        n = len(self)
        order = np.arange(n) if rng is None else rng.permutation(n)
        stop = n - (n % bsize) if truncate else n
        for i in range(0, stop, bsize):
            yield self.batch(order[i:i + bsize], width=width)

    def to_host(self) -> PPODataset:
        """The reference's host dataset (numpy) — parity tests, pickling (`save_ppo_dataset`)."""
        h = {k: getattr(self, k).cpu().numpy() for k in self.FIELDS}
        h["should_take_action"] = h["should_take_action"].astype(np.bool_)
        return PPODataset(**h)


def ppo_data_from_records(inference, rec: PPORecords, *, gamma: float, lam: float, kl_weight: float, max_length: Optional[int] = None,
                          use_advantage_whitening: bool = True, bsize: int = 256, pad_to: Optional[int] = None, lm_head_rows: int = 8192,
                          timings: Optional[dict] = None) -> Tuple[DevicePPODataset, "torch.Tensor"]:
    """`PPOInference.get_ppo_data_from_token_trajectory_chain` (ppo/base_interface.py:464-669) + `PPODataset.from_ppo_data_list` on device
    records -> (DevicePPODataset, all_kls float32 device tensor).

    inference: `GPT2PPOInference` (policy, initial_policy, value_head, pad id).  max_length: the blocking width of the reference call
    (trajectories are truncated to it on the right); pad_to: width T of the dataset arrays (default: max_length, as the task scripts block the
    dataset; without a max_length the longest trajectory).  bsize: sequences per forward (`ppo_data_bsize`).  The forwards run on
    ceil8(longest trajectory) columns — right padding never reaches a kept position of a causal model, so this equals the reference's
    [bsize, max_length] forwards on the kept positions.  A trajectory's length is its record's `n_tok`: a pad id BELOW the length (a policy
    whose vocabulary contains the pad id can sample it) is an ordinary, attended token here, whereas the reference derives masks from `ids != pad`,
    cuts at the first pad (`unpad_array`) and then fails on the shape mismatch; such tokens are counted on the device (`timings["pad_ids_inside"]`).
    timings: optional dict, filled with HIP-event milliseconds per phase (bench.py's `ppo_iteration` leg)."""
    import torch
    assert inference.initial_policy is not None
    L, sp = _lib.lib(), _lib.stream_ptr()
    pol, init, head, pad = inference.policy, inference.initial_policy, inference.value_head, int(inference.pad)
    dev = pol.dev
    n, ml = rec.n, int(max_length) if max_length is not None else 0
    i32 = lambda *s: torch.empty(*s, dtype=torch.int32, device=dev)
    f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    c = rec.c_struct()
    marks = []

    def mark(name):
        if timings is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append((name, ev))
    mark("start")
    scratch = i32(2 * n + 2 * (n + 1) + 8)
    cnt, off_rows, off_act, meta = scratch[:2 * n], scratch[2 * n:3 * n + 1], scratch[3 * n + 1:4 * n + 2], scratch[4 * n + 2:]
    _lib.check(L.lmrl_ppo_count(ctypes.byref(c), ml, pad, _lib.ptr(cnt), _lib.ptr(off_rows), _lib.ptr(off_act), _lib.ptr(meta), sp), "lmrl_ppo_count")
    host = scratch[2 * n:].cpu().numpy()                                       # the one readback that sizes what follows: row offsets + the totals
    off_rows_h = host[:n + 1]
    n_rows, n_act, longest, pads, cut_actions, bad_starts, chain_reach = (int(x) for x in host[2 * (n + 1):2 * (n + 1) + 7])
    # CombinedTokenTrajectoryChain.from_token_trajectory_chain (base_interface.py:318-327) asserts both conditions; advantages over a cut chain
    # would silently differ from anything the reference computes
    if cut_actions or bad_starts:
        raise ValueError(f"trajectory truncation error: {cut_actions} action token(s) lie beyond max_length = {ml}, {bad_starts} trajectories continue a "
                         "chain but start with an action token (ppo/base_interface.py:318-327 refuses such chains)")
    if n_rows == 0:
        raise ValueError("no trajectory has two tokens: nothing to build PPO data from")
    if pads:
        import warnings
        warnings.warn(f"{pads} token(s) equal to the pad id {pad} lie INSIDE trajectories (a policy that can sample its tokenizer's pad id): they are "
                      "attended here (lengths come from the records) and the dataset carries those lengths to the train step; the reference's "
                      "`ids != pad` masks would cut such a sequence short", RuntimeWarning, stacklevel=2)
    tf = -(-longest // 8) * 8
    tp = int(pad_to) if pad_to is not None else (ml if ml > 0 else longest)
    if tp < longest:
        raise ValueError(f"pad_to = {tp} is narrower than the longest (truncated) trajectory ({longest} tokens)")
    ids, am, pos = i32(n, tf), torch.empty(n, tf, dtype=torch.uint8, device=dev), i32(n, tf)
    rows_idx, tgt = i32(n_rows), i32(n_rows)
    _lib.check(L.lmrl_ppo_block(ctypes.byref(c), ml, pad, tf, _lib.ptr(off_rows), _lib.ptr(ids), _lib.ptr(am), _lib.ptr(pos), _lib.ptr(rows_idx),
                                _lib.ptr(tgt), sp), "lmrl_ppo_block")
    mark("block")
    # ---- the three models (base_interface.py:514-541): per `bsize` sequences one inference forward each, then the log-probs of THAT chunk's rows
    # (the reference keeps one bsize chunk of outputs live too): only the values [n, tf] and the per-row log-probs outlive a chunk.  The bf16
    # operand copies of the weights are staged once per call (nothing moves the masters in between), not once per chunk forward.
    values, lp, init_lp = f32(n * tf), f32(n_rows), f32(n_rows)
    t_fwd = t_lp = 0.0
    evs = []
    for ci, s0 in enumerate(range(0, n, bsize)):
        s1 = min(n, s0 + bsize)
        r0, r1 = int(off_rows_h[s0]), int(off_rows_h[s1])
        local = None
        if r1 > r0:                                                            # the chunk's rows of the row list, as rows of the chunk's hidden states
            local = i32(r1 - r0)
            _lib.check(L.lmrl_add_i32(_lib.ptr(rows_idx[r0:r1]), -s0 * tf, _lib.ptr(local), r1 - r0, sp), "lmrl_add_i32")
        for model, dst in ((init, init_lp), (pol, lp)):
            if timings is not None:
                evs.append(("f0", _ev(torch)))
            h, _ = model.forward(ids[s0:s1], am[s0:s1], pos[s0:s1], inference=True, restage=(ci == 0))
            if model is pol:
                v, _ = head.forward(h, (s1 - s0) * tf)
                values[s0 * tf:s1 * tf].copy_(v.view(-1) if head.ld_out == 1 else v[:, 0])
            if timings is not None:
                evs.append(("f1", _ev(torch)))
            if local is not None:
                dst[r0:r1].copy_(model.token_logprobs(h, local, tgt[r0:r1], r1 - r0, chunk=lm_head_rows))
            if timings is not None:
                evs.append(("l1", _ev(torch)))
            del h
    mark("models")
    # ---- :543-584 on the device, chain rows for the GAE
    lc = max(longest - 1, 1) if rec.chain is None else int(rec.chain_len_bound or n * max(longest - 1, 1))
    if lc < chain_reach:            # lmrl_ppo_shape / lmrl_ppo_unroll would drop the slots beyond lc
        raise ValueError(f"chain_len_bound = {lc} is shorter than the longest chain ({chain_reach} slots)")
    C = rec.n_chains
    cv, cr, cs, clen = f32(C, lc + 1), f32(C, lc), torch.empty(C, lc, dtype=torch.uint8, device=dev), i32(C)
    kls = f32(max(n_act, 1))
    ds = dict(input_ids=i32(n, tp), should_take_action=torch.empty(n, tp - 1, dtype=torch.uint8, device=dev), old_logprobs=f32(n, tp - 1),
              old_values=f32(n, tp - 1), old_advantages=f32(n, tp - 1), old_returns=f32(n, tp - 1))
    _lib.check(L.lmrl_ppo_shape(ctypes.byref(c), ml, tf, _lib.ptr(off_rows), _lib.ptr(off_act), _lib.ptr(lp), _lib.ptr(init_lp), _lib.ptr(values),
                                float(kl_weight), lc, _lib.ptr(cv), _lib.ptr(cr), _lib.ptr(cs), _lib.ptr(clen), _lib.ptr(kls), pad, tp,
                                _lib.ptr(ds["input_ids"]), _lib.ptr(ds["should_take_action"]), _lib.ptr(ds["old_logprobs"]), _lib.ptr(ds["old_values"]), sp),
               "lmrl_ppo_shape")
    adv, ret = f32(C, lc), f32(C, lc)
    npart = L.lmrl_gae_moments_partials(C, lc) if use_advantage_whitening else 0
    if npart > 0:       # rollout-sized chains: the GAE launch leaves the whitening's partial moments, the apply launch adds them up itself
        part = torch.empty(npart, 3, dtype=torch.float64, device=dev)
        _lib.check(L.lmrl_gae_moments(_lib.ptr(cv), _lib.ptr(cr), _lib.ptr(cs), _lib.ptr(clen), _lib.ptr(adv), _lib.ptr(ret), C, lc, float(gamma), float(lam),
                                      _lib.ptr(part), sp), "lmrl_gae_moments")
        adv = D.whiten_distributed(adv.view(-1), cs.view(-1), shift_mean=True, partials=part).view(C, lc)
    else:
        _lib.check(L.lmrl_gae(_lib.ptr(cv), _lib.ptr(cr), _lib.ptr(cs), _lib.ptr(clen), _lib.ptr(adv), _lib.ptr(ret), C, lc, float(gamma), float(lam), sp), "lmrl_gae")
        if use_advantage_whitening:                                        # over the action tokens of the whole batch (all ranks), :609-615
            adv = D.whiten_distributed(adv.view(-1), cs.view(-1), shift_mean=True).view(C, lc)
    _lib.check(L.lmrl_ppo_unroll(ctypes.byref(c), ml, tf, lc, _lib.ptr(adv), _lib.ptr(ret), tp, _lib.ptr(ds["old_advantages"]), _lib.ptr(ds["old_returns"]), sp),
               "lmrl_ppo_unroll")
    mark("shape_gae")
    if timings is not None:
        torch.cuda.synchronize()
        for (_, a), (name, b) in zip(marks[:-1], marks[1:]):
            if name != "models":
                timings[name + "_ms"] = timings.get(name + "_ms", 0.0) + a.elapsed_time(b)
        for k in range(0, len(evs), 3):                                        # forwards and log-prob passes interleave per chunk: summed per kind
            timings["forward_ms"] = timings.get("forward_ms", 0.0) + evs[k][1].elapsed_time(evs[k + 1][1])
            timings["logprobs_ms"] = timings.get("logprobs_ms", 0.0) + evs[k + 1][1].elapsed_time(evs[k + 2][1])
        timings.update(rows=n_rows, action_tokens=n_act, forward_width=tf, sequences=n, pad_ids_inside=timings.get("pad_ids_inside", 0) + pads)
    lengths = i32(n)
    _lib.check(L.lmrl_add_i32(_lib.ptr(cnt[:n]), 1, _lib.ptr(lengths), n, sp), "lmrl_add_i32")      # rows with a next token + 1 (0-token records: 1, all padding)
    return DevicePPODataset(longest=longest, lengths=lengths, **ds), kls[:n_act]


def _ev(torch):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def truncate_turns(rec: PPORecords, max_length: int, gamma: float) -> Tuple[PPORecords, Dict[str, int]]:
    """The task scripts' length rule (llm_rl_scripts/wordle/ppo/train_ppo_gpt2.py:323-341; the chess and maze PPO scripts carry the same loop) on
    single-trajectory chains in HBM (`lmrl_ppo_truncate_turns`): while a trajectory has more than three texts and at least `max_length` tokens its
    last (action, observation) pair is dropped, (the pair's rewards) x gamma go onto the previous action and `done` becomes False; trajectories
    left with fewer than three texts or still too long are skipped -> (records of the kept trajectories, {"shortened": .., "skipped": ..}).
    The input records are not modified (reward / n_tok / done are copied; tokens and flags are shared unless rows are skipped)."""
    import torch
    assert rec.chain is None, "the scripts apply this rule to chains of one trajectory"
    L, sp, dev = _lib.lib(), _lib.stream_ptr(), rec.tokens.device
    n = rec.n
    reward = rec.reward.clone()
    n_tok = torch.empty(n, dtype=torch.int32, device=dev)
    done, keep = torch.empty(n, dtype=torch.uint8, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)
    meta = torch.empty(2, dtype=torch.int32, device=dev)
    _lib.check(L.lmrl_ppo_truncate_turns(_lib.ptr(rec.is_action), _lib.ptr(rec.n_tok), _lib.ptr(rec.done), n, rec.cap, int(max_length), float(gamma),
                                         _lib.ptr(reward), _lib.ptr(n_tok), _lib.ptr(done), _lib.ptr(keep), _lib.ptr(meta), sp), "lmrl_ppo_truncate_turns")
    shortened, skipped = (int(x) for x in meta.cpu().numpy())
    info = dict(shortened=shortened, skipped=skipped)
    if skipped == 0:
        return PPORecords(rec.tokens, rec.is_action, reward, n_tok, done), info
    if skipped == n:
        raise ValueError("every trajectory is skipped by the length rule (fewer than three texts, or still >= max_length tokens)")
    idx, count = torch.empty(n, dtype=torch.int32, device=dev), torch.empty(1, dtype=torch.int32, device=dev)
    _lib.check(L.lmrl_compact_flags(_lib.ptr(keep), n, _lib.ptr(idx), _lib.ptr(count), sp), "lmrl_compact_flags")
    m = n - skipped

    def rows(src):
        src2 = src.view(n, -1)
        dst = torch.empty((m, src2.shape[1]), dtype=src.dtype, device=dev)
        _lib.check(L.lmrl_gather_rows_bytes(src2.data_ptr(), idx.data_ptr(), dst.data_ptr(), m, src2.shape[1] * src2.element_size(), sp), "lmrl_gather_rows_bytes")
        return dst if src.dim() == 2 else dst.view(m)
    return PPORecords(rows(rec.tokens), rows(rec.is_action), rows(reward), rows(n_tok), rows(done)), info


def masked_rows_device(should_take_action, attention_mask, input_ids, T: int):
    """`common.masked_rows` + the next-token targets of those rows for a batch that lives in HBM -> (idx int32 [Ra], targets int32 [Ra], Ra).
    One 4-byte readback (Ra sizes the LM-head product)."""
    import torch
    B = int(should_take_action.shape[0])
    dev = should_take_action.device
    cnt = torch.empty(B, dtype=torch.int32, device=dev)
    off = torch.empty(B + 1, dtype=torch.int32, device=dev)
    idx = torch.empty(B * (T - 1), dtype=torch.int32, device=dev)
    tgt = torch.empty(B * (T - 1), dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().lmrl_masked_rows(_lib.ptr(should_take_action), _lib.ptr(attention_mask), _lib.ptr(input_ids), B, T, _lib.ptr(cnt), _lib.ptr(off),
                                           _lib.ptr(idx), _lib.ptr(tgt), _lib.stream_ptr()), "lmrl_masked_rows")
    ra = int(off[B:].cpu().numpy()[0])
    return idx[:ra], tgt[:ra], ra


def mask_pos_device(input_ids, pad: int, shifted: bool = False):
    """`initialize_attn_mask_pos_ids` for device ids [B, T] -> (attention_mask uint8 [B, T], position_ids int32 [B, T]) and, with
    `shifted`, float32 attention_mask[:, 1:] [B, T-1] (the loss's mask operand)."""
    import torch
    B, T = input_ids.shape
    am = torch.empty(B, T, dtype=torch.uint8, device=input_ids.device)
    pos = torch.empty(B, T, dtype=torch.int32, device=input_ids.device)
    nxt = torch.empty(B, T - 1, dtype=torch.float32, device=input_ids.device) if shifted else None
    _lib.check(_lib.lib().lmrl_seq_mask_pos(_lib.ptr(input_ids), int(pad), _lib.ptr(am), _lib.ptr(pos), _lib.ptr(nxt), B, T, _lib.stream_ptr()), "lmrl_seq_mask_pos")
    return (am, pos, nxt) if shifted else (am, pos)
