"""PPO inference + rollout -> PPOData pipeline — counterpart of `PPOInference` / `GPT2PPOInference`
(LLM_RL/algorithms/ppo/base_interface.py:345-799, LLM_RL/algorithms/ppo/gpt2/interface.py:214-466).

`forward` runs the policy, the frozen initial policy and the value head in float32 on the HIP train kernels;
`get_ppo_data_from_token_trajectory_chain` restates base_interface.py:464-669: token log-probs of both policies, values
with the bootstrap slot, KL-penalised rewards, GAE over action tokens (`lmrl_gae`), whitening over the whole batch
(`lmrl_whiten_*`, all-reduced across ranks) and the scatter back into per-chunk `PPOData`.
"""
from __future__ import annotations

from typing import List, NamedTuple, Optional, Tuple

import numpy as np

from .. import _lib
from .. import dist as D
from ..train import ops
from ..train.gpt2_f32 import GPT2F32, LinearHeadF32
from .common import BlockingStrategy, Padding, Truncation, block_sequences, initialize_attn_mask_pos_ids
from .ppo import PPOData, _t, gae_from_chains


def text_trajectory_chains_from_interactions(raw_results, tokenizer, max_length: int, gamma: float):
    """The rollout -> chain step of the task scripts' `ppo_dataset_loader` (llm_rl_scripts/wordle/ppo/train_ppo_gpt2.py:
    310-342): per episode one TextTrajectory with reward [0, r1, 0, r2, 0, ...]; while its tokenisation reaches `max_length`
    the last (action, observation) pair is cut, its discounted reward folded into the previous action and `done` cleared;
    episodes shorter than 3 texts or still too long are skipped."""
    from ..environment import TextTrajectory, TextTrajectoryChain, TokenTrajectory
    chains = []
    for raw in raw_results:
        hist = tuple(raw[-1].post_transition_history)
        reward = sum([[it.reward, 0.0] for it in raw], [0.0])
        done = raw[-1].done
        n_tok = lambda h, r, d: TokenTrajectory.from_text_trajectory(TextTrajectory(h, tuple(r), d), tokenizer).tokens.shape[0]
        while len(hist) > 3 and n_tok(hist, reward, done) >= max_length:
            new_reward = list(reward[:-2])
            new_reward[-2] += sum(reward[-2:]) * gamma
            hist, reward, done = hist[:-2], new_reward, False
        if len(hist) < 3 or n_tok(hist, reward, done) >= max_length:
            continue
        chains.append(TextTrajectoryChain(TextTrajectory(hist, tuple(reward), done), None))
    return chains


def text_trajectory_chains_from_transitions(raw_results):
    """The rollout -> chain step for envs whose observation is only the CURRENT state (chess: llm_rl_scripts/chess/ppo/train_ppo_gpt2_online.py:
    293-314): one TextTrajectory per transition — `post_action_history` with reward [0, r] — linked through `next` in episode order."""
    from ..environment import TextTrajectory, TextTrajectoryChain
    chains = []
    for raw in raw_results:
        chain = None
        for tr in reversed(list(raw)):
            hist = tuple(tr.post_action_history)
            chain = TextTrajectoryChain(TextTrajectory(hist, (0.0,) * (len(hist) - 1) + (float(tr.reward),), bool(tr.done)), chain)
        if chain is not None:
            chains.append(chain)
    return chains


def text_trajectory_chains_partially_observed(raw_results):
    """The rollout -> chain step of the partially observed Maze script (llm_rl_scripts/maze/ppo/partially_observed_ppo_online.py:372-398): per
    transition the texts of `post_action_history[:-1]` (the item window the policy saw) joined by single spaces into ONE non-action Text, then the
    action, reward [0, r]; linked through `next` in episode order."""
    from ..environment import Text, TextTrajectory, TextTrajectoryChain
    chains = []
    for raw in raw_results:
        chain = None
        for tr in reversed(list(raw)):
            state = Text(" ".join(item.text for item in tr.post_action_history[:-1]), False)
            chain = TextTrajectoryChain(TextTrajectory((state, tr.post_action_history[-1]), (0.0, float(tr.reward)), bool(tr.done)), chain)
        if chain is not None:
            chains.append(chain)
    return chains


class PPOForwardOutput(NamedTuple):
    initial_policy_logprobs: Optional[np.ndarray]   # [B, T-1] log p_init(ids[t+1] | ids[:t+1])
    policy_logprobs: np.ndarray                     # [B, T-1]
    values: np.ndarray                              # [B, T]


def unpad_array(xs: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """LLM_RL/utils.py:33-38: cut at the first pad."""
    pad_t = np.where(1 - np.asarray(mask).astype(np.int32))[0]
    return xs[: pad_t[0]] if len(pad_t) > 0 else xs


class CombinedTokenTrajectoryChain(NamedTuple):
    """base_interface.py:295-343."""
    input_tokens: np.ndarray
    output_tokens: np.ndarray
    rewards: np.ndarray
    should_take_action: np.ndarray
    done: bool
    chunk_lens: List[int]

    @classmethod
    def from_token_trajectory_chain(cls, chain, max_length: Optional[int] = None) -> "CombinedTokenTrajectoryChain":
        tts = chain.to_list()
        assert len(tts) > 0, "token_trajectory_chain must have at least one token_trajectory"
        if max_length is None:
            max_length = max(tt.tokens.shape[0] for tt in tts) + 1
        assert not any(tt.done for tt in tts[:-1]), "done can only be true at the end of the chain"
        for i, tt in enumerate(tts):
            no_trunc = (tt.tokens.shape[0] - 1) <= max_length
            ends_with_state = not np.any(tt.is_action[1:][max_length:])
            next_starts_with_action = i < len(tts) - 1 and bool(tts[i + 1].is_action[0])
            assert not (ends_with_state and next_starts_with_action), "trajectory truncation error"
            assert no_trunc or ends_with_state, "trajectory truncation error"
        cat = lambda f: np.concatenate([f(tt)[:max_length] for tt in tts], axis=0)
        return cls(input_tokens=cat(lambda tt: tt.tokens[:-1]), output_tokens=cat(lambda tt: tt.tokens[1:]),
                   rewards=cat(lambda tt: tt.reward[1:]), should_take_action=cat(lambda tt: tt.is_action[1:]),
                   done=bool(tts[-1].done), chunk_lens=[min(tt.tokens.shape[0] - 1, max_length) for tt in tts])

    def unroll_arr(self, arr: np.ndarray) -> List[np.ndarray]:
        assert arr.shape[0] == self.input_tokens.shape[0]
        return np.split(arr, np.cumsum(self.chunk_lens)[:-1], axis=0)


class GPT2PPOInference:
    def __init__(self, policy: GPT2F32, value_head: LinearHeadF32, pad_token_id: int, initial_policy: Optional[GPT2F32] = None,
                 tokenizer=None, loss_kwargs: Optional[dict] = None, bc_loss_weight: float = 0.0):
        self.policy, self.value_head, self.initial_policy, self.pad = policy, value_head, initial_policy, pad_token_id
        self.tokenizer, self.loss_kwargs, self.bc_loss_weight = tokenizer, dict(loss_kwargs or {}), bc_loss_weight

    def forward_from_str(self, input_strs: List[str], blocking_strategy: BlockingStrategy = BlockingStrategy(Padding.RIGHT, Truncation.RIGHT, None),
                         token_process=None) -> PPOForwardOutput:
        """base_interface.py:437-462."""
        tp = token_process or (lambda x: x)
        tokens = block_sequences([tp(list(self.tokenizer.encode(s))) for s in input_strs], self.pad, np.int32, blocking_strategy)
        return self.forward(tokens)

    def get_ppo_data_from_text_trajectory_chain(self, text_trajectory_chains, bsize: int, max_length: Optional[int] = None, token_process=None,
                                                **kw) -> Tuple[List[PPOData], np.ndarray]:
        """base_interface.py:671-708: tokenise the chains, then the token-level pipeline."""
        from ..environment import TokenTrajectoryChain
        chains = [TokenTrajectoryChain.from_text_trajectory_chain(c, self.tokenizer, token_process=token_process) for c in text_trajectory_chains]
        return self.get_ppo_data_from_token_trajectory_chain(chains, bsize=bsize, max_length=max_length, **kw)

    def eval_loss(self, input_ids, should_take_action, old_logprobs, old_values, old_advantages, old_returns, attention_mask=None,
                  position_ids=None, prng_key=None, bc_data_input_ids=None, bc_data_input_attention_mask=None, bc_data_input_position_ids=None,
                  bc_data_input_training_mask=None, train: bool = False):
        """base_interface.py:747-799: the train step's loss / log dict without the update (returns (loss, logs))."""
        from .ppo import GPT2PPOTrain
        ev = GPT2PPOTrain.__new__(GPT2PPOTrain)            # loss-only view on the same weights: no optimizer state is created
        ev.policy, ev.value_head, ev.pad, ev.loss_kwargs, ev.bc_loss_weight = self.policy, self.value_head, self.pad, self.loss_kwargs, self.bc_loss_weight
        _, loss, logs = ev.step(input_ids, should_take_action, old_logprobs, old_values, old_advantages, old_returns, attention_mask=attention_mask,
                                position_ids=position_ids, bc_data_input_ids=bc_data_input_ids,
                                bc_data_input_attention_mask=bc_data_input_attention_mask, bc_data_input_position_ids=bc_data_input_position_ids,
                                bc_data_input_training_mask=bc_data_input_training_mask, train=False)
        return loss, logs

    def _logprobs(self, model: GPT2F32, ids_d, am_d, pos_d, B, T):
        import torch
        hid, _ = model.forward(ids_d, am_d, pos_d)
        logits = model.lm_logits(hid, B * T)
        tgt = torch.zeros(B * T, dtype=torch.int32, device=model.dev)
        tgt.view(B, T)[:, :-1] = ids_d[:, 1:]
        lp = torch.empty(B * T, dtype=torch.float32, device=model.dev)
        ops.lse_gather(logits, model.ld_vocab, model.vocab, tgt, B * T, logprob=lp)   # token_logprobs_from_logits, :396-403
        return hid, lp.view(B, T)[:, :-1].cpu().numpy()

    def forward(self, input_ids, attention_mask=None, position_ids=None) -> PPOForwardOutput:
        ids = np.asarray(input_ids, dtype=np.int32)
        am, pos = initialize_attn_mask_pos_ids(ids, self.pad, attention_mask, position_ids)
        B, T = ids.shape
        ids_d, am_d, pos_d = _t(ids, np.int32), _t(am, np.uint8), _t(pos, np.int32)
        init_lp = None
        if self.initial_policy is not None:
            _, init_lp = self._logprobs(self.initial_policy, ids_d, am_d, pos_d, B, T)
        hid, lp = self._logprobs(self.policy, ids_d, am_d, pos_d, B, T)
        values, _ = self.value_head.forward(hid, B * T)
        return PPOForwardOutput(init_lp, lp, values.view(B, T).cpu().numpy())

    def get_ppo_data_from_token_trajectory_chain(self, token_trajectory_chains, bsize: int, max_length: Optional[int] = None, *,
                                                 gamma: float, lam: float, kl_weight: float, use_advantage_whitening: bool = True
                                                 ) -> Tuple[List[PPOData], np.ndarray]:
        assert self.initial_policy is not None
        n_chains = len(token_trajectory_chains)
        combos = [CombinedTokenTrajectoryChain.from_token_trajectory_chain(c, max_length=max_length - 1 if max_length is not None else None)
                  for c in token_trajectory_chains]
        toks = [tt.tokens for c in token_trajectory_chains for tt in c.to_list()]
        tokens = block_sequences(toks, self.pad, np.int32, BlockingStrategy(Padding.RIGHT, Truncation.RIGHT, max_length))
        init_lps, lps, vals = [], [], []
        for i in range(0, len(tokens), bsize):
            out = self.forward(tokens[i:i + bsize])
            init_lps.append(out.initial_policy_logprobs); lps.append(out.policy_logprobs); vals.append(out.values)
        init_lps, lps, vals = np.concatenate(init_lps), np.concatenate(lps), np.concatenate(vals)
        sections = np.cumsum([len(c.chunk_lens) for c in combos])[:-1]
        mask_by_chain = np.split(tokens != self.pad, sections, axis=0)
        per_chain = lambda arr, sl: [np.concatenate([sl(unpad_array(x, m)) for x, m in zip(item, mask)], axis=0)
                                     for mask, item in zip(mask_by_chain, np.split(arr, sections, axis=0))]
        init_lp_chains = [np.concatenate([unpad_array(x, m) for x, m in zip(item, mask[:, 1:])]) for mask, item in
                          zip(mask_by_chain, np.split(init_lps, sections, axis=0))]
        lp_chains = [np.concatenate([unpad_array(x, m) for x, m in zip(item, mask[:, 1:])]) for mask, item in
                     zip(mask_by_chain, np.split(lps, sections, axis=0))]
        values_by_chain = np.split(vals, sections, axis=0)
        values_chains = [np.concatenate([unpad_array(x, m)[:-1] for x, m in zip(item, mask)]) for mask, item in zip(mask_by_chain, values_by_chain)]
        last_values = [unpad_array(item[-1], mask[-1])[-1] for mask, item in zip(mask_by_chain, values_by_chain)]
        values_chains = [np.concatenate((v, np.asarray([lv * (1.0 - float(c.done))], dtype=v.dtype)))       # bootstrap slot, :566-570
                         for v, lv, c in zip(values_chains, last_values, combos)]
        log_ratio = [(p - q) * c.should_take_action.astype(np.float32) for q, p, c in zip(init_lp_chains, lp_chains, combos)]
        valid = np.argwhere(np.concatenate([c.should_take_action.astype(np.float32).reshape(-1) for c in combos]))[:, 0]
        all_lr = np.concatenate([x.reshape(-1) for x in log_ratio])[valid]
        all_kls = np.exp(all_lr) - 1 - all_lr
        rewards = [c.rewards - kl_weight * lr for c, lr in zip(combos, log_ratio)]                               # :580-584
        # ---- GAE on the device: one row per chain, values row = [per-token values..., bootstrap]
        Lmax = max(len(r) for r in rewards)
        Vb = np.zeros((n_chains, Lmax + 1), np.float32); Rb = np.zeros((n_chains, Lmax), np.float32)
        Sb = np.zeros((n_chains, Lmax), np.uint8); lens = np.zeros(n_chains, np.int32)
        for i, (v, r, c) in enumerate(zip(values_chains, rewards, combos)):
            L = len(r)
            Vb[i, : L + 1], Rb[i, :L], Sb[i, :L], lens[i] = v, r, c.should_take_action, L
        adv, ret = gae_from_chains(Vb, Rb, Sb, lens, gamma, lam)
        if use_advantage_whitening:                                                                              # :609-615
            import torch
            x = _t(adv.reshape(-1), np.float32)
            y = D.whiten_distributed(x, _t(Sb.reshape(-1), np.uint8), shift_mean=True)
            adv = y.cpu().numpy().reshape(adv.shape)
        ppo_datas: List[PPOData] = []
        for i, (chain, c) in enumerate(zip(token_trajectory_chains, combos)):
            L = lens[i]
            ids_chunks = [tt.tokens[:max_length] for tt in chain.to_list()]
            parts = [c.unroll_arr(a) for a in (c.should_take_action, lp_chains[i], values_chains[i][:-1], adv[i, :L], ret[i, :L])]
            for k in range(len(c.chunk_lens)):
                ppo_datas.append(PPOData(input_ids=ids_chunks[k], should_take_action=parts[0][k], old_logprobs=parts[1][k],
                                         old_values=parts[2][k], old_advantages=parts[3][k], old_returns=parts[4][k]))
        return ppo_datas, all_kls
