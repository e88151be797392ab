"""The LLM_RL.environment protocol, re-implemented for the MI355X engine.

Same public names, signatures and semantics as the reference module
(`LLM_RL/environment.py`): `Text`, `TextHistory`, `TextTrajectory[Chain]`, `TextEnv`,
`BatchedTextEnv` and the two adapters, `TextPolicy` / `BatchedTextPolicy` and adapters,
`InteractionTransition`, `interact_environment`, `text_env_eval`, `TokenHistory`,
`TokenTrajectory`, `TokenTrajectoryChain` — so reference call sites work unchanged.

Differences (all additive):
  * an env may expose `as_batched()`; `interact_environment` then uses the env's own
    lock-step device implementation instead of deep-copying one Python object per slot
    (reference: `TextEnvToBatchedTextEnv.reset`, environment.py:92).
  * `interact_environment(initial_text_history=...)` works (the reference raises TypeError on the
    subscripted-generic isinstance at environment.py:172).
  * `TextPolicyToBatchedTextPolicy.act` does not print its inputs (environment.py:130-132).
"""
from __future__ import annotations

import copy as _copy
from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Any, Callable, Dict, Iterator, List, NamedTuple, Optional, Sequence, Tuple, Union

import numpy as np


# --------------------------------------------------------------------------- text records (environment.py:12-37)
@dataclass(frozen=True)
class Text:
    text: str
    is_action: bool


TextHistory = Tuple[Text, ...]
StepResult = Tuple[TextHistory, float, bool]


def text_history_to_str(text_history: TextHistory) -> str:
    return "".join(item.text for item in text_history)


@dataclass(frozen=True)
class TextTrajectory:
    """A trajectory that fits one context window (environment.py:23-31)."""
    text_history: TextHistory
    reward: Tuple[float, ...]
    done: bool

    def __post_init__(self):
        assert len(self.reward) == len(self.text_history), "reward is needed for each text"
        for r, t in zip(self.reward, self.text_history):
            assert t.is_action or r == 0.0, "reward for non-actions texts should be 0.0"


@dataclass(frozen=True)
class TextTrajectoryChain:
    """Linked list of trajectories, one per context window (environment.py:34-37)."""
    text_trajectory: TextTrajectory
    next: Optional["TextTrajectoryChain"]


# --------------------------------------------------------------------------- env protocol (environment.py:41-111)
class TextEnv(ABC):
    @abstractmethod
    def step(self, text_history: TextHistory) -> Tuple[TextHistory, float, bool]:
        ...

    @abstractmethod
    def reset(self, seed: Optional[int] = None, options: Optional[Dict] = None) -> TextHistory:
        ...

    def close(self) -> None:
        pass

    def copy(self) -> "TextEnv":
        return _copy.deepcopy(self)


class BatchedTextEnv(ABC):
    @abstractmethod
    def step(self, text_history: List[Optional[TextHistory]], done: Optional[List[bool]] = None
             ) -> List[Optional[Tuple[TextHistory, float, bool]]]:
        ...

    @abstractmethod
    def reset(self, seed: Optional[List[Optional[int]]] = None, options: Optional[List[Optional[Dict]]] = None
              ) -> List[TextHistory]:
        ...

    def close(self) -> None:
        pass

    def copy(self) -> "BatchedTextEnv":
        return _copy.deepcopy(self)


def _broadcast_seed_options(seed, options):
    """Shared normalisation of (seed, options) lists (environment.py:85-91)."""
    if seed is None and options is None:
        return [None], [None]
    if seed is None:
        seed = [None] * len(options)
    elif options is None:
        options = [None] * len(seed)
    assert len(seed) == len(options)
    return list(seed), list(options)


class TextEnvToBatchedTextEnv(BatchedTextEnv):
    """Generic adapter: one deep copy of `env` per slot, stepped in a Python loop."""

    def __init__(self, env: TextEnv):
        self.env = env
        self.batch_env_copies: Optional[List[TextEnv]] = None

    def step(self, text_history, done=None):
        assert self.batch_env_copies is not None, "reset must be called before step"
        assert len(text_history) == len(self.batch_env_copies), \
            "batch size must be the same as the number of environments initalized"
        if done is None:
            done = [False] * len(text_history)
        assert len(text_history) == len(done)
        out = []
        for e, item, d in zip(self.batch_env_copies, text_history, done):
            out.append(None if d else e.step(item))
        return out

    def reset(self, seed=None, options=None):
        seed, options = _broadcast_seed_options(seed, options)
        self.batch_env_copies = [self.env.copy() for _ in seed]
        return [e.reset(seed=s, options=o) for e, s, o in zip(self.batch_env_copies, seed, options)]

    def close(self) -> None:
        for e in self.batch_env_copies or []:
            e.close()
        self.env.close()


class BatchedTextEnvToTextEnv(TextEnv):
    def __init__(self, env: BatchedTextEnv):
        self.env = env

    def step(self, text_history):
        return self.env.step([text_history])[0]

    def reset(self, seed=None, options=None):
        return self.env.reset(seed=[seed], options=[options])[0]

    def close(self) -> None:
        self.env.close()


# --------------------------------------------------------------------------- policy protocol (environment.py:115-143)
class TextPolicy(ABC):
    @abstractmethod
    def act(self, text_history: TextHistory) -> TextHistory:
        ...


class BatchedTextPolicy(ABC):
    @abstractmethod
    def act(self, text_history: List[Optional[TextHistory]], done: Optional[List[bool]] = None
            ) -> List[Optional[TextHistory]]:
        ...


class TextPolicyToBatchedTextPolicy(BatchedTextPolicy):
    def __init__(self, policy: TextPolicy):
        self.policy = policy

    def act(self, text_history, done=None):
        if done is None:
            done = [False] * len(text_history)
        assert len(text_history) == len(done)
        return [None if d else self.policy.act(item) for item, d in zip(text_history, done)]


class BatchedTextPolicyToTextPolicy(TextPolicy):
    def __init__(self, policy: BatchedTextPolicy):
        self.policy = policy

    def act(self, text_history):
        return self.policy.act([text_history])[0]


# --------------------------------------------------------------------------- rollout driver (environment.py:147-267)
class InteractionTransition(NamedTuple):
    pre_action_history: TextHistory
    post_action_history: TextHistory
    post_transition_history: TextHistory
    reward: float
    done: bool


def _is_text_history(x) -> bool:
    return isinstance(x, tuple) and all(isinstance(t, Text) for t in x)


def as_batched_env(env: Union[TextEnv, BatchedTextEnv]) -> BatchedTextEnv:
    if isinstance(env, BatchedTextEnv):
        return env
    if hasattr(env, "as_batched"):
        return env.as_batched()
    return TextEnvToBatchedTextEnv(env)


def interact_environment(
    env: Union[TextEnv, BatchedTextEnv],
    policy: Union[TextPolicy, BatchedTextPolicy],
    initial_text_history: Optional[Union[TextHistory, List[TextHistory]]] = None,
    env_seed: Union[Optional[int], Optional[List[Optional[int]]]] = None,
    env_options: Union[Optional[Dict], Optional[List[Optional[int]]]] = None,
    bsize: int = 1,
    npad: int = 0,
) -> List[List[InteractionTransition]]:
    """Lock-step batched rollout until every slot is done (environment.py:154-207).

    `npad` dummy slots (history `(Text("", False),)`, done=True) are appended for the policy only, so a
    fixed-shape device policy always sees `bsize + npad` rows (environment.py:182).
    """
    assert bsize > 0
    env = as_batched_env(env)
    if isinstance(policy, TextPolicy):
        policy = TextPolicyToBatchedTextPolicy(policy)
    if isinstance(env_seed, int):
        env_seed = [env_seed] * bsize
    if isinstance(env_options, dict):
        env_options = [env_options] * bsize
    if initial_text_history is not None and _is_text_history(initial_text_history):
        initial_text_history = [initial_text_history] * bsize
    text_history = initial_text_history
    if text_history is None:
        text_history = env.reset(env_seed, env_options)
    text_history = list(text_history)

    pad_rows = [(Text("", is_action=False),)] * npad
    transitions: List[List[InteractionTransition]] = [[] for _ in range(bsize)]
    done = [False] * bsize
    while not all(done):
        pre = text_history
        acted = policy.act(list(text_history) + pad_rows, done=done + [True] * npad)
        post_action = list(acted[:bsize])
        results = env.step(post_action, done=done)
        post_transition, reward, new_done = [], [], []
        for res in results:
            h, r, d = (None, None, True) if res is None else res
            post_transition.append(h); reward.append(r); new_done.append(d)
        for i in range(bsize):
            incomplete = (pre[i] is None or post_action[i] is None or post_transition[i] is None or reward[i] is None)
            if new_done[i] and incomplete:   # slot was already finished before this step
                continue
            transitions[i].append(InteractionTransition(pre[i], post_action[i], post_transition[i], reward[i], new_done[i]))
        text_history, done = post_transition, new_done
    return transitions


def text_env_eval(
    env: Union[TextEnv, BatchedTextEnv],
    policy: Union[TextPolicy, BatchedTextPolicy],
    n_rollouts: int,
    initial_text_history: Optional[TextHistory] = None,
    seed_generator: Optional[Iterator[int]] = None,
    env_options: Optional[Dict] = None,
    interaction_callback: Optional[Callable[[List[InteractionTransition]], None]] = None,
    bsize: int = 1,
    verbose: bool = True,
) -> Tuple[List[List[InteractionTransition]], Dict[str, Any]]:
    """ceil(n/bsize) batched rollouts + reward/done/length summary (environment.py:211-267)."""
    benv = as_batched_env(env)
    interactions, rewards, dones, lengths = [], [], [], []
    n_batches = (n_rollouts + bsize - 1) // bsize
    it = range(n_batches)
    if verbose:
        try:
            from tqdm.auto import tqdm
            it = tqdm(it)
        except Exception:  # tqdm is cosmetic
            pass
    for _ in it:
        actual = min(n_rollouts - len(interactions), bsize)
        seeds = [None] * actual if seed_generator is None else [next(seed_generator) for _ in range(actual)]
        batch = interact_environment(
            benv, policy, initial_text_history=initial_text_history, env_seed=seeds,
            env_options=[env_options] * actual, bsize=actual, npad=bsize - actual)
        for episode in batch:
            interactions.append(episode)
            rewards.append(sum(tr.reward for tr in episode))
            dones.append(episode[-1].done)
            lengths.append(len(episode))
            if interaction_callback is not None:
                interaction_callback(episode)

    def _summary(x):
        return dict(mean=np.mean(x), std=np.std(x), min=np.min(x), max=np.max(x))

    rewards = np.asarray(rewards, dtype=np.float32)
    dones = np.asarray(dones, dtype=np.float32)
    return interactions, dict(reward=_summary(rewards), done=_summary(dones), length=_summary(lengths))


class UserPolicy(TextPolicy):
    """Interactive policy reading actions from stdin (environment.py:271-288)."""

    def __init__(self, initial_str: str, postproc_print_f: Optional[Callable[[str], str]] = None,
                 postproc_action_f: Optional[Callable[[str], str]] = None):
        self.initial_str = initial_str
        self.postproc_print_f = postproc_print_f or (lambda x: x)
        self.postproc_action_f = postproc_action_f or (lambda x: x)

    def act(self, text_history: TextHistory) -> TextHistory:
        bar = "=" * 25
        print(bar); print(self.postproc_print_f(text_history_to_str(text_history))); print(bar)
        response = self.initial_str + input(self.initial_str)
        return tuple(text_history) + (Text(self.postproc_action_f(response), True),)


# --------------------------------------------------------------------------- tokenised containers (environment.py:294-419)
def _encode(tokenizer, text: str, token_process) -> List[int]:
    ids = tokenizer.encode(text)
    return list(token_process(ids)) if token_process is not None else list(ids)


@dataclass(frozen=True)
class TokenHistory:
    tokens: np.ndarray     # 1d int32
    is_action: np.ndarray  # 1d bool

    def __post_init__(self):
        assert self.tokens.ndim == 1 and self.is_action.ndim == 1, "(tokens, is_action) must be 1 dimensional"
        assert self.tokens.shape == self.is_action.shape, "(tokens, is_action) must have the same shape"

    @classmethod
    def from_text_history(cls, text_history: TextHistory, tokenizer, token_process=None) -> "TokenHistory":
        toks: List[int] = []
        act: List[bool] = []
        for item in text_history:
            ids = _encode(tokenizer, item.text, token_process)
            toks += ids
            act += [item.is_action] * len(ids)
        return cls(np.array(toks, dtype=np.int32), np.array(act, dtype=np.bool_))


@dataclass(frozen=True)
class TokenTrajectory:
    tokens: np.ndarray     # 1d int32
    is_action: np.ndarray  # 1d bool
    reward: np.ndarray     # 1d float32, the Text's reward sits on its LAST token (environment.py:370)
    done: np.ndarray       # bool scalar

    def __post_init__(self):
        assert self.tokens.ndim == 1, "tokens must be 1 dimensional"
        assert self.is_action.ndim == 1, "is_action must be 1 dimensional"
        assert self.reward.ndim == 1, "reward must be 1 dimensional"
        assert self.done.ndim == 0, "done must be scalar"
        assert self.is_action.shape == self.tokens.shape, "is_action must have the same shape as tokens"
        assert self.reward.shape == self.tokens.shape, "reward must have the same shape as tokens"
        assert not np.any(self.reward[~self.is_action.astype(bool)] != 0.0), "reward must be 0.0 if not an action"

    @classmethod
    def from_text_trajectory(cls, text_trajectory: TextTrajectory, tokenizer, token_process=None) -> "TokenTrajectory":
        toks: List[int] = []
        act: List[bool] = []
        rew: List[float] = []
        for item, r in zip(text_trajectory.text_history, text_trajectory.reward):
            ids = _encode(tokenizer, item.text, token_process)
            toks += ids
            act += [item.is_action] * len(ids)
            rew += [0.0] * (len(ids) - 1) + [r]
        return cls(np.array(toks, dtype=np.int32), np.array(act, dtype=np.bool_), np.array(rew, dtype=np.float32),
                   np.array(text_trajectory.done, dtype=np.bool_))


@dataclass(frozen=True)
class TokenTrajectoryChain:
    token_trajectory: TokenTrajectory
    next: Optional["TokenTrajectoryChain"]

    def __post_init__(self):
        node, dones = self, []
        while node.next is not None:
            dones.append(node.token_trajectory.done)
            node = node.next
        assert not np.any(dones[:-1]), "token trajectory chain can only be done at the end"

    def to_list(self) -> List[TokenTrajectory]:
        out, node = [], self
        while node is not None:
            out.append(node.token_trajectory)
            node = node.next
        return out

    @classmethod
    def from_text_trajectory_chain(cls, text_trajectory_chain: TextTrajectoryChain, tokenizer, token_process=None
                                   ) -> "TokenTrajectoryChain":
        nxt = None
        if text_trajectory_chain.next is not None:
            nxt = cls.from_text_trajectory_chain(text_trajectory_chain.next, tokenizer, token_process=token_process)
        head = TokenTrajectory.from_text_trajectory(text_trajectory_chain.text_trajectory, tokenizer, token_process=token_process)
        return cls(head, nxt)
