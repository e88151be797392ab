"""Data formats either side of the hot path.

* the jsonl trajectory format of `llm_rl_scripts/wordle/misc/data_gen.py:53-64`
  (`{"sequence": [[text, is_action], ...], "reward": [...], "done": bool}`) — reader, writer and the item -> chain map of
  `llm_rl_scripts/wordle/ilql/train_ilql_gpt2.py:123-134`;
* builders for the blocked ILQL / MC / BC-mask datasets the train scripts feed (`train_ilql_gpt2.py:136-152`,
  `train_ppo_gpt2.py:134-142`);
* `dataloader` with the JaxSeq call shape `dataloader(rng, dataset, bsize, truncate=True)` (permutation order comes from a
  numpy generator: the reference's order comes from `jax.random.permutation` in un-vendored JaxSeq — parity unpinned);
* `generate_wordle_dataset`: the offline data generator of `wordle/misc/data_gen.py` (PolicyDataGenerator +
  RandomMixturePolicy, `wordle/env/data.py:9-37`, `scripted_policies.py:114-126`) with the env transitions on the batched
  HIP env instead of a multiprocessing pool of Python envs;
* `WordleTokenizer`: a tokenizer-shaped adapter over `WordleTokenTable` for Wordle-alphabet text (no GPT-2 tokenizer files
  exist offline; with network access pass a real `transformers` tokenizer instead).
"""
from __future__ import annotations

import json
from typing import Any, Dict, Iterable, Iterator, List, Optional, Sequence

import numpy as np

from .environment import Text, TextTrajectory, TextTrajectoryChain, TokenTrajectoryChain


# ---------------------------------------------------------------------------------------------- jsonl
def jsonl_stream(path: str) -> Iterator[Dict[str, Any]]:
    with open(path, "r") as f:
        for line in f:
            line = line.strip()
            if line:
                yield json.loads(line)


def write_jsonl(path: str, items: Iterable[Dict[str, Any]]) -> int:
    n = 0
    with open(path, "w") as f:
        for it in items:
            f.write(json.dumps(it) + "\n")
            n += 1
    return n


def item_to_text_chain(item: Dict[str, Any]) -> TextTrajectoryChain:
    """train_ilql_gpt2.py:123-134: reward gets a leading 0.0 for the 'Wordle:\\n' header element."""
    return TextTrajectoryChain(
        text_trajectory=TextTrajectory(text_history=tuple(Text(text, bool(is_action)) for text, is_action in item["sequence"]),
                                       reward=tuple([0.0] + list(item["reward"])), done=bool(item["done"])),
        next=None)


def text_trajectory_to_item(text_history, reward, done) -> Dict[str, Any]:
    """data_gen.py:56-62 (the caller passes the reformatted history; `reward` excludes the header element)."""
    return dict(sequence=[(t.text, float(t.is_action)) for t in text_history], reward=[float(r) for r in reward], done=bool(done))


def token_chains_from_jsonl(path: str, tokenizer) -> Iterator[TokenTrajectoryChain]:
    for item in jsonl_stream(path):
        yield TokenTrajectoryChain.from_text_trajectory_chain(item_to_text_chain(item), tokenizer)


def ilql_dataset_from_jsonl(path: str, tokenizer, blocking_strategy):
    from .algorithms.ilql import ILQLData, ILQLDataset
    return ILQLDataset.from_ilql_data_list([ILQLData.from_token_trajectory_chain(c) for c in token_chains_from_jsonl(path, tokenizer)],
                                           tokenizer, blocking_strategy)


def mc_data_from_jsonl(path: str, tokenizer, gamma: float):
    from .algorithms.mc_returns import MCData
    return [MCData.from_token_trajectory_chain(c, gamma) for c in token_chains_from_jsonl(path, tokenizer)]


class MaskDataset:
    """Blocked (ids, training-mask) pairs: what `MaskIterableDataset.blocked_from_str_segments_iterable` yields for the PPO BC
    batch and the BC trainer (`train_ppo_gpt2.py:134-142`): per segment `(text, weight)`, every token of the segment carries
    the segment's weight (is_action as float)."""

    def __init__(self, input_ids: np.ndarray, input_training_mask: np.ndarray):
        assert input_ids.shape == input_training_mask.shape
        self.input_ids, self.input_training_mask = input_ids, input_training_mask

    def __len__(self):
        return self.input_ids.shape[0]

    def __getitem__(self, i):
        return dict(input_ids=self.input_ids[i], input_training_mask=self.input_training_mask[i])

    @classmethod
    def blocked_from_str_segments(cls, segments_list: Sequence[Sequence], tokenizer, blocking_strategy) -> "MaskDataset":
        from .algorithms.common import block_sequences
        ids, masks = [], []
        for segs in segments_list:
            t, m = [], []
            for text, w in segs:
                e = list(tokenizer.encode(text))
                t += e
                m += [float(w)] * len(e)
            ids.append(t); masks.append(m)
        return cls(block_sequences(ids, tokenizer.pad_token_id, np.int32, blocking_strategy),
                   block_sequences(masks, 0.0, np.float32, blocking_strategy))

    @classmethod
    def from_jsonl(cls, path: str, tokenizer, blocking_strategy) -> "MaskDataset":
        return cls.blocked_from_str_segments([it["sequence"] for it in jsonl_stream(path)], tokenizer, blocking_strategy)


# ---------------------------------------------------------------------------------------------- dataloader
def dataloader(rng: Optional[np.random.Generator], dataset, bsize: int, truncate: bool = True) -> Iterator[Dict[str, np.ndarray]]:
    """One epoch of batches (dict of stacked arrays).  `rng=None` keeps file order; `truncate` drops the ragged last batch
    (the train loops call it with truncate=True: `ppo/train.py:264`, `ilql/train.py`)."""
    n = len(dataset)
    order = np.arange(n) if rng is None else rng.permutation(n)
    stop = n - (n % bsize) if truncate else n
    for i in range(0, stop, bsize):
        idx = order[i:i + bsize]
        items = [dataset[int(j)] for j in idx]
        yield {k: (None if items[0][k] is None else np.stack([np.asarray(it[k]) for it in items])) for k in items[0]}


# ---------------------------------------------------------------------------------------------- tokenizer adapter
class WordleTokenizer:
    """encode/decode for Wordle-alphabet text on the ids of a `WordleTokenTable`."""

    def __init__(self, table=None, pad_token_id: Optional[int] = None):
        from .rollout import WordleTokenTable
        self.table = table or WordleTokenTable.default_gpt2()
        self.pad_token_id = self.table.pad if pad_token_id is None else pad_token_id
        self.eos_token_id = self.table.newline

    def encode(self, s: str) -> List[int]:
        try:
            return self.table.encode_text(s)
        except AssertionError:
            return self._encode_lenient(s)

    def _encode_lenient(self, s: str) -> List[int]:
        """Text outside the canonical Wordle forms (a policy can emit anything): letters keep their first/space-prefixed
        ids, newlines are kept, every other character is dropped."""
        out: List[int] = []
        prev = "\n"
        for ch in s:
            if ch == "\n":
                out.append(self.table.newline)
            elif "a" <= ch <= "z":
                out.append(self.table.letter_sp[ord(ch) - 97] if prev == " " else self.table.letter_first[ord(ch) - 97])
            prev = ch
        return out

    def decode(self, ids) -> str:
        return "".join(self.table.strings.get(int(i), "") for i in ids if int(i) != self.pad_token_id)

    def __len__(self):
        return max(self.table.strings) + 1


class ByteTokenizer:
    """One id per UTF-8 byte (0..255), pad = 256: a stand-in for arbitrary text (Maze observations) when no GPT-2 tokenizer
    files are available; model vocabularies >= 257 work with it."""
    pad_token_id, eos_token_id = 256, 10

    def encode(self, s: str) -> List[int]:
        return list(s.encode("utf-8"))

    def decode(self, ids) -> str:
        return bytes(int(i) for i in ids if 0 <= int(i) < 256).decode("utf-8", errors="ignore")

    def __len__(self):
        return 257


# ---------------------------------------------------------------------------------------------- offline data generation
def _filtered_mask(words5: np.ndarray, trits: np.ndarray) -> np.ndarray:
    """bool [N, V]: `WordleState.word_in_state` (game.py:53-80) for every (env, word).  words5 int [V,5] letters 0..25,
    trits uint8 [N,26,5] with 0 NOT_HERE / 1 POSSIBLE / 2 HERE."""
    N, V = trits.shape[0], words5.shape[0]
    ok = np.ones((N, V), dtype=bool)
    has = np.zeros((V, 26), dtype=bool)
    has[np.arange(V)[:, None], words5] = True
    eq = words5[:, None, :] == np.arange(26)[None, :, None]          # [V,26,5]: word[i] == c
    for c in range(26):
        tc = trits[:, c, :]                                           # [N,5]
        all_possible = (tc == 1).all(axis=1)
        all_not = (tc == 0).all(axis=1)
        here_ok = ~((tc[:, None, :] == 2) & ~eq[None, :, c, :]).any(axis=2)       # HERE  => word[i] == c
        not_ok = ~((tc[:, None, :] == 0) & eq[None, :, c, :]).any(axis=2)         # NOT_HERE => word[i] != c
        mixed = here_ok & not_ok & has[None, :, c]
        sat = np.where(all_possible[:, None], True, np.where(all_not[:, None], ~has[None, :, c], mixed))
        ok &= sat
    return ok


def generate_wordle_dataset(vocab, n_data: int, prob_smart: float, seed: int = 0, bsize: int = 1024,
                            require_words_in_vocab: bool = True, bad_word_reward: float = -1.0, return_seeds: bool = False):
    """`n_data` episodes of the RandomMixturePolicy (with probability `prob_smart` a uniformly random word still consistent
    with the feedback, else a uniformly random vocabulary word) against the batched HIP Wordle env; returns jsonl items."""
    import torch
    from . import _lib
    from .envs import wordle as W
    dev = _lib.require_gpu()
    words = list(vocab.all_vocab)
    words5 = np.array([[ord(ch) - 97 for ch in w] for w in words], dtype=np.int64)
    packed = np.array([W.pack_guess(w) for w in words], dtype=np.uint32)
    rng = np.random.default_rng(seed)
    items: List[Dict[str, Any]] = []
    all_seeds: List[int] = []
    env = W.VectorWordleEnv(vocab, require_words_in_vocab, bad_word_reward)
    for b0 in range(0, n_data, bsize):
        n = min(bsize, n_data - b0)
        env_seeds = rng.integers(0, 2 ** 31 - 1, size=n, dtype=np.int64).astype(np.uint64)
        all_seeds += [int(x) for x in env_seeds]
        env.reset_device(env_seeds)
        hist = [[Text("Wordle:\n", False)] for _ in range(n)]
        rews: List[List[float]] = [[] for _ in range(n)]
        done = np.zeros(n, dtype=bool)
        while not done.all():
            trits, n_filtered, _ = env.export_state()
            fm = _filtered_mask(words5, trits)
            assert (fm.sum(axis=1) == n_filtered).all(), "host filter disagrees with the device env's filtered-vocabulary size"
            smart = rng.random(n) < prob_smart
            choice = np.empty(n, dtype=np.int64)
            for i in range(n):
                cand = np.flatnonzero(fm[i]) if smart[i] else None
                choice[i] = rng.choice(cand) if cand is not None and len(cand) else rng.integers(0, len(words))
            env.step_device(torch.from_numpy(packed[choice].view(np.int32)).to(dev), torch.from_numpy((~done).astype(np.uint8)).to(dev))
            obs = env.obs.cpu().numpy().view(np.uint32); rew = env.reward.cpu().numpy(); flg = env.flags.cpu().numpy()
            for i in range(n):
                if done[i]:
                    continue
                hist[i] += [Text(" ".join(words[choice[i]]) + "\n", True), Text(" ".join(W.obs_symbols(int(obs[i]))) + "\n", False)]
                rews[i] += [float(rew[i]), 0.0]
                done[i] = bool(flg[i] & 1)
        items += [text_trajectory_to_item(hist[i], rews[i], True) for i in range(n)]
    env.close()
    return (items, all_seeds) if return_seeds else items
