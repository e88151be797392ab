"""CPU tier: the jsonl trajectory format, item -> chain map, mask datasets, dataloader and the host-side vocabulary filter
of lmrl_gym_amd.datasets."""
import json

import numpy as np

import lmrl_gym_amd  # noqa: F401
from lmrl_gym_amd import datasets as DS
from lmrl_gym_amd import environment as E
from lmrl_gym_amd.algorithms.common import BlockingStrategy, Padding, Truncation


ITEM = {"sequence": [["Wordle:\n", 0.0], ["s t a r e\n", 1.0], ["b y b b g\n", 0.0], ["c r a n e\n", 1.0], ["g g g g g\n", 0.0]],
        "reward": [-1.0, 0.0, 0.0, 0.0], "done": True}


def test_jsonl_roundtrip_and_item_map(tmp_path):
    p = tmp_path / "d.jsonl"
    assert DS.write_jsonl(str(p), [ITEM, ITEM]) == 2
    items = list(DS.jsonl_stream(str(p)))
    assert items == [json.loads(json.dumps(ITEM))] * 2
    ch = DS.item_to_text_chain(items[0])
    tt = ch.text_trajectory
    assert ch.next is None and tt.done is True and tt.reward == (0.0, -1.0, 0.0, 0.0, 0.0)
    assert [t.is_action for t in tt.text_history] == [False, True, False, True, False]
    # data_gen.py writes (history after the header, rewards without the header element): inverse of the map above
    back = DS.text_trajectory_to_item(tt.text_history, tt.reward[1:], tt.done)
    assert back["sequence"] == [tuple(x) for x in ITEM["sequence"]] and back["reward"] == ITEM["reward"] and back["done"]


def test_ilql_dataset_and_dataloader_from_jsonl(tmp_path):
    tok = DS.WordleTokenizer()
    p = tmp_path / "d.jsonl"
    short = {"sequence": ITEM["sequence"][:3], "reward": [-1.0, 0.0], "done": False}
    DS.write_jsonl(str(p), [ITEM, short, ITEM, short, ITEM])
    bs = BlockingStrategy(Padding.RIGHT, Truncation.RIGHT, 32)
    ds = DS.ilql_dataset_from_jsonl(str(p), tok, bs)
    assert len(ds) == 5 and ds.input_ids.shape == (5, 32) and ds.should_take_action.shape == (5, 31)
    ids0 = tok.encode("Wordle:\n") + tok.encode("s t a r e\n") + tok.encode("b y b b g\n") + tok.encode("c r a n e\n") + tok.encode("g g g g g\n")
    assert ds.input_ids[0, :len(ids0)].tolist() == ids0 and (ds.input_ids[0, len(ids0):] == tok.pad_token_id).all()
    # reward sits on the last token of the action text; should_take_action marks action tokens (shifted by one)
    a0 = len(tok.encode("Wordle:\n")); a1 = a0 + len(tok.encode("s t a r e\n"))
    assert ds.should_take_action[0, a0 - 1:a1 - 1].all() and not ds.should_take_action[0, :a0 - 1].any()
    assert ds.rewards[0, a1 - 2] == -1.0 and np.count_nonzero(ds.rewards[0]) == 1
    assert ds.dones.tolist() == [True, False, True, False, True]
    assert tok.decode(ds.input_ids[1]) == "Wordle:\ns t a r e\nb y b b g\n"
    # dataloader: truncate drops the ragged batch, rng=None keeps order, a seeded rng permutes reproducibly
    b = list(DS.dataloader(None, ds, 2, truncate=True))
    assert len(b) == 2 and b[0]["input_ids"].shape == (2, 32) and b[0]["next_token_ids"] is None
    assert (b[0]["input_ids"] == ds.input_ids[:2]).all()
    assert len(list(DS.dataloader(None, ds, 2, truncate=False))) == 3
    o1 = [x["dones"].tolist() for x in DS.dataloader(np.random.default_rng(3), ds, 2)]
    o2 = [x["dones"].tolist() for x in DS.dataloader(np.random.default_rng(3), ds, 2)]
    assert o1 == o2
    # mask dataset (PPO BC batch / BC trainer input)
    md = DS.MaskDataset.from_jsonl(str(p), tok, bs)
    assert md.input_ids.shape == (5, 32) and (md.input_ids == ds.input_ids).all()
    assert md.input_training_mask[0, a0:a1].tolist() == [1.0] * (a1 - a0) and md.input_training_mask[0, :a0].sum() == 0


def test_host_vocabulary_filter_matches_oracle_state():
    """_filtered_mask restates WordleState.word_in_state on the exported 26x5 trits: compare with the oracle env's own
    filtered vocabulary along scripted games."""
    from lmrl_gym_amd.envs import wordle as W
    from oracle.wordle import OracleWordleEnv
    words = W.Vocabulary.builtin("wordle_official_400.txt").all_vocab
    words5 = np.array([[ord(c) - 97 for c in w] for w in words])
    rng = np.random.RandomState(0)
    for ep in range(12):
        o = OracleWordleEnv(words, True, -1.0)
        hist = o.reset(ep)
        done = False
        while not done:
            w = words[rng.randint(len(words))]
            hist, r, done = o.step(hist + ((" ".join(w) + "\n", True),))
            trits, n_filtered = o.state()
            fm = DS._filtered_mask(words5, trits.reshape(1, 26, 5))[0]
            assert int(fm.sum()) == n_filtered
            # every surviving word is consistent with what the feedback text says about the guess
            sym = hist[-1][0].split()
            for i in np.flatnonzero(fm)[:20]:
                for k, (g, s_) in enumerate(zip(w, sym)):
                    if s_ == "g":
                        assert words[i][k] == g
                    elif s_ == "b":
                        assert g not in words[i]
