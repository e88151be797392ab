"""Generate tests/golden/twenty_questions.json by RUNNING the reference's Twenty Questions env / data / oracle code
(llm_rl_scripts/twenty_questions/env/{env,data,oracle}.py) with (a) a scripted oracle model and (b) the documented stand-in POS
tagger `rule_pos_tag` injected in place of nltk (nltk and its perceptron tagger model are not available offline; everything except
the tagger itself is therefore pinned).

    python tests/golden/make_fixtures_20q.py          (build container only: reads /root/reference)
"""
from __future__ import annotations

import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE); sys.path.insert(0, ROOT)
import _ref_import  # noqa: E402

_ref_import.install()
import nltk  # noqa: E402  (the permissive stub)
import lmrl_gym_amd  # noqa: E402,F401
from lmrl_gym_amd.envs.twenty_questions import rule_pos_tag  # noqa: E402

nltk.word_tokenize = lambda s: s          # the reference calls nltk.pos_tag(nltk.word_tokenize(text)): tokenisation + tagging = rule_pos_tag(text)
nltk.pos_tag = rule_pos_tag

from LLM_RL.environment import Text  # noqa: E402
from llm_rl_scripts.twenty_questions.env import data as RD  # noqa: E402
from llm_rl_scripts.twenty_questions.env.env import TwentyQuestionsPolicyEnvironment, BatchedTwentyQuestionsPolicyEnvironment  # noqa: E402
from llm_rl_scripts.twenty_questions.env.oracle import T5Oracle, get_t5_oracle_prompt, TwentyQuestionsOracle  # noqa: E402

QUESTIONS = ["Is it an animal?", "Is it a cat?", "is it a dog", "Is the object a pair of shoes?", "Does it have wheels?", "Is it a type of fruit?",
             "Is it a tennis ball?", "Is it a ball?", "Is it a computer keyboard?", "Is it a keyboard?", "Is it a TV?", "Is it a piece of paper?",
             "Is it a kind of tree?", "Is it something you wear?", "Is it a plane?", "   Is it a Helmet ?  ", "Is it bigger than a bear?", "Is it a rock or a tree?",
             "Is it a pack of batteries?", "Is it the ocean?!", "?", "Is it"]


def scripted_answer(word, question):
    """Deterministic oracle: 'Yes.' iff some spelling of the word occurs in the lower-cased question or the question length is a multiple of 5."""
    ql = question.lower()
    return "Yes." if any(w.lower() in ql for w in word.words) or len(question) % 5 == 0 else "No."


class ScriptedOracle(TwentyQuestionsOracle):
    def generate_answers(self, words, questions, return_full=False):
        if not isinstance(words, list):
            return scripted_answer(words, questions)
        return [scripted_answer(w, q) if q != RD.INVALID_QUESTION else "No." for w, q in zip(words, questions)]


def th(hist):
    return [[t.text, bool(t.is_action)] for t in hist]


def main():
    out = {"_meta": "reference twenty_questions env/data/oracle run with rule_pos_tag injected for nltk and a scripted oracle"}
    wl = RD.get_default_word_list()
    out["word_list"] = [w.words for w in wl]
    out["initial_str"], out["invalid_question"] = RD.INITIAL_STR, RD.INVALID_QUESTION
    # is_done table
    idx = {"Cat": None, "Dog": None, "Shoes": None, "Tennis ball": None, "Computer keyboard": None, "Television": None, "Paper": None, "Tree": None,
           "Airplane": None, "Helmet": None, "Bear": None, "Battery": None, "Ocean": None, "Rock": None}
    for i, w in enumerate(wl):
        if w.words[0] in idx:
            idx[w.words[0]] = i
    out["is_done"] = [dict(word=i, question=q, done=bool(RD.is_done(wl[i], q))) for i in idx.values() for q in QUESTIONS]
    # asker post-processing
    raw = ["is it alive", "", "Tell me what it is.", "Is it red? ", "does it fly", " ".join(["word"] * 45), "Is it big.", "Is it a cat, dog, etc.", "can you eat it?\n",
           " ".join(["is"] * 60), "what"]
    out["asker_postproc"] = [dict(raw=r, full=RD.asker_postproc(r), simple=RD.asker_postproc_simple(r), filt=RD.asker_postproc_filter_repeats(r)) for r in raw]
    # prompts + T5Oracle answer post-processing (oracle.py:44-87) through a fake inference object
    class FakeInf:
        def __init__(self, outs): self.outs = outs
        def generate_from_str(self, input_strs, **kw):
            class R: pass
            r = R(); r.output_strs = self.outs[: len(input_strs)]; return r
    model_outs = ["yes", " Yes it is", "no.", "NO", "maybe", "", "yes\n", "nope", "y", "Noyes"]
    qs = ["Is it a cat?"] * 9 + [RD.INVALID_QUESTION]
    orc = T5Oracle(None, FakeInf(model_outs), None, None)
    ans, full = orc.generate_answers([wl[10]] * 10, qs, return_full=True)
    out["oracle"] = dict(prompt=get_t5_oracle_prompt(wl[14], "Is it a plane?"), prompt_word=14, model_outs=model_outs, questions=qs, answers=ans, full=full,
                         single=orc.generate_answers(wl[10], "Is it a cat?"))
    # single env episodes
    eps = []
    for seed, det, qstart, maxlen in [(0, False, 0, 20), (1, False, 3, 20), (10, True, 0, 20), (11, True, 1, 5), (24, True, 0, 4), (123456789, False, 2, 3)]:
        env = TwentyQuestionsPolicyEnvironment(ScriptedOracle(), wl, max_conversation_length=maxlen)
        hist = env.reset(seed, {"deterministic": det})
        word = env.curr_word.words
        steps = []
        k = qstart
        done = False
        while not done:
            q = QUESTIONS[k % len(QUESTIONS)] + "\n"; k += 1
            hist, r, done = env.step(hist + (Text(q, True),))
            steps.append(dict(question=q, history=th(hist), reward=r, done=bool(done)))
        eps.append(dict(seed=seed, deterministic=det, qstart=qstart, maxlen=maxlen, word=word, steps=steps))
    out["episodes"] = eps
    # batched env, 5 slots, padded to bsize 6, lock-step with per-slot done handling as interact_environment does
    benv = BatchedTwentyQuestionsPolicyEnvironment(ScriptedOracle(), wl, max_conversation_length=6, bsize=6)
    seeds = [3, 14, 15, 92, 65]
    hists = benv.reset(seeds, [{"deterministic": i % 2 == 0} for i in range(5)])
    words = [w.words for w in benv.curr_words]
    done = [False] * 5
    rounds = []
    k = 0
    while not all(done):
        acts = [None if done[i] else tuple(hists[i]) + (Text(QUESTIONS[(k + 2 * i) % len(QUESTIONS)] + "\n", True),) for i in range(5)]
        res = benv.step(acts, done)
        rec = []
        for i, r in enumerate(res):
            if r is None:
                rec.append(None)
            else:
                hists[i], rew, dn = r
                done[i] = dn
                rec.append(dict(history=th(hists[i]), reward=rew, done=bool(dn)))
        rounds.append(rec); k += 1
    out["batched"] = dict(seeds=seeds, words=words, rounds=rounds)
    # a second batched run whose slots WIN at different steps (word guessed + "Yes." -> reward 0, done; env.py:101-117 / data.py:83-116):
    # deterministic words (seed % len(word_list)), 4 live slots in a batch of 4, question index k + i + 1
    benv = BatchedTwentyQuestionsPolicyEnvironment(ScriptedOracle(), wl, max_conversation_length=5, bsize=4)
    seeds = [10, 11, 24, 170]
    hists = benv.reset(seeds, [{"deterministic": True}] * 4)
    words = [w.words for w in benv.curr_words]
    done = [False] * 4
    rounds = []
    k = 0
    while not all(done):
        acts = [None if done[i] else tuple(hists[i]) + (Text(QUESTIONS[(k + i + 1) % len(QUESTIONS)] + "\n", True),) for i in range(4)]
        res = benv.step(acts, done)
        rec = []
        for i, r in enumerate(res):
            if r is None:
                rec.append(None)
            else:
                hists[i], rew, dn = r
                done[i] = dn
                rec.append(dict(history=th(hists[i]), reward=rew, done=bool(dn)))
        rounds.append(rec); k += 1
    out["batched_win"] = dict(seeds=seeds, words=words, rounds=rounds, question_offset=1, maxlen=5)
    path = os.path.join(HERE, "twenty_questions.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(f"wrote twenty_questions.json: {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
