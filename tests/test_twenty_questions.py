"""Twenty Questions env / data / oracle post-processing (lmrl_gym_amd.envs.twenty_questions) against tests/golden/twenty_questions.json —
outputs of the REFERENCE code (llm_rl_scripts/twenty_questions/env/{env,data,oracle}.py) run with a scripted oracle and the documented
stand-in POS tagger injected for nltk (tests/golden/make_fixtures_20q.py)."""
import pytest

import lmrl_gym_amd  # noqa: F401
from conftest import load_golden
from lmrl_gym_amd import environment as E
from lmrl_gym_amd.envs import twenty_questions as Q

G = load_golden("twenty_questions.json")


@pytest.fixture(autouse=True)
def _tagger():
    Q.set_pos_tagger(Q.rule_pos_tag)      # the tagger the fixtures were generated with
    yield
    Q.set_pos_tagger(None)


def scripted_answer(word, question):
    ql = question.lower()
    return "Yes." if any(w.lower() in ql for w in word.words) or len(question) % 5 == 0 else "No."


class ScriptedOracle(Q.TwentyQuestionsOracle):
    def generate_answers(self, words, questions, return_full=False):
        if not isinstance(words, list):
            return scripted_answer(words, questions)
        return [scripted_answer(w, q) if q != Q.INVALID_QUESTION else "No." for w, q in zip(words, questions)]


def th(hist):
    return [[t.text, bool(t.is_action)] for t in hist]


def test_word_list_constants_is_done_and_postproc():
    wl = Q.get_default_word_list()
    assert [w.words for w in wl] == G["word_list"] and Q.INITIAL_STR == G["initial_str"] and Q.INVALID_QUESTION == G["invalid_question"]
    assert str(wl[34]) == "(Airplane, Plane)" and repr(wl[34]) == "WordVariants([Airplane, Plane])" and len(wl[34]) == 2 and wl[34][1] == "Plane"
    for e in G["is_done"]:
        assert Q.is_done(wl[e["word"]], e["question"]) == e["done"], e
    assert any(e["done"] for e in G["is_done"]) and not all(e["done"] for e in G["is_done"])
    for e in G["asker_postproc"]:
        assert Q.asker_postproc(e["raw"]) == e["full"] and Q.asker_postproc_simple(e["raw"]) == e["simple"] and Q.asker_postproc_filter_repeats(e["raw"]) == e["filt"], e


def test_oracle_prompt_and_answer_postprocessing():
    wl = Q.get_default_word_list()
    o = G["oracle"]
    assert Q.get_oracle_prompt(wl[o["prompt_word"]], "Is it a plane?") == o["prompt"]
    orc = Q.ModelOracle(lambda prompts: o["model_outs"][: len(prompts)])
    ans, full = orc.generate_answers([wl[10]] * 10, o["questions"], return_full=True)
    assert ans == o["answers"] and full == o["full"]
    assert orc.generate_answers(wl[10], "Is it a cat?") == o["single"]


def test_single_env_episodes_match_reference():
    wl = Q.get_default_word_list()
    for ep in G["episodes"]:
        env = Q.TwentyQuestionsPolicyEnvironment(ScriptedOracle(), wl, max_conversation_length=ep["maxlen"])
        hist = env.reset(ep["seed"], {"deterministic": ep["deterministic"]})
        assert th(hist) == [[Q.INITIAL_STR, False]] and env.curr_word.words == ep["word"]
        for st in ep["steps"]:
            hist, r, done = env.step(tuple(hist) + (E.Text(st["question"], True),))
            assert th(hist) == st["history"] and r == st["reward"] and done == st["done"]
        assert done
    with pytest.raises(AssertionError):
        Q.TwentyQuestionsPolicyEnvironment(ScriptedOracle(), wl).step((E.Text("x", True),))        # reset() first
    with pytest.raises(AssertionError):
        Q.TwentyQuestionsPolicyEnvironment(ScriptedOracle(), wl).reset(None, {"deterministic": True})


def test_batched_env_matches_reference_and_runs_under_interact_environment():
    wl = Q.get_default_word_list()
    b = G["batched"]
    env = Q.BatchedTwentyQuestionsPolicyEnvironment(ScriptedOracle(), wl, max_conversation_length=6, bsize=6)
    hists = env.reset(b["seeds"], [{"deterministic": i % 2 == 0} for i in range(5)])
    assert [w.words for w in env.curr_words] == b["words"]
    questions = G["is_done"]
    QS = []
    for e in questions:
        if e["question"] not in QS:
            QS.append(e["question"])
    done = [False] * 5
    for k, rec in enumerate(b["rounds"]):
        acts = [None if done[i] else tuple(hists[i]) + (E.Text(QS[(k + 2 * i) % len(QS)] + "\n", True),) for i in range(5)]
        res = env.step(acts, done)
        for i, (r, e) in enumerate(zip(res, rec)):
            if e is None:
                assert r is None
                continue
            hists[i], rew, dn = r
            done[i] = dn
            assert th(hists[i]) == e["history"] and rew == e["reward"] and dn == e["done"]
    assert all(done)

    # the env behind the reference protocol: lock-step rollouts through interact_environment with a scripted asker
    class Asker(E.BatchedTextPolicy):
        def act(self, text_history, done=None):
            return [None if (h is None or (done and done[i])) else tuple(h) + (E.Text(Q.asker_postproc(f"is it a {wl[(i * 7 + len(h)) % len(wl)][0].lower()}"), True),)
                    for i, h in enumerate(text_history)]
    inter = E.interact_environment(Q.BatchedTwentyQuestionsPolicyEnvironment(ScriptedOracle(), wl, max_conversation_length=4, bsize=4), Asker(),
                                   initial_text_history=None, env_seed=[5, 6, 7, 8], env_options=None, bsize=4, npad=0)
    assert len(inter) == 4 and all(ep[-1].done and 1 <= len(ep) <= 4 for ep in inter)
    assert all(t.reward in (-1.0, 0.0) for ep in inter for t in ep)
