"""Generate the golden fixtures under tests/golden/ by RUNNING the reference's
pure-Python code (imported from /root/reference through `_ref_import`).

Run once in the build container:   python tests/golden/make_fixtures.py
The outputs (*.json) are committed; tests only read them.  The reference never
travels to the GPU box.

What is pinned here (reference file:line of the code that produced each file):
  mt19937.json        CPython `random.Random(seed)` — used at wordle/env/env.py:53,
                      wordle/env/game.py:178-179, maze/env/env.py:187-212
  wordle_traces_*.json  llm_rl_scripts/wordle/env/env.py:28-55 + game.py:53-296
  maze_traces.json    llm_rl_scripts/maze/env/env.py:8-214 + maze_utils.py:9-52,91-116
  rl_helpers.json     LLM_RL/algorithms/ppo/base_interface.py:38-69,230-343,
                      LLM_RL/algorithms/ilql/data.py:58-79, LLM_RL/environment.py:154-419,
                      LLM_RL/algorithms/ppo/reranker_policy.py:5-34
"""
from __future__ import annotations

import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

_ref_import.install()

from LLM_RL.environment import (  # noqa: E402
    Text, TextTrajectory, TextTrajectoryChain, TokenTrajectory, TokenTrajectoryChain, TokenHistory,
    TextPolicy, interact_environment, text_env_eval,
)
from LLM_RL.algorithms.ppo.base_interface import (  # noqa: E402
    get_advantages_and_returns, get_action_state_next_state_idxs, AdaptiveKLController,
    FixedKLController, CombinedTokenTrajectoryChain,
)
from LLM_RL.algorithms.ilql.data import ILQLData  # noqa: E402
from LLM_RL.algorithms.ppo.reranker_policy import ReRankerPolicy  # noqa: E402
from llm_rl_scripts.wordle.env.env import WordleEnvironment, ReformatWordleEnvironment  # noqa: E402
from llm_rl_scripts.wordle.env.game import Vocabulary, CharKnowledge  # noqa: E402
from llm_rl_scripts.maze.env.maze_utils import setup_maze_env, maze_solver  # noqa: E402
from llm_rl_scripts.maze.env.mazes import double_t_maze, maze2d_umaze, double_t_maze_optimal_directions  # noqa: E402
from llm_rl_scripts.maze.env.env import maze_proposal_function, manhatten_actions  # noqa: E402

VOCAB_DIR = os.path.join(_ref_import.REFERENCE_ROOT, "llm_rl_scripts/wordle/vocab")


def dump(name, obj):
    path = os.path.join(HERE, name)
    with open(path, "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print(f"wrote {name}: {os.path.getsize(path)/1024:.1f} KiB")


# ----------------------------------------------------------------------------- MT19937
def gen_mt19937():
    seeds = [0, 1, 2, 123, 12345, 2**31 - 1, 2**31, 2**32 - 1, 2**32, 2**32 + 1, 2**40 + 7,
             2**64 - 1, 2**64, 2**95 + 12345678901234567890, -5, -(2**33) - 3]
    cases = []
    for s in seeds:
        r = random.Random(s)
        first = [r.getrandbits(32) for _ in range(8)]
        r = random.Random(s)
        ns = [431, 2315, 1, 2, 3, 7, 12971, 100, 25, 24, 431, 17, 5, 2315, 64, 65]
        choices = [r.choice(range(n)) for n in ns]
        # a long run crossing the 624-word regeneration boundary twice
        r = random.Random(s)
        long_idx = [0, 1, 226, 227, 396, 397, 623, 624, 625, 1247, 1248, 1300]
        outs = [r.getrandbits(32) for _ in range(1301)]
        cases.append(dict(seed=str(s), first=first, ns=ns, choices=choices,
                          long_idx=long_idx, long_vals=[outs[i] for i in long_idx]))
    dump("mt19937.json", dict(cases=cases))


# ----------------------------------------------------------------------------- Wordle
def state_trits(game):
    # 26*5 chars in {0,1,2}: NOT_HERE=0, POSSIBLE=1, HERE=2   (game.py:17-20)
    return "".join(str(k.value) for cs in game.state.state for k in cs.position_knowledge)


def spaced(w):
    return " ".join(list(w)) + "\n"


def gen_wordle(vocab_file, tag, n_eps, seed0):
    vocab = Vocabulary.from_file(os.path.join(VOCAB_DIR, vocab_file), fill_cache=False)
    words = list(vocab.all_vocab)
    wset = set(words)
    prng = random.Random(seed0)
    letters = "abcdefghijklmnopqrstuvwxyz"
    episodes = []
    for ep in range(n_eps):
        cfg = ep % 8
        require = not (cfg == 5)
        bad = {0: -1.0, 1: -10.0, 2: -10.0, 3: -1.0, 4: -4.5, 5: -10.0, 6: 0.0, 7: -10.0}[cfg]
        seed = prng.choice([ep, ep * 7919 + 13, prng.getrandbits(31), prng.getrandbits(40)])
        env = ReformatWordleEnvironment(WordleEnvironment(vocab, require_words_in_vocab=require, bad_word_reward=bad))
        hist = env.reset(seed=seed)
        assert hist == (Text("Wordle:\n", False),)
        strategy = prng.choice(["random", "consistent", "consistent", "mixed", "mixed", "invalid_heavy"])
        steps = []
        done = False
        while not done:
            game = env.env.state
            u = prng.random()
            if strategy == "random":
                a = spaced(prng.choice(words))
            elif strategy == "consistent":
                a = spaced(prng.choice(game.vocab.filtered_vocab))
            else:
                p_inv = 0.25 if strategy == "mixed" else 0.6
                if u < p_inv:
                    kind = prng.randrange(9)
                    if kind == 0:   # 5 random letters, (almost surely) not in vocab
                        w = "".join(prng.choice(letters) for _ in range(5))
                        a = spaced(w)
                    elif kind == 1:  # wrong length
                        w = "".join(prng.choice(letters) for _ in range(prng.choice([1, 2, 3, 4, 6, 7, 9])))
                        a = spaced(w)
                    elif kind == 2:  # non a-z character
                        w = list(prng.choice(words)); w[prng.randrange(5)] = prng.choice("AZ-3?é")
                        a = spaced("".join(w))
                    elif kind == 3:  # empty
                        a = "\n"
                    elif kind == 4:  # valid word, odd spacing (deformat strips / removes spaces: env.py:23)
                        w = prng.choice(words)
                        a = "  " + w[:2] + " " + w[2:] + "  \n"
                    elif kind == 5:  # valid word no spaces, no newline
                        a = prng.choice(words)
                    elif kind == 6:  # inner tab makes it invalid (only ' ' is removed)
                        w = prng.choice(words)
                        a = w[:2] + "\t" + w[2:] + "\n"
                    elif kind == 7:  # uppercase word
                        a = spaced(prng.choice(words).upper())
                    else:           # repeat a previous guess if any
                        prev = [s["action"] for s in steps]
                        a = prng.choice(prev) if prev else spaced(prng.choice(words))
                else:
                    a = spaced(prng.choice(game.vocab.filtered_vocab if prng.random() < 0.7 else words))
            hist = hist + (Text(a, True),)
            hist, r, done = env.step(hist)
            g = env.env.state
            steps.append(dict(
                action=a, obs=hist[-1].text, reward=float(r), reward_is_int=isinstance(r, int), done=bool(done),
                state=state_trits(g), n_filtered=g.vocab.filtered_vocab_size(),
            ))
            assert hist[-1].is_action is False
        episodes.append(dict(seed=seed, require_in_vocab=require, bad_word_reward=bad, strategy=strategy, steps=steps))
    n_steps = sum(len(e["steps"]) for e in episodes)
    n_win = sum(e["steps"][-1]["reward"] == 0 and e["steps"][-1]["reward_is_int"] for e in episodes)
    print(f"wordle {tag}: {n_eps} episodes, {n_steps} steps, {n_win} wins")
    dump(f"wordle_traces_{tag}.json", dict(vocab_file=vocab_file, n_words=len(words), episodes=episodes))


# ----------------------------------------------------------------------------- Maze
def gen_maze():
    prng = random.Random(777)
    out = []
    action_pool = list(manhatten_actions.keys())
    junk = ["move north\n", "left\n", "move left", "Move left\n", "\n", "", "move up \n", "jump\n"]
    for maze_name in ["double_t_maze", "umaze"]:
        for desc in ["describe_observation", "describe_observation_give_position", "describe_observation_only_walls"]:
            for rew in ["standard_reward", "illegal_penalty_reward", "illegal_penalty_diff_scale"]:
                for last_k, max_steps in [(1, 100), (40, 12), (3, 5)]:
                    for rep in range(3):
                        env = setup_maze_env(maze_name, desc, rew, last_k=last_k, max_steps=max_steps)
                        seed = prng.choice([rep, prng.getrandbits(30), 2**33 + rep])
                        use_opts = rep == 2
                        options = None
                        if use_opts:
                            free = np.argwhere(env.maze == 0).tolist()
                            goal = env.valid_goals[0].tolist()
                            free.remove(goal)
                            options = dict(goal=goal, init_position=prng.choice(free))
                        state_before = random.getstate()
                        hist = env.reset(seed=seed, options=options)
                        assert random.getstate() == state_before  # RandomState save/restore (randomness.py:9-19)
                        ep = dict(maze=maze_name, describe=desc, reward_fn=rew, last_k=last_k, max_steps=max_steps,
                                  seed=seed, options=options, reset_obs=hist[0].text,
                                  init_position=list(env.position), goal=list(env.goal), steps=[])
                        policy = maze_solver(1 - env.maze, [tuple(env.goal)])
                        done = False
                        guard = 0
                        while not done and guard < 130:
                            guard += 1
                            u = prng.random()
                            if u < 0.55:
                                a = policy[tuple(env.position)]
                            elif u < 0.85:
                                a = prng.choice(action_pool)
                            else:
                                a = prng.choice(junk)
                            hist_in = hist + (Text(a, True),)
                            hist, r, done = env.step(hist_in)
                            ep["steps"].append(dict(action=a, reward=float(r), done=bool(done),
                                                    position=list(env.position), num_steps=env.num_steps,
                                                    history=[[t.text, t.is_action] for t in hist]))
                        out.append(ep)
    # known-answer table held by the reference itself (mazes.py:20-48) vs its BFS solver (maze_utils.py:91-116)
    maze = double_t_maze()
    sol = maze_solver(1 - maze, [(8, 6)])
    ka = double_t_maze_optimal_directions()
    assert all(sol[k] == v for k, v in ka.items())
    n_steps = sum(len(e["steps"]) for e in out)
    print(f"maze: {len(out)} episodes, {n_steps} steps")
    dump("maze_traces.json", dict(
        episodes=out,
        double_t_maze=maze.tolist(), umaze=maze2d_umaze().tolist(),
        double_t_maze_optimal_directions=[[list(k), v] for k, v in ka.items()],
    ))


# ----------------------------------------------------------------------------- RL helpers
class CharTokenizer:
    """Deterministic stand-in tokenizer: one token per character (id = ord), pad = 0."""
    pad_token_id = 0

    def encode(self, s):
        return [ord(c) for c in s]

    def decode(self, ids):
        return "".join(chr(i) for i in ids)


class ScriptedWordlePolicy(TextPolicy):
    def __init__(self, words, seed):
        self.words, self.rng = words, random.Random(seed)

    def act(self, text_history):
        return text_history + (Text(spaced(self.rng.choice(self.words)), True),)


def _tt_to_json(tt):
    return dict(tokens=tt.tokens.tolist(), is_action=tt.is_action.astype(int).tolist(),
                reward=tt.reward.tolist(), done=bool(tt.done))


def gen_rl_helpers():
    rng = np.random.RandomState(0)
    out = {}

    # GAE (ppo/base_interface.py:253-293), float64 numpy in the reference
    gae = []
    for (b, n, gamma, lam) in [(1, 3, .99, .95), (4, 7, 1.0, .95), (3, 36, .99, .9), (2, 1, 1.0, 1.0), (5, 12, .9, 0.0)]:
        v = rng.randn(b, n).astype(np.float32); nv = rng.randn(b, n).astype(np.float32); r = rng.randn(b, n).astype(np.float32)
        a, ret = get_advantages_and_returns(v, nv, r, gamma=gamma, lam=lam, use_whitening=False)
        gae.append(dict(values=v.tolist(), next_values=nv.tolist(), rewards=r.tolist(), gamma=gamma, lam=lam,
                        advantages=np.asarray(a).tolist(), returns=np.asarray(ret).tolist(),
                        out_dtype=str(np.asarray(a).dtype)))
    a, ret = get_advantages_and_returns(np.array([[.1, .2, .3]]), np.array([[.2, .3, 0]]), np.array([[-1, -1, 0.]]),
                                        gamma=.99, lam=.95, use_whitening=False)
    gae.append(dict(values=[[.1, .2, .3]], next_values=[[.2, .3, 0]], rewards=[[-1, -1, 0.]], gamma=.99, lam=.95,
                    advantages=a.tolist(), returns=ret.tolist(), out_dtype=str(a.dtype)))
    out["gae"] = gae

    # idxs (ppo/base_interface.py:230-243)
    idx_cases = []
    for mask in [[0, 1, 1, 0, 0, 1, 0], [1, 1, 1], [0, 0, 0, 0], [1], [0, 0, 1], [1, 0, 0, 0, 1, 1, 0, 1]]:
        m = np.array(mask, dtype=bool)
        a_, s_, n_ = get_action_state_next_state_idxs(m)
        idx_cases.append(dict(mask=mask, action=a_.tolist(), state=s_.tolist(), next_state=n_.tolist()))
    for _ in range(6):
        m = rng.rand(rng.randint(1, 40)) < 0.4
        a_, s_, n_ = get_action_state_next_state_idxs(m)
        idx_cases.append(dict(mask=m.astype(int).tolist(), action=a_.tolist(), state=s_.tolist(), next_state=n_.tolist()))
    out["idxs"] = idx_cases

    # KL controllers (ppo/base_interface.py:38-69)
    kl = []
    for init, target, horizon in [(0.001, 0.1, 10000), (0.2, 6.0, 1000), (0.05, 0.01, 64)]:
        c = AdaptiveKLController(init, target, horizon)
        seq = []
        for cur, n_steps in [(0.05, 32), (0.5, 32), (0.0, 128), (7.0, 256), (0.0999, 1), (0.12, 64)]:
            c.update(cur, n_steps)
            seq.append(dict(current=cur, n_steps=n_steps, value=float(c.value)))
        kl.append(dict(init=init, target=target, horizon=horizon, seq=seq))
    f = FixedKLController(0.3); f.update(9.0, 10)
    out["kl"] = dict(adaptive=kl, fixed_after_update=f.value)

    # token containers + data shaping with a 1-char-per-token tokenizer
    tok = CharTokenizer()
    th1 = (Text("Wordle:\n", False), Text("s t a r e\n", True), Text("b y b b g\n", False), Text("c r a n e\n", True), Text("g g g g g\n", False))
    tt1 = TextTrajectory(th1, (0.0, -1.0, 0.0, 0.0, 0.0), True)
    th2a = (Text("obs A\n", False), Text("move up\n", True), Text("obs B\n", False), Text("move left\n", True))
    th2b = (Text("move left\n", True), Text("obs C\n", False), Text("move down\n", True), Text("Success\n", False))
    th2c = (Text("tail only state\n", False),)
    chain_c = TextTrajectoryChain(TextTrajectory(th2c, (0.0,), True), None)
    chain_b = TextTrajectoryChain(TextTrajectory(th2b, (-1.0, 0.0, 0.0, 0.0), False), chain_c)
    chain_a = TextTrajectoryChain(TextTrajectory(th2a, (0.0, -1.0, 0.0, -4.0), False), chain_b)
    chain_1 = TextTrajectoryChain(tt1, None)
    data = []
    for name, ch in [("single_done", chain_1), ("three_chunk", chain_a), ("two_chunk", chain_b)]:
        tch = TokenTrajectoryChain.from_text_trajectory_chain(ch, tok)
        entry = dict(name=name, text_chain=[], token_chain=[_tt_to_json(t) for t in tch.to_list()])
        cur = ch
        while cur is not None:
            entry["text_chain"].append(dict(history=[[t.text, t.is_action] for t in cur.text_trajectory.text_history],
                                            reward=list(cur.text_trajectory.reward), done=cur.text_trajectory.done))
            cur = cur.next
        d = ILQLData.from_token_trajectory_chain(tch)
        entry["ilql_data"] = dict(
            input_ids=d.input_ids.tolist(), should_take_action=d.should_take_action.astype(int).tolist(),
            rewards=d.rewards.tolist(), done=bool(d.done),
            next_token_ids=None if d.next_token_ids is None else d.next_token_ids.tolist(),
            next_done=None if d.next_done is None else bool(d.next_done))
        combos = {}
        for ml in [None, 64]:
            try:
                c = CombinedTokenTrajectoryChain.from_token_trajectory_chain(tch, max_length=ml)
                combos[str(ml)] = dict(input_tokens=c.input_tokens.tolist(), output_tokens=c.output_tokens.tolist(),
                                       rewards=c.rewards.tolist(), should_take_action=c.should_take_action.astype(int).tolist(),
                                       done=bool(c.done), chunk_lens=[int(x) for x in c.chunk_lens])
            except AssertionError as e:
                combos[str(ml)] = dict(error=str(e))
        entry["combined"] = combos
        data.append(entry)
    hist_tok = TokenHistory.from_text_history(th1, tok)
    out["token_history"] = dict(history=[[t.text, t.is_action] for t in th1], tokens=hist_tok.tokens.tolist(),
                                is_action=hist_tok.is_action.astype(int).tolist())
    out["chains"] = data

    # interact_environment / text_env_eval (environment.py:154-267) with a scripted policy on the real Wordle env
    vocab = Vocabulary.from_file(os.path.join(VOCAB_DIR, "wordle_official_400.txt"), fill_cache=False)
    env = ReformatWordleEnvironment(WordleEnvironment(vocab, require_words_in_vocab=True, bad_word_reward=-10.0))
    import io, contextlib
    pol = ScriptedWordlePolicy(list(vocab.all_vocab) + ["zzzzz", "abc"], seed=99)
    seeds = iter(range(1000, 1100))
    with contextlib.redirect_stdout(io.StringIO()):  # TextPolicyToBatchedTextPolicy prints (environment.py:130-132)
        inter, summary = text_env_eval(env, pol, n_rollouts=7, seed_generator=seeds, bsize=3, verbose=False)
    out["text_env_eval"] = dict(
        policy_seed=99, extra_words=["zzzzz", "abc"], n_rollouts=7, bsize=3, first_seed=1000,
        summary={k: {kk: float(vv) for kk, vv in v.items()} for k, v in summary.items()},
        interactions=[[dict(pre=[[t.text, t.is_action] for t in tr.pre_action_history],
                            post_action=[[t.text, t.is_action] for t in tr.post_action_history],
                            post_transition=[[t.text, t.is_action] for t in tr.post_transition_history],
                            reward=float(tr.reward), done=bool(tr.done)) for tr in ep] for ep in inter])

    # ReRankerPolicy (reranker_policy.py:21-34) with the maze proposal function (maze/env/env.py:101-102)
    th = (Text("There is a wall above you.\n", False),)
    scores = [0.1, 0.7, 0.7, -3.0]
    rr = ReRankerPolicy(maze_proposal_function, lambda props: scores)
    res = rr.act(th)
    out["reranker"] = dict(history=[[t.text, t.is_action] for t in th], scores=scores,
                           proposals=[[[t.text, t.is_action] for t in p] for p in maze_proposal_function(th)],
                           chosen=[[t.text, t.is_action] for t in res])
    dump("rl_helpers.json", out)


if __name__ == "__main__":
    gen_mt19937()
    gen_wordle("wordle_official_400.txt", "v431", 240, 1)
    gen_wordle("wordle_official.txt", "v2315", 48, 2)
    gen_maze()
    gen_rl_helpers()
