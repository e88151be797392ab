"""GPU tier: configs[3] and configs[4] of BASELINE.json AT THEIR SIZES on one GPU (VERDICT r03 "missing #2" / item 1) — the per-GPU share of
the 8-GPU configurations is what one rank runs; only the 8-way split itself stays unmeasured here.

  * configs[3], env side: 4096 chess boards in one lock-step `lmrl_chess_agent_step` / `lmrl_chess_opponent_step` round
    (llm_rl_scripts/chess/env/env.py:91-170) — EVERY board against the Stockfish-built fixture (the oracle of the rules: the position the
    reference's engine reaches after the same move), a sample against the host faces of the rules string for string, every illegal / garbage
    action against the host faces, FEN round trip of all 4096 final positions; then the text env (`VectorChessEnv`, 4096 slots, random opponent).
    (configs[3]'s train side — the GPT-2-medium PPO step at B = 32 x T = 1024 — is in tests/test_gpu_train_at_size.py.)
  * configs[4]: Twenty Questions with a GPT-2-LARGE oracle model and a GPT-2-MEDIUM guesser, both resident on the HIP engine, 1024 lock-step
    envs through `interact_environment` over `BatchedTwentyQuestionsPolicyEnvironment` (twenty_questions/env/env.py:66-141): protocol
    invariants on all 1024 episodes, every answer == the reference post-processing (oracle.py:62-79) of the oracle engine's own tokens, and a
    sample of the oracle's greedy generations re-scored token by token on oracle/gpt2.py (GPT-2-large, 36 layers, CPU).
"""
import json
import os
import random
import sys

import numpy as np
import pytest

import lmrl_gym_amd  # noqa: F401

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

N_BOARDS = 4096          # configs[3]: 4096 envs


def _fixture_triples():
    """(position, its legal moves, the move the fixture plays, the NEXT position + legal moves + move, the position after that or None) for every
    non-final step of tests/golden/chess_perft.json (Stockfish 15.1 built from the reference's sources: `go perft 1`, `d`)."""
    fx = json.load(open(os.path.join(HERE, "golden", "chess_perft.json")))
    out = []
    for g in fx["games"]:
        st = g["steps"]
        for t in range(len(st) - 1):
            if st[t]["move"] is None:
                continue
            nxt2 = st[t + 2] if t + 2 < len(st) and st[t + 1]["move"] is not None else None
            out.append((st[t], st[t + 1], nxt2))
    return out


def test_chess_4096_boards_one_lock_step_round_vs_stockfish_fixture_and_host_rules():
    from test_chess_rules import Board, _same_position
    from lmrl_gym_amd.envs import chess as C
    tri = _fixture_triples()
    assert len(tri) > 2000
    rng = random.Random(17)
    picks = [tri[i % len(tri)] for i in range(N_BOARDS)]
    rng.shuffle(picks)
    boards = C.VectorChessBoards()
    boards.reset([p[0]["fen"] for p in picks])
    moves, status, fens0 = boards.describe()
    GARBAGE = ["Ke9", "xx", "Qh5", "e4e5e6", "", "O-O", "--", "Nf3 ", "a9", "e8=K"]
    actions, garbage = [], []
    for i, (cur, nxt, _) in enumerate(picks):
        assert sorted(u for u, _ in moves[i]) == cur["legal"]
        if rng.random() < 0.12:
            actions.append(rng.choice(GARBAGE)); garbage.append(True)
        else:
            actions.append(dict(moves[i])[cur["move"]]); garbage.append(False)       # the fixture's move, spelled in SAN by the device
    res, rew, dn, fens1, played = boards.agent_step(actions, [True] * N_BOARDS)
    n_moved = n_garbage_legal = 0
    for i, (cur, nxt, _) in enumerate(picks):
        if garbage[i]:
            # every non-move against the host faces of the same rules (pinned to Stockfish in tests/test_chess_rules.py)
            hb = Board(fens0[i])
            hres, hrew, hdone = hb.agent(actions[i])
            assert (res[i], rew[i], bool(dn[i])) == (hres, hrew, bool(hdone)), (fens0[i], actions[i])
            assert fens1[i] == hb.fen()
            if hres in (C.MOVED, C.GAME_OVER):
                n_garbage_legal += 1                                          # "Qh5" / "O-O" can be legal somewhere
            else:
                assert fens1[i] == fens0[i] and (rew[i] == -1.0)              # illegal: -1, same position (env.py:104-113)
            continue
        # the position Stockfish reaches after the same move
        assert played[i] == cur["move"] and res[i] in (C.MOVED, C.GAME_OVER), (fens0[i], actions[i], res[i])
        _same_position(fens1[i], nxt["fen"], nxt["legal"])
        mate = nxt["check"] and not nxt["legal"]
        assert (rew[i] == 1.0) == mate and (res[i] == C.GAME_OVER) >= mate
        n_moved += 1
    assert n_moved > 3400 and sum(garbage) > 300
    # sample: device vs host string for string
    for i in rng.sample([i for i in range(N_BOARDS) if not garbage[i]], 256):
        hb = Board(fens0[i])
        hres, hrew, hdone = hb.agent(actions[i])
        assert (res[i], rew[i], bool(dn[i]), fens1[i]) == (hres, hrew, bool(hdone), hb.fen())
    # opponent half-step: the fixture's next move on every board that is still in play
    need = [(not garbage[i]) and res[i] == C.MOVED and picks[i][1]["move"] is not None for i in range(N_BOARDS)]
    ucis = [picks[i][1]["move"] if need[i] else "" for i in range(N_BOARDS)]
    moves1, _, _ = boards.describe()
    sans, rew2, dn2, fens2 = boards.opponent_step(ucis, need)
    n_opp = 0
    for i in range(N_BOARDS):
        if not need[i]:
            continue
        assert sans[i] == dict(moves1[i])[ucis[i]]
        nxt2 = picks[i][2]
        if nxt2 is not None:
            _same_position(fens2[i], nxt2["fen"], nxt2["legal"])
            assert sans[i].endswith(("+", "#")) == nxt2["check"]
            assert (rew2[i] == -1.0) == (nxt2["check"] and not nxt2["legal"])          # mated by the opponent: -1 (env.py:157-170)
        n_opp += 1
    assert n_opp > 3000
    for i in rng.sample([i for i in range(N_BOARDS) if need[i]], 256):
        hb = Board(fens1[i])
        ok, hsan, hrew, hdone = hb.opponent(ucis[i])
        assert ok and (sans[i], rew2[i], bool(dn2[i]), fens2[i]) == (hsan, hrew, bool(hdone), hb.fen())
    # property on ALL boards: the printed FEN is a fixed point (FEN -> position -> FEN) and describes the same legal set
    movesF, statusF, fensF = boards.describe()
    again = C.VectorChessBoards()
    again.reset(fensF)
    movesG, statusG, fensG = again.describe()
    assert fensG == fensF and movesG == movesF
    # repetition counters are not part of a FEN: every other status bit must agree
    assert ((statusF ^ statusG) & ~np.uint8(32 | 4)).max() == 0


def test_chess_text_env_4096_slots_random_opponent():
    """`VectorChessEnv` (FenChessHistoryEnv x 4096, env.py:213-238) for three lock-step turns: actions = a random legal move of the shown position
    (spelled as the reference's policies do), 10 % garbage; a 192-slot sample is replayed on the host rules turn by turn."""
    from test_chess_rules import Board
    from lmrl_gym_amd import environment as E
    from lmrl_gym_amd.envs import chess as C
    env = C.VectorChessEnv(max_moves=3, random_opponent=True)
    np.random.seed(5)
    rng = random.Random(6)
    hist = env.reset([None] * N_BOARDS)
    sample = rng.sample(range(N_BOARDS), 192)
    host = {i: Board(C.START_FEN) for i in sample}
    done = [False] * N_BOARDS
    n_steps = 0
    for turn in range(4):
        if all(done):
            break
        legal, _, fens = env.boards.describe()
        acts = []
        for i, h in enumerate(hist):
            if done[i]:
                acts.append(None)
                continue
            assert C.postprocess_state(h[-1].text) == fens[i]                 # the observation is the device board's FEN
            mv = rng.choice(["xx", "Ke9", ""]) if rng.random() < 0.10 or not legal[i] else rng.choice(legal[i])[1]
            acts.append(tuple(h) + (E.Text(C.preprocess_move(mv), True),))
        res = env.step(acts, done)
        for i in range(N_BOARDS):
            if done[i]:
                assert res[i] is None
                continue
            (obs,), r, d = res[i]
            assert r in (0.0, 1.0, -1.0) and not obs.is_action
            if i in host:
                hres, hrew, hdone = host[i].agent(C.postprocess_move(acts[i][-1].text))
                if hres == C.MOVED:
                    ok, hsan, hrew, hdone = host[i].opponent(env.moves[i][-1])
                    assert ok and hsan == env.last_opponent_moves[i]
                assert obs.text == C.preprocess_state_og(host[i].fen()) and r == hrew
                assert d == (bool(hdone) or env.num_moves_made[i] > env.max_moves)
            hist[i], done[i] = res[i][0], d
            n_steps += 1
    assert all(done) and n_steps >= 3 * N_BOARDS
    env.close()


class ByteTok:
    """One token per byte (no GPT-2 BPE files offline); the models keep GPT-2's 50 257-row tables.  ids >= 256 decode to nothing."""
    pad_token_id, eos_token_id = 50256, 10

    def encode(self, s):
        return list(s.encode("utf-8", errors="replace"))

    def decode(self, ids, skip_special_tokens=True):
        return bytes(int(i) for i in ids if int(i) < 256).decode("latin-1")


def test_twenty_questions_large_oracle_medium_guesser_1024_envs():
    from lmrl_gym_amd import _lib, environment as E
    from lmrl_gym_amd.envs import twenty_questions as Q
    from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine, init_hf_style_state_dict
    from lmrl_gym_amd.policies import GPT2PPOPolicy
    from oracle import gpt2 as O
    dev = _lib.require_gpu()
    B, TURNS, QTOK = 1024, 2, 12
    tok = ByteTok()
    cfg_g, cfg_o = GPT2Config.gpt2_medium(), GPT2Config.gpt2_large()
    assert (cfg_g.n_layer, cfg_g.d_model, cfg_o.n_layer, cfg_o.d_model) == (24, 1024, 36, 1280)
    sd_o = O.round_weights_to_bf16(init_hf_style_state_dict(cfg_o, seed=2))
    guesser = GPT2Engine.random_init(cfg_g, seed=1, device=dev)
    oracle_eng = GPT2Engine(cfg_o, sd_o, dev)
    Q.set_pos_tagger(Q.rule_pos_tag)
    try:
        wl = Q.get_default_word_list()
        calls, gen_ids = [], []

        class Spy(Q.GPT2EngineOracle):
            def generate_answers(self, words, questions, return_full=False):
                n0 = len(gen_ids)
                ans = super().generate_answers(words, questions, return_full)
                calls.append((list(words), list(questions), list(ans), gen_ids[n0:]))
                return ans
        MAX_IN = 192
        oracle = Spy(oracle_eng, tok, max_input_length=MAX_IN, max_new_tokens=4, eos_token_id=10)
        dec = oracle._policy._decode_generation
        oracle._policy._decode_generation = lambda ids: (gen_ids.append([int(x) for x in ids]), dec(ids))[1]      # the engine's own token ids
        asker = GPT2PPOPolicy(guesser, tok, max_input_length=64 + TURNS * (QTOK + 8), max_new_tokens=QTOK, do_sample=True, temperature=1.0, seed=3,
                              eos_token_id=10, out_str_process=Q.asker_postproc_filter_repeats)
        env = Q.BatchedTwentyQuestionsPolicyEnvironment(oracle, wl, max_conversation_length=TURNS, bsize=B)
        inter = E.interact_environment(env, asker, env_seed=list(range(B)), env_options=[{"deterministic": True}] * B, bsize=B)
        # ---- protocol invariants on all 1024 episodes
        assert len(inter) == B and [w.words for w in env.curr_words] == [wl[s % len(wl)].words for s in range(B)]
        for ep in inter:
            assert 1 <= len(ep) <= TURNS and ep[-1].done and not any(t.done for t in ep[:-1])
            for k, t in enumerate(ep):
                q, a = t.post_action_history[-1], t.post_transition_history[-1]
                assert q.is_action and q.text.endswith("?\n") and not a.is_action and a.text in ("Yes.\n", "No.\n")
                assert len(t.post_transition_history) == 3 + 2 * k and t.post_transition_history[0].text == Q.INITIAL_STR
                assert t.reward in (-1.0, 0.0) and (t.reward == 0.0) <= t.done
        assert sum(len(ep) for ep in inter) >= B * TURNS - 8            # a random oracle (almost) never confirms a random-byte question
        # ---- every answer is the reference post-processing of the oracle ENGINE's own tokens, whole batch, every turn
        assert len(calls) == TURNS and all(len(w) == B and len(ids) == B for w, _, _, ids in calls)
        for words, questions, answers, ids in calls:
            outs = [tok.decode(x) for x in ids]
            assert Q.answers_from_outputs(questions, outs)[0] == answers
        assert len({tuple(x) for _, _, _, ids in calls for x in ids}) > 8     # the oracle model's output depends on its prompt
        # ---- a sample of the oracle's generations re-scored on the CPU restatement (GPT-2-large, 36 layers, the same bf16-rounded weights):
        # every greedy token whose top-2 margin survives bf16 (> 0.05) must be the restatement's argmax on the reference prompt
        words, questions, answers, ids = calls[-1]
        rng = random.Random(1)
        n_tok = n_seq = 0
        for b in rng.sample(range(B), 24):
            if n_seq >= 5:
                break
            prompt = tok.encode(Q.get_oracle_prompt(words[b], questions[b]))[-MAX_IN:]
            seq, ok = list(prompt), True
            for t_dev in ids[b]:
                lg = O.forward(sd_o, torch.tensor([seq]), cfg_o.n_head, dtype=torch.float32)[0, -1, : cfg_o.vocab]
                top2 = lg.topk(2)
                if float(top2.values[0] - top2.values[1]) <= 0.05:
                    ok = False
                    break
                assert int(top2.indices[0]) == t_dev, (b, seq[len(prompt):], t_dev, int(top2.indices[0]))
                seq.append(t_dev); n_tok += 1
                if t_dev == 10:
                    break
            n_seq += ok
        assert n_tok >= 8, n_tok
    finally:
        Q.set_pos_tagger(None)
