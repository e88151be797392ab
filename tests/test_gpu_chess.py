"""GPU tier: the batched chess env (lmrl_gym_amd.envs.chess, csrc/chess.hip) — the device half-step kernels against the host faces of the same
rules (pinned to the reference's Stockfish in tests/test_chess_rules.py), the reward / done / observation conventions of
llm_rl_scripts/chess/env/env.py:91-238, the text protocol classes, and games against the engine built from the reference sources."""
import os
import random
import sys

import numpy as np
import pytest

import lmrl_gym_amd  # noqa: F401

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _engine_path():
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "stockfish")
    return p if os.path.exists(p) else None


class ScriptedChessPolicy:
    """Plays a random LEGAL move most of the time, otherwise garbage / an illegal move; actions are spelled as the reference's policies spell
    them: ' '.join(san) + '\\n' (chess/env/env.py:13-14)."""

    def __init__(self, seed, p_bad=0.15):
        self.rng, self.p_bad = random.Random(seed), p_bad

    def act(self, text_history, done=None):
        from test_chess_rules import Board
        from lmrl_gym_amd.envs import chess as C
        from lmrl_gym_amd.environment import Text
        out = []
        for h, d in zip(text_history, done or [False] * len(text_history)):
            if d or h is None:
                out.append(None)
                continue
            fen = C.postprocess_state(h[-1].text)
            legal = [s for _, s in Board(fen).legal()]
            if self.rng.random() < self.p_bad or not legal:
                mv = self.rng.choice(["Ke9", "xx", "Qh5", "e4e5e6", "", "O-O"])
            else:
                mv = self.rng.choice(legal)
            out.append(tuple(h) + (Text(C.preprocess_move(mv), True),))
        return out


def test_device_rules_equal_reference_stockfish_fixtures():
    """The DEVICE kernels against the oracle directly: all positions of tests/golden/chess_perft.json — legal move sets
    (`go perft 1`), positions (`d`) and check flags produced by the Stockfish 15.1 built from the reference's own sources — are walked on the
    device, every game of the fixture in lock step: `lmrl_chess_describe` (legal moves, FEN, status) and `lmrl_chess_opponent_step` (the
    fixture's move, SAN + FEN out).  Then the SAN layer on the device: every legal move's SAN, fed to `lmrl_chess_agent_step` on a fresh
    copy of its position, must play exactly that move (SAN -> move is python-chess behaviour and has no oracle here: round trip only)."""
    import json
    from test_chess_rules import _same_position
    from lmrl_gym_amd.envs import chess as C
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chess_perft.json")))
    games = fx["games"]
    G = len(games)
    boards = C.VectorChessBoards()
    boards.reset([g["fen"] for g in games])
    n_pos = n_check = n_mate = 0
    san_cases = []                      # (fen of the position as the DEVICE prints it, san, uci)
    t = 0
    while True:
        live = [len(g["steps"]) > t for g in games]
        if not any(live):
            break
        moves, status, fens = boards.describe()
        ucis = [""] * G
        for i, g in enumerate(games):
            if not live[i]:
                continue
            step = g["steps"][t]
            assert sorted(u for u, _ in moves[i]) == step["legal"], (g["fen"], t, step["fen"])
            _same_position(fens[i], step["fen"], step["legal"])
            assert bool(status[i] & 1) == step["check"], (step["fen"], status[i])
            assert bool(status[i] & 2) == (step["check"] and not step["legal"])
            assert bool(status[i] & 16) == ((not step["check"]) and not step["legal"])
            n_pos += 1; n_check += step["check"]; n_mate += bool(status[i] & 2)
            san_cases += [(fens[i], sn, u) for u, sn in moves[i]]
            if step["move"] is None:
                assert not moves[i] or (status[i] & 4)
                live[i] = False
            else:
                ucis[i] = step["move"]
        if not any(live):
            break
        sans, _, dones, _ = boards.opponent_step(ucis, live)            # raises if the device rejects a fixture move
        for i, g in enumerate(games):
            if live[i]:
                nxt = g["steps"][t + 1] if t + 1 < len(g["steps"]) else None
                expect = dict(moves[i])[ucis[i]]
                assert sans[i] == expect                              # board.san(move) of the step == the SAN listed for that move
                if nxt is not None:                                   # '+' / '#' suffix == the oracle's check flag of the next position
                    assert sans[i].endswith(("+", "#")) == nxt["check"], (sans[i], nxt["fen"])
                    assert sans[i].endswith("#") == (nxt["check"] and not nxt["legal"])
        t += 1
    assert n_pos == sum(len(g["steps"]) for g in games) >= 2500 and n_check > 50
    # illegal candidates are rejected: for a sample of positions, every from-to pair that is NOT in the oracle's list
    rng = random.Random(5)
    sample = rng.sample([(g["fen"], st) for g in games for st in g["steps"]], 40)
    sq = [f + r for f in "abcdefgh" for r in "12345678"]
    cand, expect = [], []
    for _, st in sample:
        legal = set(st["legal"])
        for a in sq:
            for b in sq:
                if a != b:
                    for promo in ("", "q", "n"):
                        u = a + b + promo
                        if promo and not (a[1] in "27" and b[1] in "18"):
                            continue
                        cand.append((st["fen"], u)); expect.append(u in legal)
    vb = C.VectorChessBoards()
    vb.reset([f for f, _ in cand])
    n = len(cand)
    act = torch.ones(n, dtype=torch.uint8, device=vb.dev)
    from lmrl_gym_amd import _lib
    _lib.check(vb.L.lmrl_chess_opponent_step(_lib.ptr(vb.pos), _lib.ptr(vb._strings([u for _, u in cand], 8)), _lib.ptr(act), _lib.ptr(vb.reward),
                                             _lib.ptr(vb.done), _lib.ptr(vb.ok), _lib.ptr(vb.san_out), _lib.ptr(vb.fen_out), n, _lib.stream_ptr()))
    ok = vb.ok.cpu().numpy().astype(bool)
    bad = [cand[i] for i in range(n) if ok[i] != expect[i]]
    assert not bad, bad[:5]
    assert sum(expect) > 800 and n > 150000
    # SAN round trip on the device, every legal move of every fixture position
    vb = C.VectorChessBoards()
    vb.reset([f for f, _, _ in san_cases])
    res, rew, dn, _, played = vb.agent_step([sn for _, sn, _ in san_cases], [True] * len(san_cases))
    assert len(san_cases) > 50000
    wrong = [san_cases[i] + (played[i],) for i in range(len(san_cases)) if played[i] != san_cases[i][2] or res[i] not in (C.MOVED, C.GAME_OVER)]
    assert not wrong, wrong[:5]
    mates = [i for i in range(len(san_cases)) if san_cases[i][1].endswith("#")]
    assert all(res[i] == C.GAME_OVER and rew[i] == 1.0 and dn[i] for i in mates)


def test_batched_env_equals_host_rules_random_opponent():
    from test_chess_rules import Board
    from lmrl_gym_amd import environment as E
    from lmrl_gym_amd.envs import chess as C
    B = 48
    env = C.VectorChessEnv(max_moves=30, random_opponent=True)
    pol = ScriptedChessPolicy(3)
    np.random.seed(11)
    hist = env.reset([None] * B)
    assert all(h == (E.Text(C.preprocess_state_og(C.START_FEN), False),) for h in hist)
    boards = [Board(C.START_FEN) for _ in range(B)]
    done = [False] * B
    n_steps = n_illegal = n_over = 0
    while not all(done):
        acts = pol.act(hist, done)
        res = env.step(acts, done)
        for i in range(B):
            if done[i]:
                assert res[i] is None
                continue
            (obs,), rew, dn = res[i]
            san = C.postprocess_move(acts[i][-1].text)
            r, hrew, hdone = boards[i].agent(san)
            if r == C.MOVED:                                    # replay the opponent's recorded reply on the host board
                reply = env.moves[i][-1]
                ok, hsan, hrew, hdone = boards[i].opponent(reply)
                assert ok and hsan == env.last_opponent_moves[i]
            else:
                assert env.last_opponent_moves[i] is None
                n_illegal += r == C.ILLEGAL
            assert obs.text == C.preprocess_state_og(boards[i].fen()) and not obs.is_action
            assert rew == hrew
            n_steps += 1
            limit = env.num_moves_made[i] > env.max_moves
            assert dn == (bool(hdone) or limit)
            n_over += bool(hdone)
            done[i] = dn
            hist[i] = res[i][0]
    assert n_steps > B * 20 and n_illegal > B and n_over >= 0
    env.close()


def test_mate_stalemate_null_move_and_text_classes():
    from lmrl_gym_amd.environment import Text
    from lmrl_gym_amd.envs import chess as C
    env = C.FenChessHistoryEnv(from_position="r1bqkbnr/pppp1ppp/2n5/4p2Q/2B1P3/8/PPPP1PPP/RNB1K1NR w KQkq - 4 4", random_opponent=True)
    h = env.reset()
    assert h == (Text(C.preprocess_state_og("r1bqkbnr/pppp1ppp/2n5/4p2Q/2B1P3/8/PPPP1PPP/RNB1K1NR w KQkq - 4 4"), False),)
    h2, rew, done = env.step(h + (Text("Q x f 7 #\n", True),))
    assert rew == 1.0 and done and C.postprocess_state(h2[-1].text).startswith("r1bqkbnr/pppp1Qpp/2n5/4p3/2B1P3/8/PPPP1PPP/RNB1K1NR b KQkq - 0 4")
    env.close()
    env = C.FenChessHistoryEnv(from_position="k7/8/1Q6/8/8/8/8/K7 w - - 0 1", random_opponent=True)
    h = env.reset()
    _, rew, done = env.step(h + (Text("Q c 7\n", True),))
    assert rew == 0.0 and done                                       # stalemate: draw
    h = env.reset()
    _, rew, done = env.step(h + (Text("--\n", True),))
    assert rew == -1.0 and done                                      # null move: -1 and the episode ends (env.py:111-113)
    h = env.reset()
    h2, rew, done = env.step(h + (Text("Q h 1\n", True),))           # illegal: the queen on b6 does not reach h1
    assert rew == -1.0 and not done and h2 == h                      # illegal: -1, same position, episode continues
    env.close()
    ce = C.ChessEnv(from_position="7k/5Q2/8/8/8/8/8/K7 w - - 0 1", random_opponent=True)
    st, info = ce.reset()
    assert st == "7k/5Q2/8/8/8/8/8/K7 w - - 0 1" and info == {}
    np.random.seed(0)
    st, rew, done, info = ce.step("Qf8+")
    assert rew == 0 and done == 0 and info["opponent move"] == "Kh7" and st == "5Q2/7k/8/8/8/8/8/K7 w - - 2 2"
    assert isinstance(ce.sample_valid_action(), str)
    ce.close()
    for pieces in ("kQK", "kRK", "kQRK"):
        fen = C.large_piece_random_endgame(pieces)
        placement = fen.split()[0]
        assert fen.endswith(" w - - 0 1") and sorted(c for c in placement if c.isalpha()) == sorted(pieces)


def test_games_against_the_reference_engine():
    from lmrl_gym_amd.envs import chess as C
    path = _engine_path()
    if path is None:
        pytest.skip("oracle/_ref/stockfish not present")
    from test_chess_rules import Board
    B = 6
    env = C.VectorChessEnv(max_moves=12, engine_path=path, engine_options={"Use NNUE": "false"}, movetime_ms=20)
    pol = ScriptedChessPolicy(5, p_bad=0.0)
    hist = env.reset([None] * B)
    done = [False] * B
    plies = 0
    while not all(done):
        acts = pol.act(hist, done)
        res = env.step(acts, done)
        for i in range(B):
            if done[i]:
                continue
            (obs,), rew, dn = res[i]
            assert rew in (0.0, 1.0, -1.0)
            b = Board(C.postprocess_state(obs.text))              # the observation is a well-formed position with white to move (or game over)
            assert dn or b.fen().split()[1] == "w"
            done[i], hist[i] = dn, res[i][0]
            plies += 1
    assert plies >= B * 5
    # the single-turn text class: initial_history + ' '.join(fen) + '\n' (env.py:188-211)
    from lmrl_gym_amd.environment import Text
    st = C.FenChessHistoryEnvSingleTurn((Text("Your move:\n", False),), engine_path=path, engine_options={"Use NNUE": "false"}, movetime_ms=10)
    h = st.reset()
    assert h == (Text("Your move:\n", False), Text(C.preprocess_state(C.START_FEN), False))
    h2, rew, dn = st.step(h + (Text("e 4\n", True),))
    assert rew == 0.0 and not dn and h2[0] == h[0] and h2[1].text.endswith("\n") and " b " not in C.postprocess_state(h2[1].text)
    st.close()
    # evaluation helper on king+queen endgames against the engine: summary keys of env.py:318-342
    np.random.seed(4)
    positions = [C.large_piece_random_endgame("kQK") for _ in range(2)]
    inter, summ = C.text_env_eval_chess_positions(positions, ScriptedChessPolicy(9, p_bad=0.1), n_rollouts=3, bsize=3, max_moves=6, engine_path=path,
                                                  engine_options={"Use NNUE": "false"}, movetime_ms=10)
    assert len(inter) == 6 and set(summ) == {"reward", "done", "victories", "percent_illegals", "episode_length"}
    env.close()
