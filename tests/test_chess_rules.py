"""CPU tier: the chess rules of csrc/chess_rules.h (host entry points of liblmrl_amd.so — the same code the device kernels run) against
(a) the committed fixtures produced by Stockfish 15.1 built from the reference's own sources (tests/golden/chess_perft.json: legal move sets
    and positions along random playouts from the initial position and from castling / en-passant / promotion / pin test positions),
(b) the live engine when oracle/_ref/stockfish is present (fresh random playouts),
(c) properties of the SAN layer, which no oracle in this image can pin (python-chess is absent): san(move) -> parse_san round trip over every
    legal move of every visited position, hand-checked SAN strings, the regex corner cases of python-chess parse_san, and
(d) the termination rules (checkmate, stalemate, insufficient material, 75-move rule, fivefold repetition).
Reference: llm_rl_scripts/chess/env/env.py:28-185."""
import ctypes
import json
import os
import random

import pytest

import lmrl_gym_amd  # noqa: F401
from lmrl_gym_amd import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
START = "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1"


class Board:
    """Host view of one position through the C ABI."""

    def __init__(self, fen=START):
        self.L = _lib.lib()
        self.buf = ctypes.create_string_buffer(self.L.lmrl_chess_pos_bytes())
        assert self.L.lmrl_chess_host_from_fen(fen.encode(), self.buf) == 0, fen

    def fen(self):
        out = ctypes.create_string_buffer(96)
        n = self.L.lmrl_chess_host_fen(self.buf, out)
        return out.raw[:n].decode()

    def legal(self):
        u, s = ctypes.create_string_buffer(256 * 8), ctypes.create_string_buffer(256 * 16)
        k = self.L.lmrl_chess_host_legal_moves(self.buf, u, s)
        g = lambda b, i, w: b.raw[i * w:(i + 1) * w].split(b"\0")[0].decode()
        return [(g(u, i, 8), g(s, i, 16)) for i in range(k)]

    def agent(self, san):
        r, d = ctypes.c_float(), ctypes.c_int()
        res = self.L.lmrl_chess_host_agent_step(self.buf, san.encode(), ctypes.byref(r), ctypes.byref(d))
        return res, r.value, d.value

    def opponent(self, uci):
        r, d, s = ctypes.c_float(), ctypes.c_int(), ctypes.create_string_buffer(16)
        ok = self.L.lmrl_chess_host_opponent_step(self.buf, uci.encode(), s, ctypes.byref(r), ctypes.byref(d))
        return ok, s.value.decode(), r.value, d.value

    def status(self):
        return self.L.lmrl_chess_host_status(self.buf)

    def push_uci(self, uci):
        """Play a legal move whatever side is to move (the opponent half-step is colour-agnostic)."""
        ok, san, _, _ = self.opponent(uci)
        assert ok, uci
        return san


def _same_position(mine: str, sf: str, legal_uci):
    a, b = mine.split(), sf.split()
    assert a[:3] == b[:3] and a[4:] == b[4:], (mine, sf)
    if a[3] != b[3]:
        # python-chess shows the en-passant square only when a LEGAL capture exists; Stockfish when an enemy pawn attacks it
        assert a[3] == "-", (mine, sf)
        f = b[3][0]
        assert not any(m[2:4] == b[3] and m[0] != m[2] and abs(ord(m[0]) - ord(f)) == 1 for m in legal_uci), (mine, sf, legal_uci)


def test_fixtures_from_reference_stockfish():
    fx = json.load(open(os.path.join(HERE, "golden", "chess_perft.json")))
    n_pos = 0
    for game in fx["games"]:
        b = Board(game["fen"])
        for step in game["steps"]:
            mine = b.legal()
            assert sorted(u for u, _ in mine) == step["legal"], (game["fen"], step["fen"])
            _same_position(b.fen(), step["fen"], step["legal"])
            assert bool(b.status() & 1) == step["check"]
            n_pos += 1
            if step["move"] is None:
                assert not mine or (b.status() & 4)
                break
            b.push_uci(step["move"])
    assert n_pos >= 1500


def test_live_against_reference_stockfish():
    from oracle import stockfish_uci as S
    if not S.available():
        pytest.skip("oracle/_ref/stockfish not built (needs /root/reference at build time)")
    eng = S.Engine()
    rng = random.Random(123)
    for g in range(12):
        b, moves = Board(START), []
        for ply in range(120):
            legal = b.legal()
            assert sorted(u for u, _ in legal) == eng.perft1(START, moves)
            f, chk = eng.describe(START, moves)
            _same_position(b.fen(), f, [u for u, _ in legal])
            assert bool(b.status() & 1) == chk
            if not legal or (b.status() & 4):
                break
            u, san = rng.choice(legal)
            b.push_uci(u)
            moves.append(u)
    eng.close()


def test_san_round_trip_and_conventions():
    rng = random.Random(7)
    seen_kinds = set()
    for g in range(40):
        b = Board(START)
        for ply in range(150):
            legal = b.legal()
            if not legal or (b.status() & 4):
                break
            fen = b.fen()
            for u, san in legal:                              # every legal move: its SAN parses back to exactly that move
                c = Board(fen)
                before = c.legal()
                if fen.split()[1] == "w":
                    res, _, _ = c.agent(san)
                    assert res in (1, 2), (fen, u, san, res)
                    d = Board(fen); d.push_uci(u)
                    assert c.fen() == d.fen(), (fen, u, san)
                assert sum(1 for _, s2 in before if s2 == san) == 1, (fen, san)      # SAN strings are unique within a position
                if "=" in san: seen_kinds.add("promo")
                if san.startswith("O-O"): seen_kinds.add("castle")
                if san.endswith("#"): seen_kinds.add("mate")
                if san.endswith("+"): seen_kinds.add("check")
                if len(san) >= 4 and san[0] in "NBRQ" and san[1] in "abcdefgh12345678" and san[2] in "abcdefghx":
                    seen_kinds.add("disambiguation")
            u, _ = rng.choice(legal)
            b.push_uci(u)
    assert {"castle", "check", "disambiguation"} <= seen_kinds, seen_kinds
    # hand-checked strings
    b = Board("r1bqkbnr/pppp1ppp/2n5/4p2Q/2B1P3/8/PPPP1PPP/RNB1K1NR w KQkq - 4 4")
    assert dict(b.legal())["h5f7"] == "Qxf7#"
    b = Board("4k3/P7/8/8/8/8/8/4K3 w - - 0 1")
    assert dict(b.legal())["a7a8q"] == "a8=Q+" and dict(b.legal())["a7a8n"] == "a8=N"
    b = Board("rnbqkbnr/ppp1p1pp/8/3pPp2/8/8/PPPP1PPP/RNBQKBNR w KQkq f6 0 3")
    assert dict(b.legal())["e5f6"] == "exf6" and b.fen().split()[3] == "f6"
    b = Board("8/8/8/8/8/2k5/8/N1K1N3 w - - 0 1")           # two knights reach c2: file disambiguation
    sans = dict(b.legal())
    assert sans["a1c2"] == "Nac2" and sans["e1c2"] == "Nec2"
    b = Board("4k3/8/8/8/R7/8/8/R3K3 w - - 0 1")             # two rooks on the a-file reach a3: rank disambiguation
    sans = dict(b.legal())
    assert sans["a1a3"] == "R1a3" and sans["a4a3"] == "R4a3"
    b = Board("r3k2r/8/8/8/8/8/8/R3K2R w KQkq - 0 1")
    sans = dict(b.legal())
    assert sans["e1g1"] == "O-O" and sans["e1c1"] == "O-O-O"
    # parse_san corner cases (python-chess 1.x): 0-0 spelling, optional check marks, lower-case / '='-less promotion, capture without 'x',
    # long form with origin square, pawn capture needs its file, ambiguous and illegal moves, null move
    for san, want in (("0-0", 1), ("O-O+", 1), ("O-O-O#", 1), ("Ra1a2", 1), ("Ra2", 1), ("Kd1", 1), ("Ke2+", 1), ("Rxa8", 1), ("Ra8", 1), ("Rhf1", 1),
                      ("Rf1", 1), ("Rd1", 1), ("e4", 0), ("Rb1b2", 0), ("--", 3), ("Z0", 3), ("", 0), ("Ke1", 0), ("O-O-O-O", 0), ("Qd1", 0)):
        c = Board("r3k2r/8/8/8/8/8/8/R3K2R w KQkq - 0 1")
        res, rew, done = c.agent(san)
        if want == 0:
            assert res == 0 and rew == -1.0 and done == 0 and c.fen() == "r3k2r/8/8/8/8/8/8/R3K2R w KQkq - 0 1", san
        elif want == 3:
            assert res == 3 and rew == -1.0 and done == 1, san
        else:
            assert res in (1, 2), san
    c = Board("4k3/8/8/8/8/4K3/8/R6R w - - 0 1")                                    # both rooks reach d1: ambiguous -> ValueError -> illegal
    assert c.agent("Rd1") == (0, -1.0, 0) and c.agent("Rad1")[0] == 1
    c = Board("4k3/P7/8/8/8/8/8/4K3 w - - 0 1")
    for san in ("a8q", "a8=q", "a8Q", "a8=Q+", "a8=Q"):
        d = Board("4k3/P7/8/8/8/8/8/4K3 w - - 0 1")
        assert d.agent(san)[0] in (1, 2), san
    assert c.agent("a8")[0] == 0 and c.agent("a8=K")[0] == 0                     # promotion piece missing / impossible
    c = Board("rnbqkbnr/ppp1pppp/8/3p4/4P3/8/PPPP1PPP/RNBQKBNR w KQkq - 0 2")
    assert c.agent("d5")[0] == 0                                                  # a pawn capture must name its file
    assert c.agent("exd5")[0] == 1


def test_termination_rules():
    # checkmate by the agent: reward 1, done; stalemate: reward 0, done
    b = Board("r1bqkbnr/pppp1ppp/2n5/4p2Q/2B1P3/8/PPPP1PPP/RNB1K1NR w KQkq - 4 4")
    assert b.agent("Qxf7#") == (2, 1.0, 1) and b.status() & 2
    b = Board("7k/5Q2/8/8/8/8/8/K7 w - - 0 1")
    assert b.agent("Qf8+")[0] == 1
    b = Board("7k/5K2/6Q1/8/8/8/8/8 w - - 0 1")              # Qg6-h6?? no: Kf7 + Qg6 already stalemates after a waiting move
    res, rew, done = b.agent("Qg5")
    assert (res, rew, done) == (1, 0.0, 0)
    b = Board("k7/8/1Q6/8/8/8/8/K7 w - - 0 1")
    assert b.agent("Qc7") == (2, 0.0, 1) and b.status() & 16                      # stalemate
    # the opponent mates the agent: reward -1, done
    b = Board("rnb1kbnr/pppp1ppp/8/4p3/6Pq/5P2/PPPPP2P/RNBQKBNR w KQkq - 1 3")
    assert b.status() & 2
    b = Board("rnbqkbnr/pppp1ppp/8/4p3/6P1/5P2/PPPPP2P/RNBQKBNR b KQkq - 0 2")
    ok, san, rew, done = b.opponent("d8h4")
    assert ok and san == "Qh4#" and rew == -1.0 and done == 1
    # insufficient material: K v K, K+N v K, K+B v K+B same colour; not K+B v K+B opposite colours, not K+N v K+N... (python-chess rules)
    for fen, want in (("8/8/8/4k3/8/8/8/4K3 w - - 0 1", True), ("8/8/8/4k3/8/8/6N1/4K3 w - - 0 1", True), ("8/8/8/4k3/8/8/6B1/4K3 w - - 0 1", True),
                      ("8/8/8/4kb2/8/8/6B1/4K3 w - - 0 1", True), ("8/8/8/4k1b1/8/8/6B1/4K3 w - - 0 1", False),
                      ("8/8/8/4kn2/8/8/6N1/4K3 w - - 0 1", False), ("8/8/8/4k3/8/8/6P1/4K3 w - - 0 1", False),
                      ("8/8/8/4k3/8/8/5NN1/4K3 w - - 0 1", False)):
        assert bool(Board(fen).status() & 8) == want, fen
    # 75-move rule: half-move clock 150 ends the game
    assert Board("8/8/8/4k3/8/8/6R1/4K3 w - - 149 90").status() & 4 == 0
    assert Board("8/8/8/4k3/8/8/6R1/4K3 w - - 150 90").status() & 4
    # fivefold repetition: shuffle the rooks back and forth; the position after white's move recurs every 4 plies
    b = Board("4k2r/8/8/8/8/8/8/R3K3 w - - 0 1")
    cycle = ["a1b1", "h8g8", "b1a1", "g8h8"]
    over_at = None
    for i in range(40):
        b.push_uci(cycle[i % 4])
        if b.status() & 32:
            over_at = i + 1
            break
    assert over_at == 16, over_at              # the start position (after ply 0, 4, 8, 12, 16) occurs for the 5th time after ply 16
    assert b.status() & 4


def test_parse_san_never_misbehaves_on_arbitrary_text():
    """A policy may emit anything: every string is either one of the position's SAN moves (possibly spelled differently) or rejected with
    reward -1 / not done / board untouched (chess/env/env.py:109-118)."""
    rng = random.Random(99)
    alphabet = "abcdefgh12345678NBRQKOx=+#-0 Z"
    fens = [START, "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1", "4k3/P6P/8/8/8/8/p6p/4K3 w - - 0 1"]
    n_ok = n_bad = 0
    for fen in fens:
        sans = {s for _, s in Board(fen).legal()}
        for _ in range(3000):
            text = "".join(rng.choice(alphabet) for _ in range(rng.randint(0, 9)))
            b = Board(fen)
            res, rew, done = b.agent(text)
            if res in (1, 2):
                n_ok += 1
                assert text.strip() and b.fen() != fen
                reach = set()
                for u, _ in Board(fen).legal():                 # an accepted string resolves to one of the legal moves
                    c = Board(fen); c.push_uci(u); reach.add(c.fen())
                assert b.fen() in reach
            elif res == 3:
                assert text in ("--", "Z0") and rew == -1.0 and done == 1
            else:
                n_bad += 1
                assert (rew, done) == (-1.0, 0) and b.fen() == fen
                assert text not in sans
    assert n_ok >= 1 and n_bad > 8000


def test_engine_options_limit_strength_like_the_stockfish_package():
    """the reference reaches Stockfish through the python `stockfish` package, whose update_engine_parameters turns
    UCI_LimitStrength on when only UCI_Elo is given (env.py:55-57) — without it Stockfish ignores UCI_Elo and plays at full strength."""
    from lmrl_gym_amd.envs import chess as C
    o = C.normalise_uci_options({"Threads": 1, "UCI_Elo": 1200})
    assert list(o.items()) == [("UCI_LimitStrength", "true"), ("Threads", 1), ("UCI_Elo", 1200)]        # the switch goes out first
    assert C.normalise_uci_options({"Skill Level": 3})["UCI_LimitStrength"] == "false"
    assert "UCI_LimitStrength" not in C.normalise_uci_options({"Threads": 2})
    assert "UCI_LimitStrength" not in C.normalise_uci_options({"Skill Level": 3, "UCI_Elo": 1500})      # both given: the package leaves it alone
    assert list(C.normalise_uci_options({"UCI_Elo": 1500, "UCI_LimitStrength": "false"}).items())[0] == ("UCI_LimitStrength", "false")
    from oracle import stockfish_uci as S
    if S.available():
        eng = C.UCIEngine(S.BINARY, {"Threads": 1, "UCI_Elo": 1350, "Use NNUE": "false"})
        assert eng.sent_options[0] == "setoption name UCI_LimitStrength value true" and "setoption name UCI_Elo value 1350" in eng.sent_options
        assert len(eng.best_move_time(START, [], 10)) in (4, 5)
        eng.close()
