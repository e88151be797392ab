"""Generates tests/golden/chess_perft.json with Stockfish 15.1 built from the reference's own sources (oracle/_ref/stockfish, see oracle/Makefile):
random playouts from the initial position and from test positions for castling, en passant, promotion and pins; for every position the legal
move set (`go perft 1`), the FEN and the check flag (`d`).  Run here (needs /root/reference at build time); the JSON is committed."""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import stockfish_uci as S  # noqa: E402

START = "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1"
SEEDS = [START] * 14 + [
    "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1",        # "kiwipete": castling, pins, en passant, promotions nearby
    "8/2p5/3p4/KP5r/1R3p1k/8/4P1P1/8 w - - 0 1",                                   # en passant that would expose the king
    "r3k2r/Pppp1ppp/1b3nbN/nP6/BBP1P3/q4N2/Pp1P2PP/R2Q1RK1 w kq - 0 1",            # promotions, checks
    "rnbq1k1r/pp1Pbppp/2p5/8/2B5/8/PPP1NnPP/RNBQK2R w KQ - 1 8",
    "r4rk1/1pp1qppp/p1np1n2/2b1p1B1/2B1P1b1/P1NP1N2/1PP1QPPP/R4RK1 w - - 0 10",
    "4k3/P6P/8/8/8/8/p6p/4K3 w - - 0 1",
    "8/8/8/3k4/2pP4/8/8/4K3 b - d3 0 1",
    "r3k2r/8/8/8/8/8/8/R3K2R w KQkq - 0 1",
    "8/5k2/8/8/8/8/1q6/K7 w - - 0 1",
    "kQ6/8/1K6/8/8/8/8/8 b - - 0 1",
]


def main():
    eng = S.Engine()
    rng = random.Random(2024)
    games = []
    for fen in SEEDS:
        moves, steps = [], []
        for ply in range(110):
            legal = eng.perft1(fen, moves)
            f, chk = eng.describe(fen, moves)
            mv = rng.choice(legal) if legal else None
            if int(f.split()[4]) >= 100:          # keep clear of the engine's own 50-move handling; the env's 75-move rule has its own test
                mv = None
            steps.append(dict(fen=f, legal=legal, check=chk, move=mv))
            if mv is None:
                break
            moves.append(mv)
        games.append(dict(fen=fen, steps=steps))
    eng.close()
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "chess_perft.json")
    json.dump(dict(engine="Stockfish 15.1 (reference stockfish/src, classical evaluation)", games=games), open(out, "w"), separators=(",", ":"))
    print(out, sum(len(g["steps"]) for g in games), "positions", os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
