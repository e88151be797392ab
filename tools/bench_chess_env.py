"""Batched chess env stepping without the engine (random legal replies chosen on the host beforehand): boards per second of the two
half-step launches of csrc/chess.hip — SAN parse + legality + game-over test, then reply -> SAN, play, FEN — at B games."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import lmrl_gym_amd
from lmrl_gym_amd.envs import chess as C
from test_chess_rules import Board
rng = random.Random(0)
# a pool of (fen, white san, black uci reply) triples from random playouts on the host rules
triples = []
for g in range(40):
    b = Board(C.START_FEN)
    for ply in range(40):
        fen = b.fen()
        lm = b.legal()
        if not lm or (b.status() & 4): break
        u, s = rng.choice(lm)
        b.push_uci(u)
        lm2 = b.legal()
        if not lm2 or (b.status() & 4): break
        u2, _ = rng.choice(lm2)
        if fen.split()[1] == "w":
            triples.append((fen, s, u2))
        b.push_uci(u2)
for B in (1024, 4096, 16384):
    sel = [triples[i % len(triples)] for i in range(B)]
    boards = C.VectorChessBoards()
    reps = 5
    t_tot = 0.0
    for _ in range(reps):
        boards.reset([t[0] for t in sel])
        act = [t[1] for t in sel]; rep = [t[2] for t in sel]
        a_d = boards._strings(act, C.ACTION_BYTES); r_d = boards._strings(rep, 8)
        on = torch.ones(B, dtype=torch.uint8, device=boards.dev)
        from lmrl_gym_amd import _lib
        L = boards.L
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _lib.check(L.lmrl_chess_agent_step(_lib.ptr(boards.pos), _lib.ptr(a_d), _lib.ptr(on), _lib.ptr(boards.reward), _lib.ptr(boards.done), _lib.ptr(boards.result),
                                           _lib.ptr(boards.fen_out), _lib.ptr(boards.uci_out), B, _lib.stream_ptr()))
        _lib.check(L.lmrl_chess_opponent_step(_lib.ptr(boards.pos), _lib.ptr(r_d), _lib.ptr(on), _lib.ptr(boards.reward), _lib.ptr(boards.done), _lib.ptr(boards.ok),
                                              _lib.ptr(boards.san_out), _lib.ptr(boards.fen_out), B, _lib.stream_ptr()))
        torch.cuda.synchronize(); t_tot += time.perf_counter() - t0
    assert int((boards.result == 1).sum()) >= B * 0.9 and int(boards.ok.min()) == 1
    print("B=%6d: agent half-step + opponent half-step %.3f ms per env step of the batch -> %.2f M env-steps/s (device part; the engine thinks 100 ms per move)"
          % (B, t_tot / reps * 1e3, B / (t_tot / reps) / 1e6), flush=True)
