"""Test infrastructure: a thin UCI client for the Stockfish binary built from the reference's own sources (oracle/Makefile -> oracle/_ref/stockfish;
the chess env's opponent engine, llm_rl_scripts/chess/env/env.py:157-170).  Used as the ORACLE of the chess rules — legal move sets through
`go perft 1`, positions and checkers through `d` — and as the opponent in the chess env tests.  Never imported by the product path."""
from __future__ import annotations

import os
import subprocess
from typing import Dict, List, Optional, Tuple

BINARY = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "stockfish")


def available() -> bool:
    return os.path.exists(BINARY) and os.access(BINARY, os.X_OK)


class Engine:
    def __init__(self, path: str = BINARY, elo: Optional[int] = None):
        self.p = subprocess.Popen([path], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, bufsize=1)
        self._cmd("uci", "uciok")
        # the NNUE net file is not in the reference tree: classical evaluation (same move generator)
        self._send("setoption name Use NNUE value false")
        self._send("setoption name Threads value 1")
        if elo is not None:
            self._send("setoption name UCI_LimitStrength value true")
            self._send(f"setoption name UCI_Elo value {max(1350, int(elo))}")       # the engine's floor is 1350
        self._cmd("isready", "readyok")

    def _send(self, s: str):
        self.p.stdin.write(s + "\n")
        self.p.stdin.flush()

    def _cmd(self, s: str, until: str) -> List[str]:
        self._send(s)
        out = []
        while True:
            line = self.p.stdout.readline()
            if not line:
                raise RuntimeError("stockfish terminated")
            line = line.rstrip("\n")
            out.append(line)
            if line.startswith(until):
                return out

    def _position(self, fen: str, moves: List[str]):
        self._send(f"position fen {fen}" + (" moves " + " ".join(moves) if moves else ""))

    def perft1(self, fen: str, moves: List[str] = ()) -> List[str]:
        """Legal moves (UCI) of the position reached from `fen` by `moves`."""
        self._position(fen, list(moves))
        lines = self._cmd("go perft 1", "Nodes searched")
        return sorted(l.split(":")[0] for l in lines if ":" in l and not l.startswith("Nodes") and not l.startswith("info"))

    def describe(self, fen: str, moves: List[str] = ()) -> Tuple[str, bool]:
        """(FEN, in check?) of the position reached from `fen` by `moves`."""
        self._position(fen, list(moves))
        lines = self._cmd("d", "Checkers")
        f = [l for l in lines if l.startswith("Fen: ")][0][5:].strip()
        chk = [l for l in lines if l.startswith("Checkers:")][0][len("Checkers:"):].strip()
        return f, bool(chk)

    def bestmove(self, fen: str, moves: List[str] = (), movetime_ms: int = 100) -> str:
        self._position(fen, list(moves))
        lines = self._cmd(f"go movetime {movetime_ms}", "bestmove")
        return lines[-1].split()[1]

    def close(self):
        try:
            self._send("quit")
            self.p.wait(timeout=2)
        except Exception:
            self.p.kill()

    def __del__(self):
        self.close()
