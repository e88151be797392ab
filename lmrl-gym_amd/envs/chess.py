"""Chess env — counterpart of llm_rl_scripts/chess/env/env.py (ChessEnv, FenChessHistoryEnv, FenChessHistoryEnvSingleTurn,
text_env_eval_chess_positions, large_piece_random_endgame and the text (de)formatters).

The reference steps ONE python-chess board per env and asks a Stockfish subprocess for the reply (100 ms per move).  Here the boards of a batch
live in device memory and both half-steps of `ChessEnv.step` — SAN parsing + legality + game-over test for the agent's move, then the
engine's reply (UCI) turned into SAN, played, and the FEN observation rendered — run for all games in one launch each (csrc/chess.hip);
the engine replies come from a pool of UCI processes queried in parallel.  python-chess is not needed.

The engine binary is the user's (`CHESS_ENGINE_PATH`, as in the reference: env.py:11); `options` are UCI options sent at start-up
(the reference passes {"Threads": 1, "UCI_Elo": elo} through the `stockfish` package, env.py:55-57).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Dict, Iterator, List, Optional, Tuple, Union

import numpy as np

from .. import _lib
from ..environment import (BatchedTextEnv, BatchedTextPolicy, Text, TextEnv, TextHistory, TextPolicy, interact_environment)

START_FEN = "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1"
FEN_BYTES, ACTION_BYTES = 96, 16
ILLEGAL, MOVED, GAME_OVER, NULL_MOVE = 0, 1, 2, 3


# ----------------------------------------------------------------------------- text (de)formatting (env.py:13-26)
def preprocess_move(move: str) -> str:
    return " ".join(move) + "\n"


def postprocess_move(move: str) -> str:
    return move.replace(" ", "").strip()


def preprocess_state(state: str) -> str:
    return " ".join(state) + "\n"


def preprocess_state_og(state: str) -> str:
    return " ".join(state)


def postprocess_state(state: str) -> str:
    return state.replace("  ", "__temp__").replace(" ", "").replace("__temp__", " ").strip()


# ----------------------------------------------------------------------------- opponent engines
def normalise_uci_options(options: Optional[Dict[str, Union[str, int]]]) -> Dict[str, Union[str, int]]:
    """What the python `stockfish` package the reference goes through (env.py:55-57: `Stockfish(path, parameters={"Threads", "UCI_Elo"})`)
    does in `update_engine_parameters`: when exactly one of {"Skill Level", "UCI_Elo"} is given without an explicit "UCI_LimitStrength", it
    sets UCI_LimitStrength itself — "true" for UCI_Elo (Stockfish ignores UCI_Elo otherwise and would play at full strength), "false" for
    Skill Level.  The switch is sent BEFORE the value it enables."""
    opts = dict(options or {})
    if (("Skill Level" in opts) != ("UCI_Elo" in opts)) and "UCI_LimitStrength" not in opts:
        limit = "true" if "UCI_Elo" in opts else "false"
        opts = {"UCI_LimitStrength": limit, **opts}
    elif "UCI_LimitStrength" in opts:
        opts = {"UCI_LimitStrength": opts.pop("UCI_LimitStrength"), **opts}
    return opts


class UCIEngine:
    """One UCI engine process (Stockfish).  Stateless between calls: every query sends `position fen <start> moves ...`."""

    def __init__(self, path: str, options: Optional[Dict[str, Union[str, int]]] = None):
        self.p = subprocess.Popen([path], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, bufsize=1)
        self._cmd("uci", "uciok")
        self.sent_options = []
        for k, v in normalise_uci_options(options).items():
            self.sent_options.append(f"setoption name {k} value {v}")
            self._send(self.sent_options[-1])
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
                raise RuntimeError("chess engine terminated")
            out.append(line.rstrip("\n"))
            if out[-1].startswith(until):
                return out

    def best_move_time(self, start_fen: str, moves: List[str], movetime_ms: int = 100) -> str:
        """`Stockfish.get_best_move_time(100)` (env.py:161) on the game `start_fen` + `moves`."""
        self._send(f"position fen {start_fen}" + (" moves " + " ".join(moves) if moves else ""))
        return self._cmd(f"go movetime {int(movetime_ms)}", "bestmove")[-1].split()[1]

    def close(self):
        try:
            self._send("quit")
            self.p.wait(timeout=2)
        except Exception:
            self.p.kill()


class EnginePool:
    """`workers` engine processes answering a batch of positions in parallel (one thread per process: the processes do the thinking)."""

    def __init__(self, path: Optional[str] = None, options: Optional[Dict[str, Union[str, int]]] = None, workers: int = 8, movetime_ms: int = 100):
        self.path = path or os.environ.get("CHESS_ENGINE_PATH")
        if not self.path or not os.path.exists(self.path):
            raise FileNotFoundError("chess engine binary not found: pass `engine_path` or set CHESS_ENGINE_PATH (the reference's "
                                    "stockfish/stockfish-ubuntu-20.04-x86-64-avx2, env.py:11)")
        self.options, self.workers, self.movetime_ms = dict(options or {}), workers, movetime_ms
        self.engines: List[UCIEngine] = []
        self.pool = ThreadPoolExecutor(max_workers=workers)

    def best_moves(self, games: List[Tuple[str, List[str]]]) -> List[str]:
        while len(self.engines) < min(self.workers, len(games)):
            self.engines.append(UCIEngine(self.path, self.options))
        n = len(self.engines)
        chunks = [list(range(i, len(games), n)) for i in range(n)]

        def run(w):
            return [(i, self.engines[w].best_move_time(games[i][0], games[i][1], self.movetime_ms)) for i in chunks[w]]
        out = [None] * len(games)
        for res in self.pool.map(run, range(n)):
            for i, mv in res:
                out[i] = mv
        return out

    def close(self):
        for e in self.engines:
            e.close()
        self.engines = []
        self.pool.shutdown(wait=False)


# ----------------------------------------------------------------------------- batched boards on the device
class VectorChessBoards:
    """N chess games in device memory (csrc/chess.hip).  `agent_step` / `opponent_step` are the two halves of ChessEnv.step (env.py:91-143)."""

    def __init__(self):
        import torch
        self.t, self.dev, self.L = torch, _lib.require_gpu(), _lib.lib()
        self.n = 0

    def _strings(self, items: List[str], pitch: int):
        buf = np.zeros((len(items), pitch), dtype=np.uint8)
        for i, s in enumerate(items):
            b = s.encode("ascii", errors="replace")[: pitch - 1]
            buf[i, : len(b)] = np.frombuffer(b, dtype=np.uint8)
        return self.t.from_numpy(buf).to(self.dev)

    @staticmethod
    def _unpack(tensor, pitch: int) -> List[str]:
        raw = tensor.cpu().numpy().tobytes()
        return [raw[i * pitch:(i + 1) * pitch].split(b"\0")[0].decode("ascii") for i in range(len(raw) // pitch)]

    def reset(self, fens: List[str]):
        t, n = self.t, len(fens)
        self.n = n
        self.pos = t.zeros(n * self.L.lmrl_chess_pos_bytes(), dtype=t.uint8, device=self.dev)
        ok = t.zeros(n, dtype=t.uint8, device=self.dev)
        _lib.check(self.L.lmrl_chess_reset(_lib.ptr(self.pos), _lib.ptr(self._strings(fens, FEN_BYTES)), _lib.ptr(ok), n, _lib.stream_ptr()), "lmrl_chess_reset")
        bad = [fens[i] for i in np.nonzero(ok.cpu().numpy() == 0)[0]]
        if bad:
            raise ValueError(f"invalid FEN: {bad[0]!r}")
        z = lambda dt: t.zeros(n, dtype=dt, device=self.dev)
        self.reward, self.done, self.result, self.ok = z(t.float32), z(t.uint8), z(t.uint8), z(t.uint8)
        self.fen_out = t.zeros(n * FEN_BYTES, dtype=t.uint8, device=self.dev)
        self.san_out = t.zeros(n * ACTION_BYTES, dtype=t.uint8, device=self.dev)
        self.uci_out = t.zeros(n * 8, dtype=t.uint8, device=self.dev)

    def agent_step(self, actions: List[str], active: List[bool]):
        """-> (result codes, rewards, dones, FENs, played moves in UCI form) per game; inactive games: result 255."""
        act = self.t.from_numpy(np.asarray(active, dtype=np.uint8)).to(self.dev)
        _lib.check(self.L.lmrl_chess_agent_step(_lib.ptr(self.pos), _lib.ptr(self._strings(actions, ACTION_BYTES)), _lib.ptr(act), _lib.ptr(self.reward),
                                                _lib.ptr(self.done), _lib.ptr(self.result), _lib.ptr(self.fen_out), _lib.ptr(self.uci_out), self.n,
                                                _lib.stream_ptr()), "lmrl_chess_agent_step")
        return (self.result.cpu().numpy().copy(), self.reward.cpu().numpy().copy(), self.done.cpu().numpy().astype(bool), self._unpack(self.fen_out, FEN_BYTES),
                self._unpack(self.uci_out, 8))

    def opponent_step(self, ucis: List[str], active: List[bool]):
        """-> (SAN of the replies, rewards, dones, FENs); raises if an engine move is not legal on our board."""
        act = self.t.from_numpy(np.asarray(active, dtype=np.uint8)).to(self.dev)
        _lib.check(self.L.lmrl_chess_opponent_step(_lib.ptr(self.pos), _lib.ptr(self._strings(ucis, 8)), _lib.ptr(act), _lib.ptr(self.reward), _lib.ptr(self.done),
                                                   _lib.ptr(self.ok), _lib.ptr(self.san_out), _lib.ptr(self.fen_out), self.n, _lib.stream_ptr()),
                   "lmrl_chess_opponent_step")
        ok = self.ok.cpu().numpy()
        if (ok == 0).any():
            i = int(np.nonzero(ok == 0)[0][0])
            raise RuntimeError(f"engine move {ucis[i]!r} is not legal in game {i}")
        return (self._unpack(self.san_out, ACTION_BYTES), self.reward.cpu().numpy().copy(), self.done.cpu().numpy().astype(bool),
                self._unpack(self.fen_out, FEN_BYTES))

    def describe(self, want_san: bool = True):
        """Every game's legal moves (UCI [+ SAN]), status bits (bit0 check, bit1 checkmate, bit2 game over, bit3 insufficient material,
        bit4 stalemate, bit5 fivefold, bit6 75-move) and FEN, generated on the device in one launch (`lmrl_chess_describe`).
        -> (moves: List[List[(uci, san)]], status: np.uint8[n], fens: List[str])"""
        t, n, L = self.t, self.n, self.L
        mm = L.lmrl_chess_max_moves()
        uci = t.zeros(n * mm * 8, dtype=t.uint8, device=self.dev)
        san = t.zeros(n * mm * ACTION_BYTES, dtype=t.uint8, device=self.dev) if want_san else None
        cnt = t.zeros(n, dtype=t.int32, device=self.dev)
        status = t.zeros(n, dtype=t.uint8, device=self.dev)
        _lib.check(L.lmrl_chess_describe(_lib.ptr(self.pos), _lib.ptr(uci), _lib.ptr(san), _lib.ptr(cnt), _lib.ptr(status), _lib.ptr(self.fen_out), n,
                                         _lib.stream_ptr()), "lmrl_chess_describe")
        cnt_h = cnt.cpu().numpy()
        u = uci.cpu().numpy().reshape(n, mm, 8)
        sn = san.cpu().numpy().reshape(n, mm, ACTION_BYTES) if want_san else None
        cut = lambda row: row.tobytes().split(b"\0")[0].decode("ascii")
        moves = [[(cut(u[i, j]), cut(sn[i, j]) if want_san else "") for j in range(int(cnt_h[i]))] for i in range(n)]
        return moves, status.cpu().numpy().copy(), self._unpack(self.fen_out, FEN_BYTES)

    def host_position(self, i: int) -> ctypes.Array:
        """Host copy of game i's position (for the host faces of the rules: legal move lists for the random opponent)."""
        nb = self.L.lmrl_chess_pos_bytes()
        raw = self.pos[i * nb:(i + 1) * nb].cpu().numpy().tobytes()
        return ctypes.create_string_buffer(raw, nb)

    def legal_moves(self, i: int) -> List[Tuple[str, str]]:
        """[(uci, san)] of game i (board.legal_moves / board.san)."""
        buf = self.host_position(i)
        u, s = ctypes.create_string_buffer(256 * 8), ctypes.create_string_buffer(256 * ACTION_BYTES)
        k = self.L.lmrl_chess_host_legal_moves(buf, u, s)
        g = lambda b, j, w: b.raw[j * w:(j + 1) * w].split(b"\0")[0].decode()
        return [(g(u, j, 8), g(s, j, ACTION_BYTES)) for j in range(k)]


class VectorChessEnv(BatchedTextEnv):
    """`bsize` FenChessHistoryEnv games in lock step (env.py:213-238): observation = ' '.join(fen) of the position after the opponent's reply,
    reward / done as ChessEnv.step; `max_moves` as the reference (done once more than max_moves agent moves were made)."""

    def __init__(self, max_moves: int = 400, from_position: Optional[str] = None, random_opponent: bool = False, engine: Optional[EnginePool] = None,
                 engine_path: Optional[str] = None, engine_options: Optional[Dict[str, Union[str, int]]] = None, stockfish_elo: int = 1200,
                 movetime_ms: int = 100, state_fn: Callable[[str], str] = preprocess_state_og, initial_history: TextHistory = ()):
        self.max_moves, self.from_position, self.random_opponent = max_moves, from_position, random_opponent
        self.start = from_position or START_FEN
        self._own_engine = engine is None and not random_opponent
        opts = {"Threads": 1, "UCI_Elo": stockfish_elo}                    # env.py:55-57
        opts.update(engine_options or {})
        self.engine = engine if engine is not None else (None if random_opponent else EnginePool(engine_path, opts, movetime_ms=movetime_ms))
        self.state_fn, self.initial_history = state_fn, tuple(initial_history)
        self.boards = VectorChessBoards()
        self.n = 0

    def reset(self, seed=None, options=None) -> List[TextHistory]:
        n = len(seed) if seed is not None else (len(options) if options is not None else 1)
        self.n = n
        starts = [(o or {}).get("from_position", self.start) if options is not None else self.start for o in (options or [None] * n)]
        self.starts, self.moves, self.num_moves_made = starts, [[] for _ in range(n)], [0] * n
        self.boards.reset(starts)
        self.last_fen = list(starts)                 # the reference shows the start position string as given (env.py:223-226)
        return [self.initial_history + (Text(self.state_fn(s), False),) for s in starts]

    def step(self, text_history, done=None):
        n = self.n
        assert n > 0 and len(text_history) == n
        done = done or [False] * n
        active = [not d and h is not None for h, d in zip(text_history, done)]
        actions = []
        for h, a in zip(text_history, active):
            if a:
                assert h[-1].is_action
            actions.append(postprocess_move(h[-1].text) if a else "")
        res, rew, dn, fens, played = self.boards.agent_step(actions, active)
        need = [bool(a and r == MOVED) for a, r in zip(active, res)]
        opp_san = [None] * n
        if any(need):
            idx = [i for i in range(n) if need[i]]
            for i in idx:
                self.moves[i].append(played[i])                    # the engine follows the game as UCI moves (env.py:121)
            if self.random_opponent:
                all_moves, _, _ = self.boards.describe(want_san=False)         # list(board.legal_moves) of every game, one device launch
                replies = []
                for i in idx:
                    lm = all_moves[i]
                    replies.append(lm[int(np.random.choice(len(lm), 1)[0])][0])          # np.random.choice(legal_moves, 1)[0]   env.py:178
            else:
                replies = self.engine.best_moves([(self.starts[i], self.moves[i]) for i in idx])
            ucis = [""] * n
            for i, mv in zip(idx, replies):
                ucis[i] = mv
                self.moves[i].append(mv)
            sans, rew2, dn2, fens2 = self.boards.opponent_step(ucis, need)
            for i in idx:
                opp_san[i], rew[i], dn[i], fens[i] = sans[i], rew2[i], dn2[i], fens2[i]
        out = []
        for i in range(n):
            if not active[i]:
                out.append(None)
                continue
            self.num_moves_made[i] += 1
            d = bool(dn[i]) or self.num_moves_made[i] > self.max_moves
            self.last_fen[i] = fens[i]
            out.append((self.initial_history + (Text(self.state_fn(fens[i]), False),), float(rew[i]), d))
        self.last_opponent_moves = opp_san
        return out

    def close(self):
        if self._own_engine and self.engine is not None:
            self.engine.close()

    def copy(self):
        return VectorChessEnv(self.max_moves, self.from_position, self.random_opponent, None if self._own_engine else self.engine,
                              getattr(self.engine, "path", None), getattr(self.engine, "options", None), state_fn=self.state_fn,
                              initial_history=self.initial_history)


# ----------------------------------------------------------------------------- the reference's single-env classes
class ChessEnv:
    """env.py:28-185 on one device board.  `step(action)` -> (fen, reward, done, {"opponent move": san})."""
    metadata = {"render.modes": ["human"]}

    def __init__(self, side="w", fen=True, from_position=None, stockfish_level=3, stockfish_elo=1200, random_opponent=False, engine: Optional[EnginePool] = None,
                 engine_path: Optional[str] = None, engine_options: Optional[Dict[str, Union[str, int]]] = None):
        assert side == "w", "the agent plays white (env.py:105 asserts it)"
        self.starting_position = from_position or START_FEN
        self.fen = fen
        self.prev_moves: List[str] = []
        self._vec = VectorChessEnv(max_moves=1 << 30, from_position=self.starting_position, random_opponent=random_opponent, engine=engine,
                                   engine_path=engine_path, engine_options=engine_options, stockfish_elo=stockfish_elo, state_fn=lambda s: s)
        self._vec.reset([None])

    def reset(self):
        self._vec.reset([None])
        self.prev_moves = []
        return self.starting_position, {}

    def step(self, action: str, opponent_move: bool = True):
        assert opponent_move, "every caller in the reference plays with the opponent's reply"
        self.prev_moves.append(action)
        (hist, reward, done), = self._vec.step([(Text(action, True),)])
        return hist[-1].text, reward, int(done), {"opponent move": self._vec.last_opponent_moves[0]}

    def _get_state(self):
        return self._vec.last_fen[0] if self.fen else None

    def get_board(self):
        """The reference returns the python-chess Board; here: the FEN of the current position."""
        return self._vec.last_fen[0]

    def sample_valid_action(self):
        lm = self._vec.boards.legal_moves(0)
        return lm[int(np.random.choice(len(lm), 1)[0])][1]

    def render(self, mode="human", close=False):
        rows = self._vec.last_fen[0].split()[0].split("/")
        txt = "\n".join(" ".join("".join("." * int(c) if c.isdigit() else c for c in row)) for row in rows)
        print(txt)
        return txt

    def close(self):
        self._vec.close()


class _SingleFromVector(TextEnv):
    def __init__(self, vec: VectorChessEnv):
        self._vec = vec

    def reset(self, seed: Optional[int] = None, options: Optional[Dict] = None):
        return self._vec.reset([seed], [options] if options is not None else None)[0]

    def step(self, text_history: TextHistory):
        return self._vec.step([text_history])[0]

    def close(self):
        self._vec.close()

    def as_batched(self) -> VectorChessEnv:
        return self._vec.copy()


class FenChessHistoryEnv(_SingleFromVector):
    """env.py:213-238: observation (Text(' '.join(fen), False),) — only the current position, no move history."""

    def __init__(self, max_moves=400, from_position=None, random_opponent=False, **engine_kw):
        self._args = (max_moves, from_position, random_opponent, engine_kw)
        super().__init__(VectorChessEnv(max_moves, from_position, random_opponent, state_fn=preprocess_state_og, **engine_kw))
        self.max_moves, self.from_position = max_moves, from_position

    def copy(self):
        return FenChessHistoryEnv(self._args[0], self._args[1], self._args[2], **self._args[3])


class FenChessHistoryEnvSingleTurn(_SingleFromVector):
    """env.py:188-211: `initial_history` + (Text(' '.join(fen) + '\\n', False),)."""

    def __init__(self, initial_history: TextHistory, max_moves=400, from_position=None, **engine_kw):
        self._args = (initial_history, max_moves, from_position, engine_kw)
        super().__init__(VectorChessEnv(max_moves, from_position, False, state_fn=preprocess_state, initial_history=initial_history, **engine_kw))
        self.initial_history, self.max_moves, self.from_position = initial_history, max_moves, from_position

    def copy(self):
        return FenChessHistoryEnvSingleTurn(self._args[0], self._args[1], self._args[2], **self._args[3])


def large_piece_random_endgame(pieces: str) -> str:
    """env.py:240-255: random placement of `pieces` (e.g. 'kQK') until the position is valid and the side to move (white) is not in check.
    Validity as python-chess `Board.is_valid()` for such material: one king each, and the side NOT to move is not in check."""
    L = _lib.lib()
    buf = ctypes.create_string_buffer(L.lmrl_chess_pos_bytes())
    while True:
        board = [None] * 64
        possible = np.arange(0, 64)
        for piece in pieces:
            sq = int(np.random.choice(possible))
            board[sq] = piece
            possible = possible[possible != sq]
        rows = []
        for r in range(7, -1, -1):
            row, run = "", 0
            for f in range(8):
                c = board[r * 8 + f]
                if c is None:
                    run += 1
                else:
                    row += (str(run) if run else "") + c
                    run = 0
            rows.append(row + (str(run) if run else ""))
        placement = "/".join(rows)
        if placement.count("K") != 1 or placement.count("k") != 1 or any(c in "Pp" for c in rows[0] + rows[7]):
            continue
        fen_w, fen_b = placement + " w - - 0 1", placement + " b - - 0 1"
        L.lmrl_chess_host_from_fen(fen_b.encode(), buf)
        black_in_check = bool(L.lmrl_chess_host_status(buf) & 1)           # "opposite check": white to move while black is in check is invalid
        L.lmrl_chess_host_from_fen(fen_w.encode(), buf)
        white_in_check = bool(L.lmrl_chess_host_status(buf) & 1)
        if not black_in_check and not white_in_check:
            return fen_w


def text_env_eval_chess_positions(positions: List[str], policy: Union[TextPolicy, BatchedTextPolicy], n_rollouts: int,
                                  initial_text_history: Optional[TextHistory] = None, seed_generator: Optional[Iterator[int]] = None,
                                  env_options: Optional[Dict] = None, interaction_callback=None, bsize: int = 1, verbose: bool = True,
                                  random_opponent: bool = False, max_moves: int = 400, **engine_kw):
    """env.py:257-342: `n_rollouts` games from each start position; the summary adds victories / percent_illegals / episode_length."""
    interactions, rs, dones, victories, percent_illegals, episode_length = [], [], [], [], [], []
    for position in positions:
        env = FenChessHistoryEnv(from_position=position, random_opponent=random_opponent, max_moves=max_moves, **engine_kw)
        env_interactions = []
        for _ in range((n_rollouts + (bsize - 1)) // bsize):
            actual = min(n_rollouts - len(env_interactions), bsize)
            batch = interact_environment(env, policy, initial_text_history=initial_text_history,
                                         env_seed=[None] * actual if seed_generator is None else [next(seed_generator) for _ in range(actual)],
                                         env_options=[env_options] * actual, bsize=actual, npad=bsize - actual)
            for interaction in batch:
                env_interactions.append(interaction)
                rewards = [x.reward for x in interaction]
                victories.append(1 if 1 in rewards else 0)
                num_illegal = sum(1 if x.reward == -1 and i < len(rewards) - 1 else 0 for i, x in enumerate(interaction))
                percent_illegals.append(num_illegal / len(rewards) * 100)
                episode_length.append(len(rewards))
                rs.append(sum(rewards))
                dones.append(interaction[-1].done)
                if interaction_callback is not None:
                    interaction_callback(interaction)
        env.close()
        interactions.extend(env_interactions)
    summ = lambda x: dict(mean=np.mean(x), std=np.std(x), min=np.min(x), max=np.max(x))
    return interactions, dict(reward=summ(np.asarray(rs, dtype=np.float32)), done=summ(np.asarray(dones, dtype=np.float32)), victories=summ(victories),
                              percent_illegals=summ(percent_illegals), episode_length=summ(episode_length))
