"""Maze on the MI355X: host text rendering around `lmrl_maze_*` (csrc/maze.hip).

Public names mirror the reference (`llm_rl_scripts/maze/env/{env,maze_utils,mazes}.py`):
`MazeEnv`, `setup_maze_env`, `maze_solver`, `manhatten_actions`, `maze_proposal_function`, the three
observation describers and the three reward functions; plus `VectorMazeEnv`, the lock-step
`BatchedTextEnv` stepping N envs per launch.
"""
from __future__ import annotations

import random
from collections import deque
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .. import _lib
from ..environment import BatchedTextEnv, Text, TextEnv, TextHistory

KIND_OBS, KIND_FAILURE, KIND_SUCCESS, KIND_OBS_ONLY = 0, 1, 2, 3
ACTION_OTHER = 4

# action string -> (d_row, d_col); dict order is the proposal order (env.py:94-102)
manhatten_actions: Dict[str, Tuple[int, int]] = {
    "move left\n": (0, -1), "move right\n": (0, 1), "move up\n": (-1, 0), "move down\n": (1, 0),
}
_ACTION_CODE = {a: i for i, a in enumerate(manhatten_actions)}   # matches LMRL_MAZE_LEFT..DOWN
_WALL_NAMES = ("to your right", "to your left", "above you", "below you")   # bit order of walls_d


def double_t_maze() -> np.ndarray:
    g = np.ones((10, 13), dtype=np.uint8)
    g[1, 1:6] = 0; g[1, 7:12] = 0          # the two top corridors
    g[2:6, 3] = 0; g[2:6, 9] = 0           # the two stems
    g[5, 3:10] = 0                         # the lower bar
    g[6:9, 6] = 0                          # the tail down to the goal
    return g


def maze2d_umaze() -> np.ndarray:
    g = np.ones((5, 5), dtype=np.uint8)
    g[1, 1:4] = 0; g[2:4, 1] = 0; g[2:4, 3] = 0
    return g


def maze_proposal_function(text_history: TextHistory) -> List[TextHistory]:
    return [tuple(text_history) + (Text(a, True),) for a in manhatten_actions]


# ----------------------------------------------------------------------------- observation text (env.py:8-81)
def describe_objects(obj: str, relations: List[str]) -> str:
    if not relations:
        return f"There are no {obj}s near you."
    if len(relations) == 1:
        return f"There is a {obj} {relations[0]}."
    return f"There are {obj}s {', '.join(relations)}."


def _spaced(x) -> str:
    return " ".join(str(x))


def _wall_sentence(maze, position) -> str:
    r, c = position
    rel = []
    for name, (dy, dx) in zip(_WALL_NAMES, ((0, 1), (0, -1), (-1, 0), (1, 0))):
        if maze[r + dy, c + dx] == 1:
            rel.append(name)
    return describe_objects("wall", rel)


def describe_observation(maze, position, goal_position, initial_position=None, move_history=None) -> str:
    return (f"The goal is at position {_spaced(goal_position[0])}, {_spaced(goal_position[1])}. "
            f"{_wall_sentence(maze, position)}\n")


def describe_observation_give_position(maze, position, goal_position, initial_position=None, move_history=None) -> str:
    return (f"The goal is at position {_spaced(goal_position[0])}, {_spaced(goal_position[1])}. "
            f"Your current position is at position {_spaced(position[0])}, {_spaced(position[1])}. "
            f"{_wall_sentence(maze, position)}\n")


def describe_observation_only_walls(maze, position, goal_position=None, initial_position=None, move_history=None) -> str:
    return f"{_wall_sentence(maze, position)}\n"


_DESCRIBERS = {f.__name__: f for f in (describe_observation, describe_observation_give_position, describe_observation_only_walls)}


# ----------------------------------------------------------------------------- rewards (env.py:109-131)
def _reward_fn(name: str, at_goal: float, illegal: float, otherwise: float):
    def f(action, goal, position, possible_actions):
        if position[0] == goal[0] and position[1] == goal[1]:
            return at_goal
        return illegal if action not in possible_actions else otherwise
    f.__name__ = name
    f.table = (at_goal, illegal, otherwise)
    return f


standard_reward = _reward_fn("standard_reward", 0.0, -4.0, -1.0)
illegal_penalty_reward = _reward_fn("illegal_penalty_reward", 1.0, -1.0, 0.0)
illegal_penalty_diff_scale = _reward_fn("illegal_penalty_diff_scale", 1.0, -100.0, -1.0)
_REWARDS = {f.__name__: f for f in (standard_reward, illegal_penalty_reward, illegal_penalty_diff_scale)}


def _reward_table(fn) -> Tuple[float, float, float]:
    if hasattr(fn, "table"):
        return fn.table
    # any callable with the reference signature that depends only on (at goal?, legal action?)
    acts = manhatten_actions
    legal = next(iter(acts))
    return (float(fn(legal, (1, 1), (1, 1), acts)), float(fn("<illegal>", (1, 1), (0, 0), acts)),
            float(fn(legal, (1, 1), (0, 0), acts)))


def maze_solver(maze: np.ndarray, goal_positions: Sequence[Tuple[int, int]]) -> Dict[Tuple[int, int], str]:
    """BFS from the goals over cells == 1 of `maze` (callers pass 1 - walls) -> optimal move per cell
    (maze_utils.py:91-116)."""
    grid = np.asarray(maze).tolist()
    assert len(grid) > 0 and len(grid[0]) > 0, "maze must be non-zero in area"
    assert all(grid[g[0]][g[1]] == 1 for g in goal_positions), "goal pos must be 1"
    back = {(1, 0): "move up\n", (0, 1): "move left\n", (-1, 0): "move down\n", (0, -1): "move right\n"}
    R, C = len(grid), len(grid[0])
    frontier = deque(tuple(g) for g in goal_positions)
    seen = set(frontier)
    policy: Dict[Tuple[int, int], str] = {}
    while frontier:
        r, c = frontier.popleft()
        for dr, dc in ((1, 0), (0, 1), (-1, 0), (0, -1)):
            nr, nc = r + dr, c + dc
            if (nr, nc) in seen or not (0 <= nr < R and 0 <= nc < C) or grid[nr][nc] == 0:
                continue
            seen.add((nr, nc))
            frontier.append((nr, nc))
            policy[(nr, nc)] = back[(dr, dc)]
    return policy


def _seed_magnitude(seed: Optional[int]) -> int:
    if seed is None:
        return random.SystemRandom().getrandbits(63)
    m = abs(int(seed))
    if m >= 1 << 64:
        raise ValueError("lmrl_gym_amd: env seeds must fit in 64 bits")
    return m


# ----------------------------------------------------------------------------- device-backed batched env
class VectorMazeEnv(BatchedTextEnv):
    def __init__(self, maze: np.ndarray, valid_goals: np.ndarray, actions: Dict[str, Tuple[int, int]] = manhatten_actions,
                 max_steps: Optional[int] = None, display_initial_position: bool = False,
                 describe_function: Callable = describe_observation_give_position,
                 reward_function: Callable = standard_reward, last_k: int = 40):
        import torch
        assert len(maze.shape) == 2
        assert all(maze[g[0], g[1]] == 0 for g in valid_goals)
        assert dict(actions) == manhatten_actions, "the device kernel implements the 4 manhattan moves"
        self.maze = np.ascontiguousarray(maze, dtype=np.uint8)
        self.valid_goals = np.ascontiguousarray(valid_goals, dtype=np.int32).reshape(-1, 2)
        self.actions, self.max_steps, self.last_k = actions, max_steps, last_k
        self.display_initial_position = display_initial_position
        self.describe_function, self.reward_function = describe_function, reward_function
        self._torch = torch
        self.device = _lib.require_gpu()
        self._L = _lib.lib()
        rew = np.asarray(_reward_table(reward_function), dtype=np.float32)
        self._ctx = self._L.lmrl_maze_create(self.maze.ctypes.data, self.maze.shape[0], self.maze.shape[1],
                                             self.valid_goals.ctypes.data, len(self.valid_goals),
                                             -1 if max_steps is None else int(max_steps), rew.ctypes.data)
        if not self._ctx:
            raise _lib.LmrlError(self._L.lmrl_last_error().decode())
        self.n = 0

    def _alloc(self, n: int):
        t = self._torch
        self.n = n
        self.state = t.zeros((5, n), dtype=t.int32, device=self.device)
        self.mt = t.zeros(self._L.lmrl_mt_bytes(n), dtype=t.uint8, device=self.device)
        self.reward = t.zeros(n, dtype=t.float32, device=self.device)
        self.done = t.zeros(n, dtype=t.uint8, device=self.device)
        self.kind = t.zeros(n, dtype=t.uint8, device=self.device)
        self.walls = t.zeros(n, dtype=t.uint8, device=self.device)
        self.initial_positions = [None] * n
        self.move_history: List[List[str]] = [[] for _ in range(n)]

    def positions(self) -> np.ndarray:
        """[N][5] host copy of (pos_r, pos_c, goal_r, goal_c, num_steps)."""
        return self.state.cpu().numpy().T.copy()

    def _describe(self, st_row, i: int) -> str:
        pos, goal = [int(st_row[0]), int(st_row[1])], [int(st_row[2]), int(st_row[3])]
        if self.describe_function in _DESCRIBERS.values():      # the module's describers depend on (position, goal) only: render each pair once
            memo = self.__dict__.setdefault("_describe_memo", {})
            key = (pos[0], pos[1], goal[0], goal[1])
            text = memo.get(key)
            if text is None:
                text = memo[key] = self.describe_function(self.maze, pos, goal, None, [])
            return text
        return self.describe_function(self.maze, pos, goal, self.initial_positions[i], self.move_history[i])

    def reset_device(self, seed, options=None) -> None:
        """MazeEnv.reset for len(seed) envs without leaving the device: state only, no observation text (device rollout loops)."""
        t = self._torch
        if options is None:
            options = [None] * len(seed)
        assert len(seed) == len(options)
        n = len(seed)
        if n != self.n:
            self._alloc(n)
        goal = np.full((n, 2), -1, dtype=np.int32)
        init = np.full((n, 2), -1, dtype=np.int32)
        free = None
        for i, o in enumerate(options):
            if o is not None and "goal" in o:
                goal[i] = o["goal"]
            if o is not None and "init_position" in o:
                free = np.argwhere(self.maze == 0).tolist() if free is None else free
                g = list(o["goal"]) if "goal" in o else None
                assert list(o["init_position"]) in free and list(o["init_position"]) != g
                init[i] = o["init_position"]
        seeds = np.array([_seed_magnitude(s) for s in seed], dtype=np.uint64).view(np.int64)
        seeds_d, goal_d, init_d = (t.from_numpy(x.copy()).to(self.device) for x in (seeds, goal, init))
        _lib.check(self._L.lmrl_maze_reset(self._ctx, _lib.ptr(self.state), _lib.ptr(self.mt), _lib.ptr(seeds_d),
                                           _lib.ptr(goal_d), _lib.ptr(init_d), None, n, _lib.stream_ptr()), "lmrl_maze_reset")

    def reset(self, seed=None, options=None) -> List[TextHistory]:
        if seed is None and options is None:
            seed, options = [None], [None]
        elif seed is None:
            seed = [None] * len(options)
        elif options is None:
            options = [None] * len(seed)
        self.reset_device(seed, options)
        n = self.n
        st = self.positions()
        out = []
        for i in range(n):
            self.move_history[i] = []
            self.initial_positions[i] = [int(st[i, 0]), int(st[i, 1])] if self.display_initial_position else None
            out.append((Text(self._describe(st[i], i), False),))
        return out

    def step(self, text_history, done=None):
        t = self._torch
        assert self.n > 0, "reset must be called before step"
        assert len(text_history) == self.n
        if done is None:
            done = [False] * self.n
        action = np.full(self.n, ACTION_OTHER, dtype=np.uint8)
        active = np.zeros(self.n, dtype=np.uint8)
        for i, (h, d) in enumerate(zip(text_history, done)):
            if d or h is None:
                continue
            assert h[-1].is_action
            action[i] = _ACTION_CODE.get(h[-1].text, ACTION_OTHER)
            active[i] = 1
        action_d = t.from_numpy(action).to(self.device)
        active_d = t.from_numpy(active).to(self.device)
        _lib.check(self._L.lmrl_maze_step(self._ctx, _lib.ptr(self.state), _lib.ptr(action_d), _lib.ptr(active_d),
                                          _lib.ptr(self.reward), _lib.ptr(self.done), _lib.ptr(self.kind),
                                          _lib.ptr(self.walls), self.n, _lib.stream_ptr()), "lmrl_maze_step")
        rew, dn, kind = self.reward.cpu().numpy(), self.done.cpu().numpy(), self.kind.cpu().numpy()
        st = self.positions()
        out = []
        for i in range(self.n):
            if not active[i]:
                out.append(None)
                continue
            h = tuple(text_history[i])
            if kind[i] == KIND_FAILURE:
                out.append(((Text("Failure\n", False),), -1.0, True))
                continue
            self.move_history[i].append(h[-1].text.replace("\n", ""))
            if kind[i] == KIND_SUCCESS:
                out.append(((Text("Success\n", False),), float(rew[i]), True))
                continue
            obs = Text(self._describe(st[i], i), False)
            if kind[i] == KIND_OBS_ONLY:
                out.append(((obs,), float(rew[i]), False))
            else:
                new = list(h) + [obs]
                out.append((tuple(new[max(0, len(new) - self.last_k):]), float(rew[i]), bool(dn[i])))
        return out

    def close(self) -> None:
        if getattr(self, "_ctx", None):
            self._L.lmrl_maze_destroy(self._ctx)
            self._ctx = None

    def copy(self):
        return VectorMazeEnv(self.maze, self.valid_goals, self.actions, self.max_steps, self.display_initial_position,
                             self.describe_function, self.reward_function, self.last_k)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MazeEnv(TextEnv):
    """Single env with the reference constructor signature (env.py:133-159), one device slot."""

    def __init__(self, maze: np.ndarray, valid_goals: np.ndarray, actions: Dict[str, Tuple[int, int]],
                 max_steps: Optional[int] = None, display_initial_position: bool = False,
                 describe_function: Callable = describe_observation_give_position,
                 reward_function: Callable = standard_reward, last_k: int = 40):
        self._args = (maze, valid_goals, actions, max_steps, display_initial_position, describe_function,
                      reward_function, last_k)
        self.maze, self.valid_goals, self.actions = maze, valid_goals, actions
        self.max_steps, self.last_k = max_steps, last_k
        self._vec: Optional[VectorMazeEnv] = None

    def as_batched(self) -> VectorMazeEnv:
        return VectorMazeEnv(*self._args)

    def _v(self) -> VectorMazeEnv:
        if self._vec is None:
            self._vec = self.as_batched()
        return self._vec

    @property
    def position(self):
        return self._v().positions()[0, :2].tolist()

    @property
    def goal(self):
        return self._v().positions()[0, 2:4].tolist()

    @property
    def num_steps(self):
        return int(self._v().positions()[0, 4])

    def reset(self, seed: Optional[int] = None, options: Optional[Dict] = None) -> TextHistory:
        return self._v().reset([seed], [options])[0]

    def step(self, text_history: TextHistory):
        return self._v().step([text_history])[0]

    def close(self) -> None:
        if self._vec is not None:
            self._vec.close()

    def copy(self):
        return MazeEnv(*self._args)


def setup_maze_env(maze_name, describe_function, reward_function=None, last_k=1, max_steps=100) -> MazeEnv:
    """maze_utils.py:9-52."""
    if maze_name == "umaze":
        maze, valid_goals = maze2d_umaze(), np.array([[3, 3]])
    elif maze_name == "double_t_maze":
        maze, valid_goals = double_t_maze(), np.array([[8, 6]])
    else:
        raise ValueError(f"unknown maze name: {maze_name}")
    if isinstance(describe_function, str):
        if describe_function not in _DESCRIBERS:
            raise ValueError(f"unknown describe function: {describe_function}")
        describe_function = _DESCRIBERS[describe_function]
    if reward_function is None:
        reward_function = standard_reward
    elif isinstance(reward_function, str):
        if reward_function not in _REWARDS:
            raise ValueError(f"unknown reward function: {reward_function}")
        reward_function = _REWARDS[reward_function]
    return MazeEnv(maze=maze, valid_goals=valid_goals, actions=manhatten_actions, max_steps=max_steps,
                   display_initial_position=True, describe_function=describe_function,
                   reward_function=reward_function, last_k=last_k)


def update_position(maze: np.ndarray, position: Tuple[int, int], action: str, actions: Dict[str, Tuple[int, int]] = manhatten_actions
                    ) -> Tuple[int, int]:
    """maze/env/env.py:104-107 (host face of the move the device kernel applies)."""
    if action in actions and maze[position[0] + actions[action][0], position[1] + actions[action][1]] == 0:
        return (position[0] + actions[action][0], position[1] + actions[action][1])
    return tuple(position)


def double_t_maze_optimal_directions() -> Dict[Tuple[int, int], str]:
    """The cell -> optimal move table of maze/env/mazes.py:20-48, derived with `maze_solver` (tests pin it to the
    reference's table through tests/golden/maze_traces.json)."""
    sol = maze_solver(1 - double_t_maze(), [(8, 6)])
    return {k: v for k, v in sol.items() if k != (8, 6)}


def compute_move_accuracy(policy, reranker: bool = False, verbose: bool = False) -> float:
    """maze/env/maze_utils.py:63-89: percentage of free cells of the double-T maze (goal (8, 6)) where the policy's action
    for `describe_observation_give_position` is the optimal move.  Batched policies get all cells in ONE `act` call
    instead of the reference's per-cell loop (same observations, same answers)."""
    maze, goal = double_t_maze(), (8, 6)
    answers = double_t_maze_optimal_directions()
    positions = [tuple(p) for p in np.argwhere(maze == 0).tolist()]
    hists = [(Text(describe_observation_give_position(maze, pos, goal), False),) for pos in positions]
    if reranker:
        preds = [policy.act(h)[-1].text for h in hists]
    else:
        outs = policy.act(list(hists), done=[False] * len(hists))
        preds = [o[-1].text for o in outs]
    n_ok = 0
    for pos, pred in zip(positions, preds):
        if pos == goal:
            continue
        ok = pred == answers[pos]
        n_ok += int(ok)
        if verbose:
            print("correct!" if ok else "incorrect!", pos, repr(pred), repr(answers[pos]))
    return n_ok / (len(positions) - 1) * 100


def pick_start_position(maze_name):
    if maze_name == "umaze":
        return (3, 1)
    if maze_name == "double_t_maze":
        return (1, 1)
    raise ValueError(f"unknown maze name: {maze_name}")
