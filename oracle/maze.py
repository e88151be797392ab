"""Oracle Maze env (TEST INFRASTRUCTURE): small pure-Python restatement of the reference MazeEnv.

Follows llm_rl_scripts/maze/env/env.py:8-214 and maze_utils.py:9-52 (setup), on plain
(text, is_action) tuples.  Reset draws use `random.Random(seed)` — the same CPython
MT19937 the reference reaches through `random.seed(seed)` under its RandomState
save/restore shim (randomness.py:9-19), so the global generator is never touched here.
Pinned by tests/test_oracle_maze.py against tests/golden/maze_traces.json.
"""
from __future__ import annotations

import random
from typing import Dict, List, Optional, Tuple

import numpy as np

TextItem = Tuple[str, bool]

MAZES = {
    # mazes.py:6-18 / 50-58 (data: 1 = wall)
    "double_t_maze": dict(
        grid=[[1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1],
              [1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1],
              [1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 1, 1, 1],
              [1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 1, 1, 1],
              [1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 1, 1, 1],
              [1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1],
              [1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1],
              [1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1],
              [1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1],
              [1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1]],
        valid_goals=[[8, 6]]),
    "umaze": dict(
        grid=[[1, 1, 1, 1, 1], [1, 0, 0, 0, 1], [1, 0, 1, 0, 1], [1, 0, 1, 0, 1], [1, 1, 1, 1, 1]],
        valid_goals=[[3, 3]]),
}

# env.py:94-99 (dict order matters for nothing but proposals)
ACTIONS: Dict[str, Tuple[int, int]] = {
    "move left\n": (0, -1), "move right\n": (0, 1), "move up\n": (-1, 0), "move down\n": (1, 0),
}
# env.py:27 / 59 / 74: description order right, left, above, below
DELTAS = [("to your right", (0, 1)), ("to your left", (0, -1)), ("above you", (-1, 0)), ("below you", (1, 0))]


def _digits(x: int) -> str:
    return " ".join(str(x))  # env.py:24, 57-58


def describe_objects(obj: str, relations: List[str]) -> str:  # env.py:8-13
    if len(relations) == 0:
        return f"There are no {obj}s near you."
    if len(relations) == 1:
        return f"There is a {obj} {relations[0]}."
    return f"There are {obj}s {', '.join(relations)}."


def _walls(maze, pos):
    return [k for k, (dy, dx) in DELTAS if maze[pos[0] + dy][pos[1] + dx] == 1]


def describe(kind: str, maze, pos, goal) -> str:
    wall = describe_objects("wall", _walls(maze, pos))
    goal_d = f"The goal is at position {_digits(goal[0])}, {_digits(goal[1])}."
    if kind == "describe_observation":  # env.py:15-49
        return f"{goal_d} {wall}\n"
    if kind == "describe_observation_give_position":  # env.py:51-68
        cur = f"Your current position is at position {_digits(pos[0])}, {_digits(pos[1])}."
        return f"{goal_d} {cur} {wall}\n"
    if kind == "describe_observation_only_walls":  # env.py:70-81
        return f"{wall}\n"
    raise ValueError(kind)


def reward_value(kind: str, action: str, goal, pos) -> float:  # env.py:109-131
    at_goal = pos[0] == goal[0] and pos[1] == goal[1]
    legal = action in ACTIONS
    table = {"standard_reward": (0.0, -4.0, -1.0), "illegal_penalty_reward": (1.0, -1.0, 0.0),
             "illegal_penalty_diff_scale": (1.0, -100.0, -1.0)}[kind]
    return table[0] if at_goal else (table[1] if not legal else table[2])


class OracleMazeEnv:
    def __init__(self, maze_name: str, describe_function: str, reward_function: str = "standard_reward",
                 last_k: int = 1, max_steps: Optional[int] = 100):
        m = MAZES[maze_name]
        self.maze = [list(r) for r in m["grid"]]
        self.valid_goals = [list(g) for g in m["valid_goals"]]
        self.describe_function, self.reward_function = describe_function, reward_function
        self.last_k, self.max_steps = last_k, max_steps
        self.num_steps = 0
        self.position = self.goal = None

    def reset(self, seed: Optional[int] = None, options: Optional[Dict] = None):  # env.py:186-214
        rng = random.Random(seed)
        self.num_steps = 0
        if options is not None and "goal" in options:
            self.goal = list(options["goal"])
        else:
            self.goal = list(rng.choice(self.valid_goals))
        positions = [[int(r), int(c)] for r, c in np.argwhere(np.asarray(self.maze) == 0).tolist()]
        positions.remove(self.goal)
        if options is not None and "init_position" in options:
            assert list(options["init_position"]) in positions
            self.position = list(options["init_position"])
        else:
            self.position = list(rng.choice(positions))
        return ((describe(self.describe_function, self.maze, self.position, self.goal), False),)

    def step(self, text_history: Tuple[TextItem, ...]):  # env.py:161-184
        assert text_history[-1][1]
        if self.max_steps is not None and self.num_steps >= self.max_steps:
            return (("Failure\n", False),), -1.0, True
        action = text_history[-1][0]
        if action in ACTIONS:  # update_position, env.py:104-107
            dy, dx = ACTIONS[action]
            if self.maze[self.position[0] + dy][self.position[1] + dx] == 0:
                self.position = [self.position[0] + dy, self.position[1] + dx]
        reward = reward_value(self.reward_function, action, self.goal, self.position)
        if self.position == self.goal:
            return (("Success\n", False),), reward, True
        self.num_steps += 1
        obs = describe(self.describe_function, self.maze, self.position, self.goal)
        if action not in ACTIONS:
            return ((obs, False),), reward, False
        new_history = list(text_history) + [(obs, False)]
        new_history = new_history[max(0, len(new_history) - self.last_k):]
        return tuple(new_history), reward, False


def maze_solver(maze, goal) -> Dict[Tuple[int, int], str]:
    """BFS from the goal over free cells (maze_utils.py:91-116); returns cell -> optimal move string."""
    from collections import deque
    move = {(0, 1): "move right\n", (0, -1): "move left\n", (1, 0): "move down\n", (-1, 0): "move up\n"}
    R, C = len(maze), len(maze[0])
    q, seen, pol = deque([tuple(goal)]), {tuple(goal)}, {}
    while q:
        x, y = q.popleft()
        for dx, dy in [(1, 0), (0, 1), (-1, 0), (0, -1)]:
            n = (x + dx, y + dy)
            if n in seen or n[0] < 0 or n[0] >= R or n[1] < 0 or n[1] >= C or maze[n[0]][n[1]] == 1:
                continue
            q.append(n); seen.add(n); pol[n] = move[(-dx, -dy)]
    return pol
