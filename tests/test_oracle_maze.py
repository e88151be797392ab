"""Pins oracle/maze.py against traces produced by the reference MazeEnv (tests/golden/maze_traces.json)."""
from conftest import load_golden
from oracle.maze import MAZES, OracleMazeEnv, maze_solver


def test_maze_grids_match_reference_data():
    g = load_golden("maze_traces.json")
    assert MAZES["double_t_maze"]["grid"] == g["double_t_maze"]
    assert MAZES["umaze"]["grid"] == g["umaze"]


def test_known_answer_table():
    # the only known-answer data the reference holds for this path (mazes.py:20-48)
    g = load_golden("maze_traces.json")
    sol = maze_solver(MAZES["double_t_maze"]["grid"], (8, 6))
    for pos, mv in g["double_t_maze_optimal_directions"]:
        assert sol[tuple(pos)] == mv


def test_maze_traces():
    g = load_golden("maze_traces.json")
    n = 0
    for ep in g["episodes"]:
        env = OracleMazeEnv(ep["maze"], ep["describe"], ep["reward_fn"], last_k=ep["last_k"], max_steps=ep["max_steps"])
        hist = env.reset(seed=ep["seed"], options=ep["options"])
        assert hist == ((ep["reset_obs"], False),)
        assert env.position == ep["init_position"] and env.goal == ep["goal"]
        for st in ep["steps"]:
            hist, r, done = env.step(hist + ((st["action"], True),))
            assert [[t, a] for t, a in hist] == st["history"]
            assert r == st["reward"] and done == st["done"]
            assert env.position == st["position"] and env.num_steps == st["num_steps"]
            n += 1
    assert n > 1000
