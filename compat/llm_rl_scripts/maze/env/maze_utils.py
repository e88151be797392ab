"""`llm_rl_scripts.maze.env.maze_utils` (reference: maze/env/maze_utils.py:9-116)."""
from lmrl_gym_amd.envs.maze import compute_move_accuracy, maze_solver, pick_start_position, setup_maze_env  # noqa: F401
