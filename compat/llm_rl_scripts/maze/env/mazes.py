"""`llm_rl_scripts.maze.env.mazes` (reference: maze/env/mazes.py)."""
from lmrl_gym_amd.envs.maze import double_t_maze, double_t_maze_optimal_directions, maze2d_umaze  # noqa: F401
