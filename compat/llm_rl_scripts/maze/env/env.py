"""`llm_rl_scripts.maze.env.env` (reference: maze/env/env.py:8-214)."""
from lmrl_gym_amd.envs.maze import (MazeEnv, describe_observation, describe_observation_give_position,  # noqa: F401
                                    describe_observation_only_walls, illegal_penalty_diff_scale, illegal_penalty_reward,
                                    manhatten_actions, maze_proposal_function, standard_reward, update_position)
