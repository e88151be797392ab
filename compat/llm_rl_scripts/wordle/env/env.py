"""`llm_rl_scripts.wordle.env.env` (reference: wordle/env/env.py:7-55) on the batched HIP env."""
from lmrl_gym_amd.envs.wordle import (ReformatWordleEnvironment, WordleEnvironment, deformat_history, reformat_history)  # noqa: F401
