"""`LLM_RL.environment` (reference: LLM_RL/environment.py) served by lmrl_gym_amd.environment."""
from lmrl_gym_amd.environment import *  # noqa: F401,F403
from lmrl_gym_amd import environment as _m
__all__ = [n for n in dir(_m) if not n.startswith("_")]
