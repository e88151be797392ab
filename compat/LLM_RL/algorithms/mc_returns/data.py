"""`LLM_RL.algorithms.mc_returns.data` (reference: mc_returns/data.py:10-74)."""
from lmrl_gym_amd.algorithms.mc_returns import MCData, get_rtg  # noqa: F401
