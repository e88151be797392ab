"""`LLM_RL.algorithms.mc_returns.base_interface` (reference: mc_returns/base_interface.py:19-60)."""
from lmrl_gym_amd.algorithms.mc_returns import mc_loss  # noqa: F401
