"""`LLM_RL.algorithms.ilql.base_interface` (reference: ilql/base_interface.py:22-119)."""
from lmrl_gym_amd.algorithms.ilql import ilql_loss  # noqa: F401
