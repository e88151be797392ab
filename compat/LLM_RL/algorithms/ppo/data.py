"""`LLM_RL.algorithms.ppo.data` (reference: ppo/data.py:9-114)."""
from lmrl_gym_amd.algorithms.ppo import PPOData, PPODataset  # noqa: F401
