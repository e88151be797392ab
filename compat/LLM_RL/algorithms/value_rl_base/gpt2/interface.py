"""`LLM_RL.algorithms.value_rl_base.gpt2.interface` (reference: value_rl_base/gpt2/interface.py:239-330)."""
from lmrl_gym_amd.policies import GPT2ValuePolicy  # noqa: F401
