"""`LLM_RL.algorithms.ppo.reranker_policy` (reference: ppo/reranker_policy.py:5-34)."""
from lmrl_gym_amd.algorithms.reranker import ReRankerPolicy, ReRankerSamplePolicy  # noqa: F401
