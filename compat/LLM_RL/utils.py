"""`LLM_RL.utils`: the numeric helpers on the hot path (reference: LLM_RL/utils.py:12-38)."""
from lmrl_gym_amd.algorithms.common import stats_from_sums  # noqa: F401
from lmrl_gym_amd.algorithms.ppo_inference import unpad_array  # noqa: F401
