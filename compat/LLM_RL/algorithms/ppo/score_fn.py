"""`LLM_RL.algorithms.ppo.score_fn` (reference: ppo/score_fn.py:10-126)."""
from lmrl_gym_amd.algorithms.reranker import build_bc_score_fn, build_ppo_score_fn  # noqa: F401
