"""`LLM_RL.algorithms.ilql.gpt2.score_fn` (reference: ilql/gpt2/score_fn.py:11-68)."""
from lmrl_gym_amd.algorithms.reranker import build_ilql_score_fn  # noqa: F401
