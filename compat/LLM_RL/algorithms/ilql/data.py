"""`LLM_RL.algorithms.ilql.data` (reference: ilql/data.py:10-132)."""
from lmrl_gym_amd.algorithms.ilql import ILQLData, ILQLDataset  # noqa: F401
