"""`LLM_RL.algorithms.ilql.gpt2.interface` (reference: ilql/gpt2/interface.py)."""
from lmrl_gym_amd.algorithms.ilql import GPT2ILQLInference, GPT2ILQLTrain  # noqa: F401
