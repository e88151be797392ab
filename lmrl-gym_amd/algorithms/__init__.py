"""LLM_RL.algorithms counterparts (PPO, ILQL, MC returns, BC) on the HIP kernels."""
