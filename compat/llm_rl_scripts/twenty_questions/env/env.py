"""`llm_rl_scripts.twenty_questions.env.env` (reference: twenty_questions/env/env.py:9-141)."""
from lmrl_gym_amd.envs.twenty_questions import BatchedTwentyQuestionsPolicyEnvironment, TwentyQuestionsPolicyEnvironment  # noqa: F401
