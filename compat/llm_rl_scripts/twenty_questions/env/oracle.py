"""`llm_rl_scripts.twenty_questions.env.oracle` (reference: twenty_questions/env/oracle.py:14-87); `get_t5_oracle_prompt` keeps its name."""
from lmrl_gym_amd.envs.twenty_questions import GPT2EngineOracle, ModelOracle, TwentyQuestionsOracle  # noqa: F401
from lmrl_gym_amd.envs.twenty_questions import get_oracle_prompt as get_t5_oracle_prompt  # noqa: F401
