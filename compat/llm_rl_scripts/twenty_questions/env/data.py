"""`llm_rl_scripts.twenty_questions.env.data` (reference: twenty_questions/env/data.py:20-116,292-391)."""
from lmrl_gym_amd.envs.twenty_questions import (DEFAULT_OBJECT_DICT, INITIAL_STR, INVALID_QUESTION, WordVariants, asker_postproc,  # noqa: F401
                                               asker_postproc_filter_repeats, asker_postproc_simple, create_trajectory_from_history,
                                               get_default_word_list, is_done)
