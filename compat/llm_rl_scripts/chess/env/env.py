"""`llm_rl_scripts.chess.env.env` (reference: chess/env/env.py:13-342)."""
from lmrl_gym_amd.envs.chess import (ChessEnv, FenChessHistoryEnv, FenChessHistoryEnvSingleTurn, large_piece_random_endgame, postprocess_move,  # noqa: F401
                                     postprocess_state, preprocess_move, preprocess_state, preprocess_state_og, text_env_eval_chess_positions)
