"""ctypes binding of liblmrl_amd.so — the only way the Python host reaches the HIP kernels.

There is NO CPU fallback: if the library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "liblmrl_amd.so")

_lib: Optional[ctypes.CDLL] = None

c_void_p, c_int, c_float, c_size_t, c_char_p = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_char_p

# name -> (restype, argtypes); mirrors include/lmrl_amd.h one to one
_SIGS = {
    "lmrl_last_error": (c_char_p, []),
    "lmrl_version": (c_int, []),
    "lmrl_device_arch": (c_char_p, []),
    "lmrl_mt_bytes": (c_size_t, [c_int]),
    "lmrl_mt_seed": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "lmrl_mt_stream": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "lmrl_mt_randbelow": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "lmrl_wordle_create": (c_void_p, [c_char_p, c_int, c_int, c_float]),
    "lmrl_wordle_destroy": (None, [c_void_p]),
    "lmrl_wordle_state_bytes": (c_size_t, [c_int]),
    "lmrl_wordle_reset": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "lmrl_wordle_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "lmrl_wordle_set_variant": (c_int, [c_void_p, c_int]),
    "lmrl_wordle_export_state": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "lmrl_maze_create": (c_void_p, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "lmrl_maze_destroy": (None, [c_void_p]),
    "lmrl_maze_state_bytes": (c_size_t, [c_int]),
    "lmrl_maze_reset": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "lmrl_maze_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "lmrl_gae": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_void_p]),
    "lmrl_rtg": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "lmrl_whiten_moments": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "lmrl_gae_moments_partials": (c_int, [c_int, c_int]),
    "lmrl_gae_moments": (c_int, [c_void_p] * 6 + [c_int, c_int, c_float, c_float, c_void_p, c_void_p]),
    "lmrl_whiten_finish": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "lmrl_whiten_apply_partials": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_int, c_void_p]),
    "lmrl_whiten_apply": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "lmrl_ppo_count": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lmrl_ppo_block": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lmrl_ppo_shape": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lmrl_ppo_unroll": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "lmrl_ppo_truncate_turns": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, ctypes.c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lmrl_compact_flags": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "lmrl_len_mask_pos": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "lmrl_add_i32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "lmrl_seq_mask_pos": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "lmrl_masked_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lmrl_gather_rows_bytes": (c_int, [c_void_p, c_void_p, c_void_p, c_int, ctypes.c_long, c_void_p]),
    "lmrl_gpt2_refresh": (c_int, [c_void_p, c_void_p]),
    "lmrl_exclusive_scan_i32": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "lmrl_maze_tok_ppo_records": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int] + [c_void_p] * 10),
    "lmrl_maze_tok_ppo_records_hist": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int] + [c_void_p] * 10),
    "lmrl_maze_tok_set_spaced": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int]),
    "lmrl_gpt2_create": (c_void_p, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lmrl_gpt2_destroy": (None, [c_void_p]),
    "lmrl_gpt2_kv_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "lmrl_gpt2_ws_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "lmrl_gpt2_forward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                  c_void_p, c_void_p, ctypes.c_uint, c_void_p]),
    "lmrl_gemm_bf16_ld": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "lmrl_gemm_bf16_resid": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "lmrl_gemm_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "lmrl_gemm_set_variant": (None, [c_int]),
    "lmrl_gpt2_kv_broadcast": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lmrl_sgemm_set_variant": (None, [c_int]),
    "lmrl_train_ops_set_variant": (None, [c_int]),
    "lmrl_rl_reduce_set_variant": (None, [c_int]),
    "lmrl_sampler_set_variant": (None, [c_int]),
    "lmrl_flash_set_variant": (None, [c_int]),
    "lmrl_sample_ws_bytes": (c_size_t, [c_int, c_int]),
    "lmrl_sample_fb_offset": (c_size_t, [c_int, c_int]),
    "lmrl_sample_logits_steer": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lmrl_chunk_begin_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "lmrl_attn_cached_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "lmrl_attn_cached_f32_split3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "lmrl_chunk_end_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "lmrl_threefry2x32": (None, [c_void_p, c_void_p, c_void_p]),
    "lmrl_jax_random_bits_host": (c_int, [c_void_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, c_void_p]),
    "lmrl_lm_head_sample": (c_int, [c_void_p] * 8 + [c_int, c_int, c_int, c_int] + [c_void_p] * 8),
    "lmrl_maze_tok_create": (c_void_p, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int]),
    "lmrl_maze_tok_destroy": (None, [c_void_p]),
    "lmrl_maze_tok_begin": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "lmrl_maze_tok_turn": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "lmrl_maze_tok_prompt": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "lmrl_maze_tok_action": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "lmrl_maze_tok_result": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "lmrl_maze_hist_begin": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "lmrl_maze_hist_observe": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "lmrl_maze_hist_chunk": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "lmrl_maze_hist_action": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "lmrl_maze_tok_set_actions": (c_int, [c_void_p, c_void_p, c_int]),
    "lmrl_gpt2_kv_gather": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "lmrl_gpt2_kv_attach": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lmrl_gpt2_forward_prefixed": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                           ctypes.c_uint, c_void_p]),
    "lmrl_chess_pos_bytes": (c_size_t, []),
    "lmrl_chess_reset": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "lmrl_chess_agent_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "lmrl_chess_opponent_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "lmrl_chess_max_moves": (c_int, []),
    "lmrl_chess_describe": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "lmrl_chess_host_from_fen": (c_int, [ctypes.c_char_p, c_void_p]),
    "lmrl_chess_host_fen": (c_int, [c_void_p, c_void_p]),
    "lmrl_chess_host_legal_moves": (c_int, [c_void_p, c_void_p, c_void_p]),
    "lmrl_chess_host_agent_step": (c_int, [c_void_p, ctypes.c_char_p, c_void_p, c_void_p]),
    "lmrl_chess_host_opponent_step": (c_int, [c_void_p, ctypes.c_char_p, c_void_p, c_void_p, c_void_p]),
    "lmrl_chess_host_status": (c_int, [c_void_p]),
    "lmrl_wordle_tok_create": (c_void_p, [c_void_p, c_void_p, c_int, c_int, c_int]),
    "lmrl_wordle_tok_destroy": (None, [c_void_p]),
    "lmrl_wordle_tok_begin": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "lmrl_wordle_tok_accept": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "lmrl_wordle_tok_guess": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "lmrl_wordle_tok_observe": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "lmrl_wordle_tok_steer": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "lmrl_prof_enable": (None, [ctypes.c_uint]),
    "lmrl_prof_reset": (None, []),
    "lmrl_prof_n_tags": (c_int, []),
    "lmrl_prof_tag_name": (c_char_p, [c_int]),
    "lmrl_prof_read": (c_int, [c_int, c_void_p, c_void_p, c_void_p]),
    "lmrl_sgemm": (c_int, [c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int, ctypes.c_long, ctypes.c_long, c_void_p, c_int,
                           ctypes.c_long, ctypes.c_long, c_float, c_void_p, c_int, ctypes.c_long, ctypes.c_long, c_int, c_int, c_void_p, c_void_p]),
    "lmrl_embed_fwd": (c_int, [c_void_p] * 5 + [c_int, c_int, c_int, c_void_p]),
    "lmrl_cast_bf16": (c_int, [c_void_p, ctypes.c_long, c_int, c_int, c_void_p, ctypes.c_long, c_int, c_int, c_void_p]),
    "lmrl_cast_bf16_segments": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "lmrl_split3_bf16": (c_int, [c_void_p, ctypes.c_long, c_int, c_int, c_void_p, ctypes.c_long, c_void_p]),
    "lmrl_cast_bf16_t_colsum_ws_bytes": (c_size_t, [c_int, c_int]),
    "lmrl_cast_bf16_t_colsum": (c_int, [c_void_p, ctypes.c_long, c_int, c_int, c_void_p, ctypes.c_long, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "lmrl_gemm_bf16_splitk_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "lmrl_gemm_bf16_splitk_kmajor": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "lmrl_colsum_bf16": (c_int, [c_void_p, ctypes.c_long, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "lmrl_gemm_bf16_splitk": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "lmrl_gemm_bf16_splitk_bias": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "lmrl_ce_bwd_bf16": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, ctypes.c_long, c_int, c_void_p]),
    "lmrl_transpose_bf16_colsum": (c_int, [c_void_p, ctypes.c_long, c_int, c_int, c_void_p, ctypes.c_long, c_int, c_void_p, c_int, c_void_p,
                                           c_void_p]),
    "lmrl_transpose_add_f32": (c_int, [c_void_p, ctypes.c_long, c_void_p, ctypes.c_long, c_int, c_int, c_float, c_void_p]),
    "lmrl_gather_dot_f32": (c_int, [c_void_p, ctypes.c_long, c_void_p, ctypes.c_long, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "lmrl_flash_attn_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "lmrl_flash_attn_lse_bytes": (c_size_t, [c_int, c_int, c_int]),
    "lmrl_flash_attn_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "lmrl_flash_attn_fwd_staged": (c_int, [c_void_p] * 6 + [ctypes.c_long, c_int, c_int, c_int, c_int, c_void_p]),
    "lmrl_flash_attn_bwd": (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "lmrl_gemm_bf16_ce_slots": (c_int, [c_int, c_int, c_int]),
    "lmrl_gemm_bf16_ce": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lmrl_lse_from_partials": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lmrl_ce_bwd_bf16_inplace": (c_int, [c_void_p, ctypes.c_long, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "lmrl_gemm_bf16_gelu_dual": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "lmrl_gemm_bf16_qkv_heads": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "lmrl_gemm_bf16_gelu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "lmrl_gemm_bf16_gelu_dual_prebf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "lmrl_gemm_bf16_gelu_bwd_prebf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "lmrl_flash_attn_stage_ptrs": (c_int, [c_void_p, c_int, c_int, c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(ctypes.c_long)]),
    "lmrl_flash_attn_finish_staging": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "lmrl_flash_attn_bwd_staged": (c_int, [c_void_p] * 6 + [ctypes.c_long, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "lmrl_flash_attn_bwd_staged_attb": (c_int, [c_void_p] * 3 + [ctypes.c_long] + [c_void_p] * 3 + [ctypes.c_long, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "lmrl_adamw_segments": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_long, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float, c_int, c_void_p]),
    "lmrl_adamw_segments_polyak": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_long, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float, c_int,
                                           c_void_p, c_float, c_float, c_void_p]),
    "lmrl_layernorm_add_fwd_split3": (c_int, [c_void_p] * 7 + [c_int, c_int, c_float, c_void_p]),
    "lmrl_gelu_split3": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "lmrl_layernorm_add_fwd": (c_int, [c_void_p] * 8 + [ctypes.c_long, c_int, c_int, c_float, c_void_p]),
    "lmrl_layernorm_fwd_staged": (c_int, [c_void_p] * 7 + [ctypes.c_long, c_int, c_int, c_float, c_void_p]),
    "lmrl_gelu_fwd_staged": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_long, c_int, c_int, c_void_p]),
    "lmrl_embed_bwd": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, c_int, c_void_p]),
    "lmrl_layernorm_fwd": (c_int, [c_void_p] * 6 + [c_int, c_int, c_float, c_void_p]),
    "lmrl_layernorm_bwd": (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_void_p]),
    "lmrl_layernorm_bwd_fused_supported": (c_int, [c_int]),
    "lmrl_layernorm_bwd_fused_ws_bytes": (c_size_t, [c_int, c_int]),
    "lmrl_layernorm_bwd_fused": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                         c_int, c_void_p, c_void_p, ctypes.c_long, c_void_p]),
    "lmrl_gelu_bwd_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, ctypes.c_long, c_int, c_void_p]),
    "lmrl_gather_rows_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "lmrl_scatter_rows_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "lmrl_colsum_ws_bytes": (c_size_t, [c_int]),
    "lmrl_colsum": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "lmrl_colsum_weighted": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "lmrl_gelu_fwd": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "lmrl_gelu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "lmrl_relu_fwd": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "lmrl_relu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "lmrl_axpby": (c_int, [c_float, c_void_p, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "lmrl_adamw": (c_int, [c_void_p] * 4 + [c_size_t, c_float, c_float, c_float, c_float, c_float, c_int, c_void_p]),
    "lmrl_softmax_causal_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "lmrl_softmax_bwd": (c_int, [c_void_p, c_void_p, ctypes.c_long, c_int, c_void_p]),
    "lmrl_lse_gather": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "lmrl_ce_bwd": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "lmrl_mask_sum": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "lmrl_ppo_loss_blocks": (c_int, [c_size_t]),
    "lmrl_ppo_loss_nstats": (c_int, []),
    "lmrl_ppo_loss": (c_int, [c_void_p] * 8 + [c_size_t, c_float, c_float, c_float] + [c_void_p] * 5),
    "lmrl_ilql_loss_nstats": (c_int, []),
    "lmrl_ilql_loss": (c_int, [c_void_p] * 11 + [c_int, c_int, c_float, c_float, c_float] + [c_void_p] * 7),
    "lmrl_mc_loss_blocks": (c_int, [c_size_t]),
    "lmrl_mc_loss_nstats": (c_int, []),
    "lmrl_mc_loss": (c_int, [c_void_p] * 5 + [c_size_t, c_float] + [c_void_p] * 5),
    "lmrl_gen_accept": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "lmrl_sample_logits": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
}


class LmrlError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Load liblmrl_amd.so (once). Raises if it has not been built — never falls back to CPU code."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise LmrlError(
                f"{SO_PATH} not found: build it with `python lmrl-gym_amd/build.py` "
                "(or `__graft_entry__.build()`); there is no CPU fallback for the HIP path.")
        # torch first: it brings its own libamdhip64 and the library must bind to that same HIP runtime — loading ours
        # first leaves two runtimes in the process and every device call of this library then fails.
        import torch  # noqa: F401
        L = ctypes.CDLL(SO_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise LmrlError(f"{what or 'lmrl call'} failed (rc={rc}): {lib().lmrl_last_error().decode()}")


def ptr(t) -> Optional[int]:
    """Device/host address of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "lmrl: tensors handed to the C ABI must be contiguous"
    return t.data_ptr()


def stream_ptr(device=None) -> Optional[int]:
    import torch
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu() -> "torch.device":
    import torch
    if not torch.cuda.is_available():
        raise LmrlError("no MI355X visible: the lmrl_gym_amd HIP path has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())
