/*
 * lmrl_amd.h — C ABI of liblmrl_amd.so, the MI355X (gfx950) engine behind LMRL-Gym's
 * rollout-and-train hot path.
 *
 * The reference has no FFI: its boundary is the Python protocol in LLM_RL/environment.py and
 * the duck-typed algorithm objects.  The entry points below are what a
 * ctypes binding on the reference side would call (INTEGRATION.md shows the stubs); each
 * one cites the reference code it replaces (paths relative to the reference repo root).
 *
 * Conventions
 *   - every `*_d` / "device" pointer is HBM owned by the CALLER (e.g. a torch tensor);
 *     contexts (`lmrl_*_ctx`) own only small read-only tables.
 *   - `stream` is a hipStream_t passed as void*; NULL = the null stream.  All calls are
 *     asynchronous with respect to the host unless stated otherwise.
 *   - return value: 0 = ok, non-zero = error (text via lmrl_last_error()).
 *   - batched state is struct-of-arrays over the env index so that lane i touches
 *     consecutive addresses (coalesced HBM access); layouts are documented next to the
 *     *_bytes() function that sizes them.
 */
#ifndef LMRL_AMD_H
#define LMRL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LMRL_OK 0
#define LMRL_ERR_ARG 1
#define LMRL_ERR_HIP 2

const char *lmrl_last_error(void);
int lmrl_version(void);
/* Name of the device the library is running on (hipDeviceProp.gcnArchName), "" if none. */
const char *lmrl_device_arch(void);

/* ------------------------------------------------------------------------------------------
 * MT19937 streams — CPython `random.Random(seed)` bit-for-bit.
 * Replaces: wordle/env/env.py:53 (`vocab.rng = random.Random(seed)`), wordle/env/game.py:178-179
 * (`rng.choice`), maze/env/env.py:187-212 + randomness.py:9-19 (`random.seed` / `random.choice`).
 *
 * Layout of an mt buffer for N streams (uint32):  mt[624][N] then idx[N]   -> 625*N*4 bytes.
 * ------------------------------------------------------------------------------------------ */
size_t lmrl_mt_bytes(int n);
/* seed streams where mask_d[i] != 0 (mask_d NULL = all). seeds_d[i] = |seed| as uint64
 * (key length 1 if < 2^32 else 2, exactly as CPython's init_by_array over 32-bit limbs). */
int lmrl_mt_seed(void *mt_d, const uint64_t *seeds_d, const uint8_t *mask_d, int n, void *stream);
/* test hook: out_d[k*n + i] = k-th genrand_uint32 of stream i, k < n_out (advances the streams). */
int lmrl_mt_stream(void *mt_d, uint32_t *out_d, int n_out, int n, void *stream);
/* test hook: out_d[k*n + i] = _randbelow(bounds_d[k]) of stream i (random.Random.choice index). */
int lmrl_mt_randbelow(void *mt_d, const uint32_t *bounds_d, uint32_t *out_d, int n_draws, int n, void *stream);

/* ------------------------------------------------------------------------------------------
 * Batched Wordle MDP — replaces WordleEnvironment / WordleGame / Vocabulary / WordleState
 * (wordle/env/env.py:39-55, wordle/env/game.py:53-296) stepped in lock-step for N envs.
 *
 * Game-state buffer for N envs (uint32, SoA):
 *   forb[5][N]  bit c set  <=> letter c is NOT_HERE at position i   (game.py:17-20 CharKnowledge)
 *   must[5][N]  bit c set  <=> letter c is HERE at position i       (neither bit = POSSIBLE)
 *   n_filtered[N]           len(vocab.filtered_vocab)
 *   n_actions[N]            len(action_history)
 *   hist[6][N]              packed guesses (5 x 5 bit letters), 0xFFFFFFFF = not a 5-letter a-z string
 * -> 19*N*4 bytes.
 * ------------------------------------------------------------------------------------------ */
typedef struct lmrl_wordle_ctx lmrl_wordle_ctx;

#define LMRL_WORDLE_BAD_GUESS 0xFFFFFFFFu /* guess word that is not 5 letters a-z (game.py:214) */
/* pack 5 letters 'a'..'z': sum (c_i - 'a') << (5*i) */

/* words5: V*5 bytes, file order (Vocabulary.from_file, game.py:163-170). Host pointer. */
lmrl_wordle_ctx *lmrl_wordle_create(const char *words5, int n_words, int require_words_in_vocab,
                                    float bad_word_reward);
void lmrl_wordle_destroy(lmrl_wordle_ctx *ctx);
size_t lmrl_wordle_state_bytes(int n);
/* WordleEnvironment.reset for envs with mask_d[i] != 0 (NULL = all): seeds the env's stream in mt_d
 * (env.py:53) and sets the all-POSSIBLE state with filtered_vocab = all_vocab (game.py:208-211). */
int lmrl_wordle_reset(lmrl_wordle_ctx *ctx, void *state_d, void *mt_d, const uint64_t *seeds_d,
                      const uint8_t *mask_d, int n, void *stream);
/*
 * One env.step for every env with active_d[i] != 0 (NULL = all)  (env.py:46-50 -> game.py:213-222).
 *   guess_d[i]  packed guess or LMRL_WORDLE_BAD_GUESS
 *   obs_d[i]    bits [3k,3k+3) k<5: 0 none / 1 'g' / 2 'y' / 3 'b' (transition_sequence()[-1], game.py:273-288);
 *               bits [16,19): number of symbols (0 for an invalid action -> observation text "\n")
 *   reward_d[i] -1 / 0 (int in the reference) or bad_word_reward (game.py:290-293)
 *   flags_d[i]  bit0 done (game.py:295-296), bit1 a target was drawn (valid transition),
 *               bit2 reward is the float bad_word_reward,
 *               bit3 `rng.choice(filtered_vocab)` was reached with an EMPTY filtered vocabulary: the reference raises IndexError
 *                    there (game.py:178-179, 219); the env is marked done and the Python face raises IndexError (unreachable from a
 *                    state produced by reset/step: the sampled target always stays in its own filtered list)
 * Inactive envs: outputs untouched, state untouched.
 */
int lmrl_wordle_step(lmrl_wordle_ctx *ctx, void *state_d, void *mt_d, const uint32_t *guess_d,
                     const uint8_t *active_d, uint32_t *obs_d, float *reward_d, uint8_t *flags_d, int n,
                     void *stream);
/* Kernel form of lmrl_wordle_step: 0 (default) = by batch size — one wavefront per env for the lock-step rollout batches (the 64 lanes sweep
 * the vocabulary together), one LANE per env for env-only workloads from 65 536 envs up (262 144 with a vocabulary above 1024 words): every
 * state access of a wave is one coalesced line, no cross-lane work; 1 / 2 force either.  Both forms are bit-identical (tests/test_gpu_envs.py). */
int lmrl_wordle_set_variant(lmrl_wordle_ctx *ctx, int variant);
/* Export the knowledge state as the reference's 26x5 trits (0 NOT_HERE / 1 POSSIBLE / 2 HERE) + counts. */
int lmrl_wordle_export_state(const void *state_d, uint8_t *trits_d /* [N][26][5] */,
                             uint32_t *n_filtered_d, uint32_t *n_actions_d, int n, void *stream);

/* ------------------------------------------------------------------------------------------
 * Batched Maze MDP — replaces MazeEnv.reset/step, update_position and the reward functions
 * (maze/env/env.py:104-131,161-214).  Text rendering of observations stays on the host.
 *
 * State buffer for N envs (int32, SoA): pos_r[N] pos_c[N] goal_r[N] goal_c[N] num_steps[N] -> 20*N bytes.
 * ------------------------------------------------------------------------------------------ */
typedef struct lmrl_maze_ctx lmrl_maze_ctx;

#define LMRL_MAZE_LEFT 0   /* 'move left\n'  (0,-1)   env.py:94-99 */
#define LMRL_MAZE_RIGHT 1  /* 'move right\n' (0,+1) */
#define LMRL_MAZE_UP 2     /* 'move up\n'    (-1,0) */
#define LMRL_MAZE_DOWN 3   /* 'move down\n'  (+1,0) */
#define LMRL_MAZE_OTHER 4  /* any other action string */

#define LMRL_MAZE_KIND_OBS 0      /* history + observation, truncated to last_k      env.py:182-184 */
#define LMRL_MAZE_KIND_FAILURE 1  /* (Text("Failure\n"),), -1.0, True                env.py:164-165 */
#define LMRL_MAZE_KIND_SUCCESS 2  /* (Text("Success\n"),), reward, True              env.py:173-174 */
#define LMRL_MAZE_KIND_OBS_ONLY 3 /* illegal action string: history := (observation,) env.py:179-180 */

/* grid: rows*cols bytes (1 = wall); valid_goals: n_goals (row, col) int32 pairs; max_steps < 0 = None;
 * rewards: {at_goal, illegal_action, otherwise} e.g. standard_reward = {0,-4,-1} (env.py:109-115). */
lmrl_maze_ctx *lmrl_maze_create(const uint8_t *grid, int rows, int cols, const int32_t *valid_goals,
                                int n_goals, int max_steps, const float rewards[3]);
void lmrl_maze_destroy(lmrl_maze_ctx *ctx);
size_t lmrl_maze_state_bytes(int n);
/* MazeEnv.reset (env.py:186-214). opt_goal_d / opt_init_d: [N][2] int32, row < 0 = not given (then drawn with
 * random.choice from the env's freshly seeded stream, goal first).  NULL = never given. */
int lmrl_maze_reset(lmrl_maze_ctx *ctx, void *state_d, void *mt_d, const uint64_t *seeds_d,
                    const int32_t *opt_goal_d, const int32_t *opt_init_d, const uint8_t *mask_d, int n,
                    void *stream);
/* MazeEnv.step (env.py:161-184). walls_d: bit0 right, bit1 left, bit2 above, bit3 below (env.py:59). */
int lmrl_maze_step(lmrl_maze_ctx *ctx, void *state_d, const uint8_t *action_d, const uint8_t *active_d,
                   float *reward_d, uint8_t *done_d, uint8_t *kind_d, uint8_t *walls_d, int n, void *stream);

/* ------------------------------------------------------------------------------------------
 * Chess env stepping (csrc/chess.hip, csrc/chess_rules.h): python-chess as the reference's chess env uses it
 * (llm_rl_scripts/chess/env/env.py:28-185 — Board(fen), push_san, san, fen, is_checkmate, is_game_over), one game per lane.
 * The opponent (Stockfish over UCI, env.py:157-170) is a host process pool and hands its moves back in UCI form.
 * Position buffer: lmrl_chess_pos_bytes() per game.  Strings are fixed-pitch, NUL-terminated: FEN LMRL_CHESS_FEN_BYTES, SAN action
 * LMRL_CHESS_ACTION_BYTES, UCI move 8 bytes.
 * ------------------------------------------------------------------------------------------ */
#define LMRL_CHESS_FEN_BYTES 96
#define LMRL_CHESS_ACTION_BYTES 16
#define LMRL_CHESS_ILLEGAL 0   /* unparsable / illegal / ambiguous SAN: reward -1, not done, board unchanged   env.py:109-118 */
#define LMRL_CHESS_MOVED 1     /* legal move, game goes on: the opponent is to move */
#define LMRL_CHESS_GAME_OVER 2 /* legal move ended the game: reward 1 if checkmate else 0, done                env.py:121-125 */
#define LMRL_CHESS_NULL_MOVE 3 /* '--' / 'Z0': reward -1, done                                                env.py:111-113 */
size_t lmrl_chess_pos_bytes(void);
/* chess.Board(fen) for n games; ok_d[i] = 0 for a malformed FEN */
int lmrl_chess_reset(void *pos_d, const char *fens_d, uint8_t *ok_d, int n, void *stream);
/* the agent's half of ChessEnv.step: actions_d [n][LMRL_CHESS_ACTION_BYTES] SAN (blanks already removed); result_d = LMRL_CHESS_*,
 * fen_out_d [n][LMRL_CHESS_FEN_BYTES] = board.fen() after the half-step; uci_out_d [n][8] (optional) = the played move in UCI form (what the
 * engine is told, env.py:121), empty for an illegal action; inactive games: result 255, nothing written */
int lmrl_chess_agent_step(void *pos_d, const char *actions_d, const uint8_t *active_d, float *reward_d, uint8_t *done_d, uint8_t *result_d,
                          char *fen_out_d, char *uci_out_d, int n, void *stream);
/* the opponent's half: uci_d [n][8] a legal engine move; san_out_d = board.san(move) BEFORE it is played (env.py:163), reward -1 if the agent
 * is mated, done = is_game_over(); ok_d = 0 if the move is not legal in the position */
int lmrl_chess_opponent_step(void *pos_d, const char *uci_d, const uint8_t *active_d, float *reward_d, uint8_t *done_d, uint8_t *ok_d, char *san_out_d,
                             char *fen_out_d, int n, void *stream);
/* board.legal_moves (+ board.san of each), board.fen(), is_check / is_checkmate / is_game_over ... of n games in one launch
 * (env.py:157-170 random opponent `random.choice(list(board.legal_moves))`, :150-155 `is_game_over`; chess/eval move-accuracy loops):
 * uci_out_d [n][lmrl_chess_max_moves()][8], san_out_d [n][lmrl_chess_max_moves()][LMRL_CHESS_ACTION_BYTES], count_d [n], status_d [n] with
 * the bits of lmrl_chess_host_status, fen_out_d [n][LMRL_CHESS_FEN_BYTES]; every output is optional (NULL) */
int lmrl_chess_max_moves(void);
int lmrl_chess_describe(const void *pos_d, char *uci_out_d, char *san_out_d, int32_t *count_d, uint8_t *status_d, char *fen_out_d, int n, void *stream);
/* the same rules on ONE position in host memory (CPU-tier tests, oracle comparisons; no GPU needed).  Negative return = bad argument. */
int lmrl_chess_host_from_fen(const char *fen, void *pos);
int lmrl_chess_host_fen(const void *pos, char *out);                                  /* -> length */
int lmrl_chess_host_legal_moves(const void *pos, char *out_uci /* [256][8] */, char *out_san /* [256][16] or NULL */);   /* -> count */
int lmrl_chess_host_agent_step(void *pos, const char *san, float *reward, int *done);  /* -> LMRL_CHESS_* */
int lmrl_chess_host_opponent_step(void *pos, const char *uci, char *san_out, float *reward, int *done);   /* -> 1 if the move was legal */
/* bit0 check, bit1 checkmate, bit2 is_game_over, bit3 insufficient material, bit4 stalemate, bit5 fivefold repetition, bit6 75-move rule */
int lmrl_chess_host_status(const void *pos);

/* ------------------------------------------------------------------------------------------
 * Per-token RL reductions (wavefront-shuffle kernels).
 * ------------------------------------------------------------------------------------------ */
/*
 * GAE over compacted action tokens, one chain per row — get_advantages_and_returns
 * (LLM_RL/algorithms/ppo/base_interface.py:253-293) fed exactly as at :586-606:
 *   values_d  [B][Lv]   per-chain values incl. the bootstrap slot (values_chains, :554-570)
 *   rewards_d [B][L]    per-token rewards (already KL-penalised, :580-584), L = Lv-1
 *   sta_d     [B][L]    should_take_action (uint8)
 *   len_d     [B]       valid length of each chain (<= L)
 * Outputs adv_d/ret_d [B][L]: advantage / return scattered back to action-token positions, 0 elsewhere
 * (:635-645).  The state / next-state pairing is get_action_state_next_state_idxs (:230-243).
 */
int lmrl_gae(const float *values_d, const float *rewards_d, const uint8_t *sta_d, const int32_t *len_d,
             float *adv_d, float *ret_d, int b, int l, float gamma, float lam, void *stream);
/* Reward-to-go over action tokens — get_rtg + MCData scatter (LLM_RL/algorithms/mc_returns/data.py:10-14,49-74). */
int lmrl_rtg(const float *rewards_d, const uint8_t *sta_d, const int32_t *len_d, float *rtg_d, int b, int l,
             float gamma, void *stream);
/* Masked whitening over ALL selected elements of x (ppo/base_interface.py:245-251, :609-615):
 * y = (x-mean)*rsqrt(var+1e-8) (+mean if !shift_mean) where mask, y = x elsewhere. In place allowed.
 * moments_d (3 doubles: sum, sumsq, count) is caller scratch; when `moments_only` != 0 only the local
 * moments are produced (so ranks can all-reduce them) and `lmrl_whiten_apply` finishes the job. */
int lmrl_whiten_moments(const float *x_d, const uint8_t *mask_d, double *moments_d, size_t n, void *stream);
/* The rollout-sized chain GAE -> whiten in two launches instead of four: lmrl_gae_moments = lmrl_gae that also leaves per-workgroup partial moments
 * of the advantages on action slots (partials_d: 3 doubles x lmrl_gae_moments_partials(b, l), 0 = shape not covered: use lmrl_gae +
 * lmrl_whiten_moments); lmrl_whiten_apply_partials adds the partials in a fixed order in every workgroup and applies (one rank); ranks that
 * all-reduce the moments call lmrl_whiten_finish (partials -> the 3 moments) + the collective + lmrl_whiten_apply.  Same numbers as the separate
 * calls up to the fp64 association order of the moments (ppo/base_interface.py:245-293, 609-615). */
int lmrl_gae_moments_partials(int b, int l);
int lmrl_gae_moments(const float *values_d, const float *rewards_d, const uint8_t *sta_d, const int32_t *len_d, float *adv_d, float *ret_d, int b, int l,
                     float gamma, float lam, double *partials_d, void *stream);
int lmrl_whiten_finish(const double *partials_d, int n_partials, double *moments_d, void *stream);
int lmrl_whiten_apply_partials(const float *x_d, const uint8_t *mask_d, const double *partials_d, int n_partials, float *y_d, size_t n, int shift_mean,
                               void *stream);
int lmrl_whiten_apply(const float *x_d, const uint8_t *mask_d, const double *moments_d, float *y_d, size_t n,
                      int shift_mean, void *stream);

/* ------------------------------------------------------------------------------------------
 * Rollout records -> PPO data without leaving the device (csrc/ppo_data.hip).
 * Replaces the host side of PPOInference.get_ppo_data_from_token_trajectory_chain
 * (LLM_RL/algorithms/ppo/base_interface.py:464-669) as the task scripts feed it
 * (llm_rl_scripts/wordle/ppo/train_ppo_gpt2.py:301-353), PPOData.block / PPODataset
 * (LLM_RL/algorithms/ppo/data.py:9-114) and the masks GPT2PPOTrain.step derives from a batch
 * (base_interface.py:172-228).  The forwards between lmrl_ppo_block and lmrl_ppo_shape, lmrl_gae and
 * lmrl_whiten_* between lmrl_ppo_shape and lmrl_ppo_unroll are the caller's.
 * ------------------------------------------------------------------------------------------ */
/* n token trajectories = the fields of TokenTrajectory (LLM_RL/environment.py:329-380) as the rollout engines record
 * them, grouped into n_chains TokenTrajectoryChains (environment.py:383-420).  All pointers are device pointers. */
typedef struct {
    const int32_t *tokens;      /* [n][cap] */
    const uint8_t *is_action;   /* [n][cap] */
    const float *reward;        /* [n][cap] */
    const int32_t *n_tok;       /* [n] tokens of each trajectory (<= cap) */
    const uint8_t *done;        /* [n_chains] `done` of each chain's LAST trajectory (every earlier one must be not done, :310) */
    const int32_t *chain;       /* [n] chain of trajectory k, or NULL: trajectory k is chain k (n_chains == n) */
    const int32_t *pos;         /* [n] where trajectory k starts in its chain's concatenation (sum of earlier len - 1), or NULL: 0 */
    const uint8_t *last;        /* [n] 1 when trajectory k ends its chain, or NULL: all 1 */
    int32_t n, cap, n_chains;
} lmrl_ppo_records;
/* Per trajectory: effective length = min(n_tok, max_len) (max_len <= 0: no truncation; Truncation.RIGHT, :500-512), rows that predict a next
 * token (len - 1) and action tokens (is_action[1:len]), with their exclusive scans.  cnt_d [2n] scratch; off_rows_d / off_act_d [n + 1];
 * meta_d [8] = {rows, action tokens, longest effective length, pad ids found below a length, action tokens CUT by max_len, trajectories that
 * continue a chain but start with an action token, longest chain concatenation (max of pos + len - 1), 0}.  Words 4 and 5 are the two arms of the
 * reference's 'trajectory truncation error' assert (CombinedTokenTrajectoryChain.from_token_trajectory_chain, :318-327): a caller mirrors it by
 * refusing the batch when either is non-zero; word 6 is the `lc` lmrl_ppo_shape / lmrl_ppo_unroll need. */
int lmrl_ppo_count(const lmrl_ppo_records *rec, int max_len, int pad, int32_t *cnt_d, int32_t *off_rows_d, int32_t *off_act_d, int32_t *meta_d,
                   void *stream);
/* block_sequences(Padding.RIGHT, pad) to width tf + initialize_attn_mask_pos_ids + the compacted list of rows k * tf + t (t < len - 1) with
 * their next-token targets (the LM head of token_logprobs_from_logits, :396-403, runs on these rows only): ids / am / pos [n][tf],
 * rows_idx / tgt [meta[0]]. */
int lmrl_ppo_block(const lmrl_ppo_records *rec, int max_len, int pad, int tf, const int32_t *off_rows_d, int32_t *ids_d, uint8_t *am_d, int32_t *pos_d,
                   int32_t *rows_idx_d, int32_t *tgt_d, void *stream);
/* :543-584 + the PPOData rows: logprobs_d / init_logprobs_d [meta[0]] in row-list order, values_d [n][tf].  Writes per chain
 * (concatenated trajectories, row pitch lc / lc + 1, lc >= every chain's length): values incl. the bootstrap slot value * (1 - done), KL-shaped
 * rewards, should_take_action, the chain length; kls_d [meta[1]] = exp(lr) - 1 - lr over the action tokens in trajectory order; and the blocked
 * dataset rows of width tp (>= the longest effective length): ds_ids [n][tp], ds_sta / ds_logprobs / ds_values [n][tp - 1]. */
int lmrl_ppo_shape(const lmrl_ppo_records *rec, int max_len, int tf, const int32_t *off_rows_d, const int32_t *off_act_d, const float *logprobs_d,
                   const float *init_logprobs_d, const float *values_d, float kl_weight, int lc, float *chain_values_d, float *chain_rewards_d,
                   uint8_t *chain_sta_d, int32_t *chain_len_d, float *kls_d, int pad, int tp, int32_t *ds_ids_d, uint8_t *ds_sta_d, float *ds_logprobs_d,
                   float *ds_values_d, void *stream);
/* unroll_arr (:335-343, :635-660): chain rows of advantages / returns [n_chains][lc] back to ds_adv / ds_ret [n][tp - 1] (0 past a length) */
int lmrl_ppo_unroll(const lmrl_ppo_records *rec, int max_len, int tf, int lc, const float *chain_adv_d, const float *chain_ret_d, int tp, float *ds_adv_d,
                    float *ds_ret_d, void *stream);
/* The task scripts' length rule on single-trajectory chains (llm_rl_scripts/wordle/ppo/train_ppo_gpt2.py:323-341, same loop in the chess / maze
 * PPO scripts): while a trajectory has more than three texts (maximal runs of equal is_action flags) and n_tok >= max_len, drop its last two texts,
 * add (sum of their rewards) * gamma to the reward of the previous action (its last token) and clear `done`; keep[k] = 0 for trajectories left
 * with fewer than three texts or still >= max_len tokens (the script's two `continue`s).  reward_d [n][cap] is updated IN PLACE (pass a copy of a
 * rollout engine's record), n_tok_out_d / done_out_d / keep_d [n]; meta_d [2] = {trajectories shortened, trajectories skipped}. */
int lmrl_ppo_truncate_turns(const uint8_t *is_action_d, const int32_t *n_tok_d, const uint8_t *done_d, int n, int cap, int max_len, double gamma,
                            float *reward_d, int32_t *n_tok_out_d, uint8_t *done_out_d, uint8_t *keep_d, int32_t *meta_d, void *stream);
/* idx_d[j] = the j-th index k with flags_d[k] != 0, increasing; count_d[0] = how many (row compaction of a record after lmrl_ppo_truncate_turns:
 * lmrl_gather_rows_bytes by idx_d); one workgroup */
int lmrl_compact_flags(const uint8_t *flags_d, int n, int32_t *idx_d, int32_t *count_d, void *stream);
/* initialize_attn_mask_pos_ids (JaxSeq; call sites base_interface.py:190-195): am = ids != pad, pos = max(cumsum(am) - 1, 0); [b][t].
 * am_next_f32_d (optional) [b][t - 1] = float(am[:, 1:]), the mask ppo_loss_fn multiplies in (base_interface.py:208-214). */
int lmrl_seq_mask_pos(const int32_t *ids_d, int pad, uint8_t *am_d, int32_t *pos_d, float *am_next_f32_d, int b, int t, void *stream);
/* The same two arrays for right-padded rows of KNOWN lengths (am[r][i] = i < len[r]): a device-resident PPO dataset hands the train step the masks its
 * data build used instead of re-deriving them from `ids != pad` (lmrl_gym_amd.algorithms.ppo_device.DevicePPODataset.batch). */
int lmrl_len_mask_pos(const int32_t *len_d, int b, int t, uint8_t *am_d, int32_t *pos_d, void *stream);
/* out[i] = in[i] + c (row-list indices re-based to a chunk of the batch; lengths from row counts) */
int lmrl_add_i32(const int32_t *in_d, int c, int32_t *out_d, int n, void *stream);
/* rows r = row * t + i (i < t - 1) with sta[row][i] && am[row][i + 1] (am NULL: sta alone), increasing, + tgt = ids[row][i + 1] (tgt NULL: no
 * targets) — the rows of [b * t, d] hidden states whose LM-head outputs a masked loss reads.  cnt_d [b] scratch, off_d [b + 1] (off_d[b] = count),
 * idx_d / tgt_d capacity b * (t - 1). */
int lmrl_masked_rows(const uint8_t *sta_d, const uint8_t *am_d, const int32_t *ids_d, int b, int t, int32_t *cnt_d, int32_t *off_d, int32_t *idx_d,
                     int32_t *tgt_d, void *stream);
/* out_d [n + 1] = exclusive prefix sums of in_d [n] (out_d[n] = the total); one workgroup */
int lmrl_exclusive_scan_i32(const int32_t *in_d, int32_t *out_d, int n, void *stream);
/* dst[i] = src[idx[i]] for rows of row_bytes bytes (a shuffled batch of a device-resident dataset: ppo/data.py:88-98) */
int lmrl_gather_rows_bytes(const void *src_d, const int32_t *idx_d, void *dst_d, int n, long row_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * GPT-2 rollout forward with a persistent per-env KV cache (csrc/gpt2.hip).
 * Replaces the model side of GPT2PPOPolicy.act / GPT2ValuePolicy.act
 * (LLM_RL/algorithms/ppo/gpt2/interface.py:507-546, value_rl_base/gpt2/interface.py:281-320), whose
 * transformer lives in JaxSeq / HF-Flax (third party).
 *
 * Engine weight layout (all device pointers, caller-owned):
 *   wte  bf16 [vocab_padded][d]  (rows >= vocab zero)        wpe  bf16 [n_pos][d]
 *   per layer, 12 pointers in this order:
 *     ln1_g f32[d], ln1_b f32[d], w_qkv bf16[3d][d], b_qkv f32[3d], w_proj bf16[d][d], b_proj f32[d],
 *     ln2_g f32[d], ln2_b f32[d], w_fc bf16[dff][d], b_fc f32[dff], w_fc2 bf16[d][dff], b_fc2 f32[d]
 *   i.e. every matrix is stored [out][in] (HF Conv1D weights transposed once at load time).
 * KV cache: bf16 [n_layer][2 (K,V)][B][tmax][n_head*64] (token-major: one env's keys for all heads are one contiguous stream).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t n_layer, n_head, d_model, d_ff, vocab, vocab_padded, n_pos;
    float ln_eps;
} lmrl_gpt2_config;

typedef struct lmrl_gpt2 lmrl_gpt2;

lmrl_gpt2 *lmrl_gpt2_create(const lmrl_gpt2_config *cfg, const void *wte_d, const void *wpe_d, const float *lnf_g_d,
                            const float *lnf_b_d, const void *const *layer_ptrs /* host array [n_layer*12] */);
void lmrl_gpt2_destroy(lmrl_gpt2 *m);
/* The caller has overwritten the weight tensors handed to lmrl_gpt2_create IN PLACE (the online loops copy the trainer's parameters into the
 * rollout engine after every round: train_ppo_gpt2.py `policy.set_params(...)`): re-derive the model-owned LayerNorm-folded copies on `stream`.
 * Pointers held by sessions and captured hipGraphs stay valid. */
int lmrl_gpt2_refresh(lmrl_gpt2 *m, void *stream);
size_t lmrl_gpt2_kv_bytes(const lmrl_gpt2 *m, int b, int tmax);
size_t lmrl_gpt2_ws_bytes(const lmrl_gpt2 *m, int b, int c);
/*
 * Forward B envs x C token slots (C = 1 decode, C = 8 or 16 chunked prefill).  Env b contributes cnt_d[b] <= C new
 * tokens tokens_d[b*C + j] at positions len_d[b] + j; their K/V rows are appended to the cache and len_d[b] is
 * advanced by cnt_d[b].  last_hidden_d (bf16 [B][d], optional): ln_f of each env's LAST new token (row untouched
 * when cnt_d[b] == 0).  all_hidden_d (bf16 [B*C][d], optional): ln_f of every row.
 * `flags` (LMRL_FWD_*, 0 = defaults) select per-CALL variants; there is no process-wide forward state, so sessions with
 * different settings may be interleaved on any streams.  Every variant computes the same function (the LN-folded and
 * stand-alone paths within bf16 rounding of each other, ragged vs padded bit-identically).
 */
#define LMRL_FWD_LN_STANDALONE 1u  /* stand-alone LayerNorm launches instead of LN folded into the neighbouring GEMMs (A/B, cross-check) */
#define LMRL_FWD_RAGGED_ALWAYS 2u  /* run on the compacted rows (sum of cnt_d) whatever the batch size; default: only when b*c >= 2048 */
#define LMRL_FWD_RAGGED_NEVER  4u  /* always run all b*c slots */
#define LMRL_FWD_ATTN_VALU     8u  /* per-(env, head) multi-round-trip attention kernels instead of the default ones (cross-check) */
#define LMRL_FWD_KV_FROM_GEMM  16u /* decode: the qkv GEMM epilogue appends the new K/V rows, the attention kernel only reads (A/B: attention faster, GEMM slower, net slower) */
#define LMRL_FWD_ATTN_ITEMS2   64u /* decode attention: 2 (env, head) items per wave on a grid of half as many waves (A/B of the resident-size grid; bit-identical) */
#define LMRL_FWD_ATTN_ITEMS3   128u /* ... 3 items per wave */
#define LMRL_FWD_SKINNY       (1u << 30) /* single-token decode of <= 16 sequences: the layer's Dense products on the skinny-M kernels (csrc/skinny_gemm.h: one MFMA row block,
                                          * K split over the waves of a workgroup, every load issued up front) — 7-16 us -> ~3 us per product at 8 rows; same formulas, K summed in
                                          * 8 slices (not bit-identical to the tile kernels: a per-session choice) */
#define LMRL_FWD_FULL_LAST_LAYER 32u /* chunk forwards: run the last layer's projection + MLP on every row (default: only on each env's last new token — the only row whose hidden state is returned; A/B and cross-check, bit-identical) */
/* bits 16-31: reserved — the product library rejects them (LMRL_ERR_ARG). */
/* bits 8-15: LMRL_FWD_SHARED_PREFIX(n) — positions [0, n) of EVERY env's cache hold the same K/V rows as env 0's (the caller copied them
 * with lmrl_gpt2_kv_broadcast and has not reset the cache since): the decode attention then reads them from env 0 (one L2-resident copy)
 * instead of once per env.  Results are bit-identical; a wrong promise reads env 0's prefix for every env. */
#define LMRL_FWD_SHARED_PREFIX(n) (((unsigned)(n) & 0xffu) << 8)
int lmrl_gpt2_forward(lmrl_gpt2 *m, void *kv_d, int tmax, void *ws_d, const int32_t *tokens_d, const int32_t *cnt_d,
                      int32_t *len_d, int b, int c, void *last_hidden_d, void *all_hidden_d, unsigned flags, void *stream);

/*
 * Shared prompt prefix: copy positions [0, n_pos) of the single env of a 1-env KV session (src) into all `b` envs of dst,
 * set dst_len_d[i] = n_pos and (optionally) broadcast the prefix's last hidden state.  All envs of a lock-step Wordle batch
 * start from the same header text ('Wordle:\n', wordle/env/env.py:8), so its K/V are computed once per episode instead of
 * once per env; the copied rows are bit-identical to what a per-env forward writes.
 */
int lmrl_gpt2_kv_broadcast(const lmrl_gpt2 *m, const void *src_kv_d, int src_tmax, void *dst_kv_d, int dst_tmax, int b, int n_pos,
                           const void *src_hidden_d, void *dst_hidden_d, int32_t *dst_len_d, void *stream);

/*
 * Prompt-prefix cache (indexed form of lmrl_gpt2_kv_broadcast): env i of dst starts from the prompt held by row idx_d[i] of the session
 * `src` (src_b envs, same model): K/V rows [0, src_len_d[row]), dst_len_d[i] = that length and (optionally) the row's last hidden state.
 * idx_d[i] < 0 leaves env i with an empty cache.  Text envs whose observations form a finite set (Maze with last_k = 1: one prompt per
 * (goal, cell), llm_rl_scripts/maze/env/env.py:8-81,182-184) prefill each distinct prompt once per set of weights instead of once per
 * env per turn (the reference re-encodes and re-runs the whole prompt on every act(), ppo/gpt2/interface.py:519-546); the copied rows are
 * the rows a per-env prefill of the same tokens writes.
 */
int lmrl_gpt2_kv_gather(const lmrl_gpt2 *m, const void *src_kv_d, int src_b, int src_tmax, const int32_t *src_len_d, const void *src_hidden_d,
                        const int32_t *idx_d, void *dst_kv_d, int dst_tmax, int b, void *dst_hidden_d, int32_t *dst_len_d, void *stream);

/*
 * Indexed prompt prefix: the copy-free form of lmrl_gpt2_kv_gather.  lmrl_gpt2_kv_attach sets dst_len_d[i] = pfx_n_d[i] = the prompt length
 * of prefix row idx_d[i] (0 for idx < 0), copies that row's last hidden state and (order_d, optional, [b]) lists the envs grouped by prefix
 * row.  Single-token decode forwards then go through lmrl_gpt2_forward_prefixed: positions [0, n_d[i]) of env i are READ from row row_d[i]
 * of the prefix session's cache — the bytes of a prompt exist once however many envs stand on it, and with order_d each XCD walks envs that
 * share prompts, so those rows are served by its L2 — while the generated tokens' rows go to / come from env i's own cache at their
 * absolute positions.  Results are bit-identical to the copying form (same rows, same order of arithmetic).
 * The reference re-runs the whole prompt per env per act() (ppo/gpt2/interface.py:519-546).
 */
typedef struct {
    const void *kv_d;           /* the prefix session's cache: lmrl_gpt2_kv_bytes(m, n_rows, tmax) */
    int32_t n_rows, tmax;
    const int32_t *row_d;       /* [b] prefix row behind env i (< 0: none) — the idx_d given to lmrl_gpt2_kv_attach */
    const int32_t *n_d;         /* [b] prefix length of env i — pfx_n_d of lmrl_gpt2_kv_attach */
    const int32_t *order_d;     /* [b] launch order (a permutation of the envs) or NULL */
} lmrl_kv_prefix;
int lmrl_gpt2_kv_attach(const lmrl_gpt2 *m, int src_b, const int32_t *src_len_d, const void *src_hidden_d, const int32_t *idx_d, int b,
                        int dst_tmax, void *dst_hidden_d, int32_t *dst_len_d, int32_t *pfx_n_d, int32_t *order_d, void *stream);
int lmrl_gpt2_forward_prefixed(lmrl_gpt2 *m, void *kv_d, int tmax, void *ws_d, const int32_t *tokens_d, const int32_t *cnt_d,
                               int32_t *len_d, int b, void *last_hidden_d, const lmrl_kv_prefix *pfx, unsigned flags, void *stream);

/* C[m][n] = A[m][k] . W[n][k]^T + bias[n]; A, W bf16.  epilogue: 0 bf16, 1 gelu_new->bf16, 2 f32 += (residual),
 * 3 f32, 4 relu->bf16.  Used for the value heads (heads/linear_head.py:112-119, heads/mlp_head.py:139-148). */
int lmrl_gemm_bf16(const void *a_d, const void *w_d, const float *bias_d, void *c_d, int m, int n, int k, int lda,
                   int ldc, int n_store, int epilogue, void *stream);
/* same with an explicit row pitch of W (ldw elements, >= k; 0 = dense): operands whose natural pitch is a large power of two (the
 * transposed operands of the train step's dW products, k = B*T) are padded so that the rows of a tile do not all start in one HBM channel */
int lmrl_gemm_bf16_ld(const void *a_d, const void *w_d, const float *bias_d, void *c_d, int m, int n, int k, int lda, int ldw, int ldc, int n_store,
                      int epilogue, void *stream);
/* c[m][n_store] fp32 = resid[m][..] + a . w^T + bias with the residual operand in ANOTHER buffer (row pitch ldr): the residual add of a
 * transformer block (x_mid = x + attn.c_proj(att), x_out = x_mid + mlp.c_proj(g)) folded into the projection's epilogue when the block's
 * input / middle / output streams are kept as separate tensors — the train step's forward (its LayerNorm backward reads all three). */
int lmrl_gemm_bf16_resid(const void *a_d, const void *w_d, const float *bias_d, const float *resid_d, int ldr, float *c_d, int m, int n, int k, int lda,
                         int ldw, int ldc, int n_store, void *stream);

/* Train step, bf16-matmul mode: Dense products whose epilogue writes the NEXT kernel's bf16 operand, so that no stand-alone elementwise pass
 * re-reads an [B*T][n] fp32 tensor (n a multiple of 128, k of 64; a, w as for lmrl_gemm_bf16_ld).  The HF GPT-2 block the reference differentiates
 * (ppo/gpt2/interface.py:111-133, ilql/gpt2/interface.py:139-177):
 *   gelu_dual : c fp32 [m][ldc] = a . w^T + bias (mlp.c_fc pre-activation, read again by the backward) AND act bf16 [m][ldact] = gelu_new(c)
 *   qkv_heads : attn.c_attn straight into the flash kernels' per-head bf16 matrices q | k | v = [3][batch*heads][pad64(t)][64] (plane_elems
 *               apart, from lmrl_flash_attn_stage_ptrs), q scaled by 1/8; follow with lmrl_flash_attn_finish_staging; m = batch * t rows
 *   gelu_bwd  : c bf16 [m][ldc] = (a . w^T) * gelu_new'(pre[m][n]) — d(pre-activation) of mlp.c_fc as the bf16 dy operand of its backward products */
int lmrl_gemm_bf16_gelu_dual(const void *a_d, const void *w_d, const float *bias_d, float *c_d, int ldc, void *act_bf16_d, int ldact, int m, int n, int k,
                             int lda, int ldw, void *stream);
int lmrl_gemm_bf16_qkv_heads(const void *a_d, const void *w_d, const float *bias_d, void *q_heads_d, long plane_elems, int m, int k, int lda, int ldw,
                             int heads, int t, void *stream);
int lmrl_gemm_bf16_gelu_bwd(const void *a_d, const void *w_d, const float *pre_d, int ldpre, void *c_bf16_d, int ldc, int m, int n, int k, int lda,
                            int ldw, void *stream);
/* The two entries above with the pre-activation stored ROUNDED TO bf16 ([m][ldpre] bf16 elements, ldpre a multiple of 8): the gelu backward is its
 * only reader (ppo/gpt2/interface.py:72-211 and ilql/gpt2/interface.py:88-367 differentiate HF-Flax GPT-2's gelu_new; its bf16 mode keeps every
 * activation in bf16) — half the bytes written by the forward and read by the backward. */
int lmrl_gemm_bf16_gelu_dual_prebf16(const void *a_d, const void *w_d, const float *bias_d, void *pre_bf16_d, int ldpre, void *act_bf16_d, int ldact, int m,
                                     int n, int k, int lda, int ldw, void *stream);
int lmrl_gemm_bf16_gelu_bwd_prebf16(const void *a_d, const void *w_d, const void *pre_bf16_d, int ldpre, void *c_bf16_d, int ldc, int m, int n, int k,
                                    int lda, int ldw, void *stream);

/* Vocabulary-wide heads (the Q heads of ILQL / MC, heads/mlp_head.py:139-148; the tied LM head of PPO) in the bf16-matmul mode: logits written ONCE,
 * in bf16 ([m][ldc], columns [0, n_store)), with everything the cross-entropy / take_along_axis terms need taken from the fp32 accumulators in the
 * same launch — partials_d [m][lmrl_gemm_bf16_ce_slots(m, n, k)] (max, sum exp) pairs -> lmrl_lse_from_partials gives log-sum-exp (and the target's
 * log-probability), tgt_logit_d[r] = logit[r][targets_d[r]] in fp32.  Replaces an fp32 [m][V] logits tensor + a pass over it
 * (optax.softmax_cross_entropy_with_integer_labels forward, ilql/base_interface.py:57-66 gathers).  lmrl_ce_bwd_bf16_inplace then turns the bf16
 * logits into the bf16 d(logits) operand of the head's backward products in place (padding rows / columns zeroed).
 * logits_bf16_d NULL (inference: PPOInference.forward's token log-probabilities, ppo/base_interface.py:396-403): no logits are stored at all. */
int lmrl_gemm_bf16_ce_slots(int m, int n, int k);
int lmrl_gemm_bf16_ce(const void *a_d, const void *w_d, const float *bias_d, void *logits_bf16_d, int ldc, int m, int n, int k, int lda, int ldw, int n_store,
                      const int32_t *targets_d, float *tgt_logit_d, void *partials_d, void *stream);
int lmrl_lse_from_partials(const void *partials_d, int nslots, int rows, const float *tgt_logit_d, float *lse_d, float *logprob_d, void *stream);
int lmrl_ce_bwd_bf16_inplace(void *logits_bf16_d, long ld, int vocab, const float *lse_d, const int32_t *targets_d, const float *coef_ce_d,
                             const float *coef_gather_d, int rows, int rows_dst, void *stream);

/* Split-K form for products with few output tiles and a long K (the train step's weight-gradient products dW = x^T . dy: K = B*T):
 * S copies of the 128 x 128 tile grid each accumulate a slice of K into fp32 partials in ws_d, a fixed-order reduce then writes
 * c (=|+=) their sum — deterministic.  lmrl_gemm_bf16_splitk_ws_bytes returns 0 when the shape is better served by lmrl_gemm_bf16_ld. */
size_t lmrl_gemm_bf16_splitk_ws_bytes(int m, int n, int k);
int lmrl_gemm_bf16_splitk(const void *a_d, const void *w_d, void *c_d, int m, int n, int k, int lda, int ldw, int ldc, int n_store, int accumulate,
                          void *ws_d, void *stream);
/* the same product with a bias and no accumulation, c[m][n] = sum_k a[m][k] w[n][k] + bias[n] (any shape lmrl_gemm_bf16_splitk_ws_bytes has a plan for) — the MLP output
 * projection of the bf16x3 rollout mode at decode size (K' = 3 d_ff = 9216 on 48 tiles) */
int lmrl_gemm_bf16_splitk_bias(const void *a_d, const void *w_d, const float *bias_d, void *c_d, int m, int n, int k, int lda, int ldw, int ldc,
                               void *ws_d, void *stream);

/* The same product on the operands as their producers staged them — a_d = x [k][lda], w_d = dy [k][ldw], both bf16 with k = B*T rows:
 * c[m][n] (=|+=) sum_kk a[kk][m] * w[kk][n]  (m, n multiples of 128, k of 64; a 128 x 128 split-K plan must exist: lmrl_gemm_bf16_splitk_ws_bytes).
 * No transposed copy of x or dy is needed (the kernel gathers MFMA operands with ds_read_b64_tr_b16). */
int lmrl_gemm_bf16_splitk_kmajor(const void *a_d, const void *w_d, void *c_d, int m, int n, int k, int lda, int ldw, int ldc, int n_store, int accumulate,
                                 void *ws_d, void *stream);
/* colsum_d[c] (=|+=) sum_r src[r][c] of a staged bf16 matrix [rows][ld_src] — a Dense layer's bias gradient from its staged dy; ws_d: at least
 * ceil(rows / 64) * cols floats.  Deterministic (fixed partition and order). */
int lmrl_colsum_bf16(const void *src_d, long ld_src, int rows, int cols, float *colsum_d, int accumulate, float *ws_d, void *stream);

/* TOOLS ONLY (tools/bench_gemm.py tile-configuration sweeps; never called by the package): forces a GEMM tile configuration,
 * 0 = the shape policy, 1 = the round-1 register-staged kernels, 10.. = fixed tiles. */
void lmrl_gemm_set_variant(int v);

/* ------------------------------------------------------------------------------------------
 * Fused LM-head + sampling (csrc/sampler.hip).  Replaces logits[:, -1] -> warpers -> jax.random.categorical in
 * the reference's generation loop and the ILQL perturbation logits = pi_beta + beta*min(q1,q2)
 * (LLM_RL/algorithms/value_rl_base/gpt2/generation.py:97-119).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    float temperature;      /* <= 0: greedy (do_sample=False) */
    int32_t top_k;          /* <= 0: off; needs logits_out_d (1 .. 64 on the Philox path — policy-only and ILQL value-policy heads alike: scratch of
                             * the rare rows the fused candidate selection hands back, NOT the logits, unless `flags` asks for them; otherwise
                             * the logits are materialised there) */
    uint64_t seed;          /* Philox key */
    uint32_t step;          /* counter word: one per sampled token position */
    float steer_strength;   /* added to logit[steer_tok_d[row]] (synthetic workloads only; 0 = off) */
    float beta;             /* ILQL logit perturbation weight */
    int32_t pad_token;      /* written for inactive rows */
    const uint32_t *epoch_d;/* optional DEVICE word = 4th Philox counter word (0 when NULL): fresh noise per hipGraph replay */
    float top_p;            /* nucleus mass in (0, 1); <= 0 or >= 1 = off.  Applied after temperature and top_k (HF warper order):
                             * the smallest prefix of the descending-sorted tokens whose cumulative probability reaches top_p */
    int32_t rng;            /* LMRL_RNG_PHILOX (0, default): the package's own counter stream above.
                             * LMRL_RNG_JAX: the stream of `jax.random.categorical(key, logits[m, vocab])` as HF-Flax `_sample` calls it for the
                             * reference (ppo/gpt2/interface.py:524-535): `seed` = the call's PRNG key, (key[0] << 32) | key[1]; Gumbel word of
                             * (row, column) = element row * vocab + column of jax.random.gumbel(key, (m * vocab,)) — threefry2x32-20 over the
                             * iota split in halves (csrc/threefry.h); logits / temperature by division; `step` and `epoch_d` are unused (the
                             * caller walks the key schedule: lmrl_gym_amd/jax_prng.py).  All m rows of the call form ONE noise array. */
    int32_t flags;          /* LMRL_SAMPLE_WANT_LOGITS: the caller reads logits_out_d after the call — materialise them even where the fused top-k
                             * path would not (same tokens either way) */
} lmrl_sample_params;
#define LMRL_SAMPLE_WANT_LOGITS 1
#define LMRL_RNG_PHILOX 0
#define LMRL_RNG_JAX 1
/* host faces of the LMRL_RNG_JAX stream: the Threefry-2x32 (20 rounds) block function and words [i0, i0 + count) of
 * jax.random.bits(key, (n,), uint32) — for known-answer tests without a GPU */
void lmrl_threefry2x32(const uint32_t key[2], const uint32_t ctr[2], uint32_t out[2]);
int lmrl_jax_random_bits_host(const uint32_t key[2], uint32_t n, uint32_t i0, uint32_t count, uint32_t *out);

/* workspace of lmrl_lm_head_sample (per-(row, tile) partials or top-k candidate records + the flagged-row list).  One workspace per concurrent stream. */
size_t lmrl_sample_ws_bytes(int m, int vocab_padded);
/* byte offset, inside that workspace, of the fused top-k path's flagged-row list: int32 words — [0] = rows the exactness check handed back to the
 * materialised path in the LAST call (diagnostics / tests), 15 reserved words, 64 per-row-block flags, then the row indices */
size_t lmrl_sample_fb_offset(int m, int vocab_padded);
/* hidden_d bf16 [m][d] . wte_d bf16 [vocab_padded][d]^T -> token_d[m] (+ logprob_d[m] under the sampling
 * distribution).  Optional ILQL operands: q_hidden{1,2}_d bf16 [m][d] (= relu(dense1(h)) of each Q head),
 * q_w{1,2}_d bf16 [vocab_padded][d] (dense2 kernels, [out][in]), q_b{1,2}_d f32 [vocab_padded].
 * logits_out_d (optional f32 [m][vocab_padded]) receives the combined (untempered) logits — except on the fused warper path
 * (either random stream, one to three operands, temperature > 0, and 0 < top_k <= 256 [above 64: top_k <= vocab_padded / 128] and / or
 * 0 < top_p < 1: `FlaxTopKLogitsWarper` / `FlaxTopPLogitsWarper` of train_ppo_gpt2.py:98-99, 218-227 inside the LM-head epilogue): there the
 * epilogue keeps the 8 largest logits of every (row, 128-column tile) — and, for top_p without top_k, the tile's probability mass —, a reduce
 * kernel finds the row's k-th largest among them / the nucleus' crossing token, checks that no tile can hide a logit that matters, and draws;
 * logits_out_d is written only for the 128-row blocks of rows that fail the check (they are re-done from materialised logits: same tokens as
 * the materialised path in every case).  LMRL_SAMPLE_WANT_LOGITS in p->flags: always materialise. */
int lmrl_lm_head_sample(const void *hidden_d, const void *wte_d, const void *q_hidden1_d, const void *q_w1_d,
                        const float *q_b1_d, const void *q_hidden2_d, const void *q_w2_d, const float *q_b2_d, int m,
                        int d_model, int vocab, int vocab_padded, const lmrl_sample_params *p, const int32_t *steer_tok_d,
                        const uint8_t *active_d, int32_t *token_d, float *logprob_d, float *logits_out_d, void *ws_d,
                        void *stream);
/* Generation bookkeeping for any tokenizer (the per-token host loop of HF `generate` / GPT2PPOPolicy.act, ppo/gpt2/interface.py:
 * 527-535): for every live row append sampled_d[row] to out_tokens_d[row][out_len_d[row]++]; a row stops (active_d[row] = 0) at
 * eos_token (< 0: none) or at `cap` tokens; next_tok_d / next_cnt_d are the inputs of the next single-token forward. */
int lmrl_gen_accept(const int32_t *sampled_d, uint8_t *active_d, int32_t *out_tokens_d /* [n][cap] */, int32_t *out_len_d,
                    int32_t *next_tok_d, int32_t *next_cnt_d, int eos_token, int cap, int n, void *stream);
/* Sample from materialised logits (temperature, top-k, top-p) with the same random stream. */
int lmrl_sample_logits(const float *logits_d, int ld, int m, int vocab, const lmrl_sample_params *p,
                       const uint8_t *active_d, int32_t *token_d, float *logprob_d, void *stream);
/* the same, after adding p->steer_strength to logits_d[row][steer_tok_d[row]] IN PLACE (synthetic workloads; the fused path does this in
 * its epilogue) — the sampling step of the fp32 rollout mode, whose LM head is a plain fp32 GEMM */
int lmrl_sample_logits_steer(float *logits_d, int ld, int m, int vocab, const lmrl_sample_params *p, const int32_t *steer_tok_d,
                             const uint8_t *active_d, int32_t *token_d, float *logprob_d, void *stream);

/* ------------------------------------------------------------------------------------------
 * fp32 rollout mode (csrc/attn_cached_f32.hip + the fp32 train-step kernels below): the reference's DEFAULT rollout arithmetic is float32
 * (llm_rl_scripts/wordle/bc/eval_bc_gpt2.py:34,69); `lmrl_gym_amd.gpt2_f32_engine.GPT2EngineF32` sequences these calls — lmrl_embed_fwd,
 * lmrl_layernorm_fwd, lmrl_sgemm (exact fp32 MFMA), lmrl_gelu_fwd — around an fp32 K/V cache.
 *   lmrl_chunk_begin_f32  pos[b*c + j] = len[b] + j for j < cnt[b]; padding slots get position 0 and token id 0
 *   lmrl_attn_cached_f32  one layer's attention for the c new tokens of every env: qkv_d fp32 [b*c][3*H*64]; kcache_d / vcache_d fp32
 *                         [b][tmax][H*64]; query j sees the cached positions [0, len[b]) and the chunk's tokens [0, j]; appends the new
 *                         K / V rows at len[b] + j; out_d fp32 [b*c][H*64] (padding slots untouched).  c == 1 (single-token decode) runs the
 *                         batched-load kernel (one wave per (env, head), 32 cached positions requested before any arithmetic).
 *   lmrl_attn_cached_f32_split3  the same, and (split3_out_d != NULL) the output ALSO as the bf16 three-term split operand
 *                         [b*c][3*H*64] = [hi | lo | hi] of the projection GEMM of the bf16x3 mode (lmrl_split3_bf16's layout and rounding)
 *   lmrl_chunk_end_f32    last_d[b] = x_d[b*c + cnt[b] - 1] (envs with cnt == 0 keep theirs), then len[b] += cnt[b]
 * ------------------------------------------------------------------------------------------ */
int lmrl_chunk_begin_f32(const int32_t *len_d, const int32_t *cnt_d, int32_t *ids_d, int32_t *pos_d, int b, int c, int n_pos, void *stream);
int lmrl_attn_cached_f32(const float *qkv_d, float *kcache_d, float *vcache_d, const int32_t *len_d, const int32_t *cnt_d, float *out_d, int b, int c,
                         int n_head, int tmax, void *stream);
int lmrl_attn_cached_f32_split3(const float *qkv_d, float *kcache_d, float *vcache_d, const int32_t *len_d, const int32_t *cnt_d, float *out_d,
                                void *split3_out_d, int b, int c, int n_head, int tmax, void *stream);
int lmrl_chunk_end_f32(const float *x_d, const int32_t *cnt_d, float *last_d, int32_t *len_d, int b, int c, int d, void *stream);

/* ------------------------------------------------------------------------------------------
 * On-device token <-> game bookkeeping for lock-step Wordle rollouts (csrc/wordle_tokens.hip).
 * Replaces the per-turn host work of interact_environment + GPT2PPOPolicy.act
 * (LLM_RL/environment.py:180-206, LLM_RL/algorithms/ppo/gpt2/interface.py:519-546) and the text
 * (de)formatting of llm_rl_scripts/wordle/env/env.py:7-26.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t newline;          /* id of '\n' (= eos of an action, train_ilql_gpt2.py:393) */
    int32_t pad;
    int32_t letter_first[26]; /* 'a'..'z' as the first symbol of a line (no leading space) */
    int32_t letter_sp[26];    /* ' a'..' z' */
    int32_t sym_first[3];     /* 'g','y','b' */
    int32_t sym_sp[3];        /* ' g',' y',' b' */
    int32_t header[8];        /* ids of "Wordle:\n" (env.py:8) */
    int32_t n_header;
} lmrl_wordle_tokens;

/* Per-env rollout record, all device pointers owned by the caller; N envs, cap = traj_cap, G = max_new_tokens.
 * tokens/is_action/reward are exactly the fields of TokenTrajectory (LLM_RL/environment.py:329-380). */
typedef struct {
    int32_t *tokens;      /* [N][cap] */
    uint8_t *is_action;   /* [N][cap] */
    float *reward;        /* [N][cap]  reward on the last token of each action Text */
    int32_t *n_tok;       /* [N] */
    int32_t *gen;         /* [N][G] ids sampled for the current action */
    int32_t *gen_len;     /* [N] */
    uint8_t *gen_active;  /* [N] still generating the current action */
    uint8_t *env_done;    /* [N] episode finished */
    uint8_t *pend_newline;/* [N] action ended without eos: '\n' forced (interface.py:541) */
    int32_t *n_steps;     /* [N] env steps taken */
    float *ep_reward;     /* [N] sum of rewards */
} lmrl_wordle_traj;

typedef struct lmrl_wordle_tok_ctx lmrl_wordle_tok_ctx;
/* token_class (host, [vocab]): bits 0-24 up to five 5-bit letters, bits 25-27 their count (7 = any char other than
 * a-z / whitespace, or more than 5 letters, or whitespace between letters), bit 28 non-' ' whitespace before the first
 * letter (anywhere when the token has no letters), bit 29 non-' ' whitespace after the last letter. */
lmrl_wordle_tok_ctx *lmrl_wordle_tok_create(const lmrl_wordle_tokens *tokens, const uint32_t *token_class, int vocab,
                                            int max_new_tokens, int traj_cap);
void lmrl_wordle_tok_destroy(lmrl_wordle_tok_ctx *c);
/* episode start: trajectory := header tokens; first model chunk (chunk_tok_d [N][8], chunk_cnt_d [N]) := header */
int lmrl_wordle_tok_begin(lmrl_wordle_tok_ctx *c, const lmrl_wordle_traj *tr, int32_t *chunk_tok_d, int32_t *chunk_cnt_d,
                          int n, void *stream);
/* k-th sampled token of the current action: record it; envs that hit eos / max_new_tokens stop generating;
 * next_tok_d/next_cnt_d (and active_d) describe the next single-token decode step. */
int lmrl_wordle_tok_accept(lmrl_wordle_tok_ctx *c, const lmrl_wordle_traj *tr, const int32_t *sampled_d, int k,
                           int32_t *next_tok_d, int32_t *next_cnt_d, uint8_t *active_d, int n, void *stream);
/* action tokens -> packed guess for lmrl_wordle_step (+ active mask = episode not done); appends the action to the record */
int lmrl_wordle_tok_guess(lmrl_wordle_tok_ctx *c, const lmrl_wordle_traj *tr, uint32_t *guess_d, uint8_t *active_d, int n,
                          void *stream);
/* env outputs -> reward placement, observation tokens, done flags, and the next model chunk
 * ([last unforwarded action token][forced '\n'] + observation tokens; count 0 for finished envs) */
int lmrl_wordle_tok_observe(lmrl_wordle_tok_ctx *c, const lmrl_wordle_traj *tr, const uint32_t *obs_d, const float *reward_d,
                            const uint8_t *flags_d, int32_t *chunk_tok_d, int32_t *chunk_cnt_d, int n, void *stream);
/* synthetic workloads: token spelling letter k (k = 5: '\n') of a scripted packed guess, for lmrl_sample_params.steer;
 * k < 0 writes all six positions at once, steer_d = int32 [6][n] */
int lmrl_wordle_tok_steer(lmrl_wordle_tok_ctx *c, const uint32_t *scripted_guess_d, int k, int32_t *steer_d, int n,
                          void *stream);

/* ------------------------------------------------------------------------------------------
 * On-device token <-> game bookkeeping for lock-step Maze rollouts (csrc/maze_tokens.hip), for histories of one item
 * (last_k = 1, the reference's Maze harness: maze/bc/fully_observed_bc.py:230-237): a turn's prompt is the observation of the current
 * cell, a pure function of (goal, cell) (maze/env/env.py:8-81), so prompts come from a token table instead of the host tokenizer, and the
 * generated ids are decoded through a per-token byte table and matched against the four action strings (env.py:94-99) on the device.
 * Replaces the per-turn host work of interact_environment + GPT2PPOPolicy.act (LLM_RL/environment.py:180-206,
 * ppo/gpt2/interface.py:519-546).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t *pos;         /* [N][T]  (row << 16 | col) the turn's observation was rendered from */
    int32_t *gen;         /* [N][T][G] generated ids of the turn's action (before decoding) */
    int32_t *gen_len;     /* [N][T] */
    uint8_t *action;      /* [N][T]  LMRL_MAZE_LEFT..LMRL_MAZE_OTHER the decoded text maps to */
    float *reward;        /* [N][T] */
    uint8_t *kind;        /* [N][T]  LMRL_MAZE_KIND_* of the step's result */
    int32_t *n_turns;     /* [N] env steps taken */
    uint8_t *live;        /* [N] episode still running */
    float *ep_reward;     /* [N] */
    int32_t *obs_idx;     /* [N] row of the observation table for the current turn (-1: finished) */
    int32_t *out_tok;     /* [N][G] generation buffer of the current turn (lmrl_gen_accept) */
    int32_t *out_len;     /* [N] */
    uint8_t *gen_active;  /* [N] */
    uint8_t *act;         /* [N] action code of the current turn (input of lmrl_maze_step) */
    uint8_t *stepping;    /* [N] env takes part in the current turn (active mask of lmrl_maze_step) */
} lmrl_maze_traj;

typedef struct lmrl_maze_tok_ctx lmrl_maze_tok_ctx;
/* obs_tok [n_obs][obs_cap] / obs_len [n_obs]: token ids of every observation text, row = goal_slot * rows * cols + r * cols + c;
 * goal_slot [rows * cols]: slot of a goal cell (-1: not a goal of this table).  tok_bytes [vocab][16] / tok_blen [vocab]: the decoded
 * bytes of each token id; blen 0 = skipped (special tokens under skip_special_tokens=True, value_rl_base/base_interface.py:126),
 * 255 = not representable (non-ASCII or longer than 16 bytes: such a token cannot be part of an action string).  T = max_turns. */
lmrl_maze_tok_ctx *lmrl_maze_tok_create(const int32_t *obs_tok, const int32_t *obs_len, int n_obs, int obs_cap, const int32_t *goal_slot,
                                        int rows, int cols, const uint8_t *tok_bytes, const uint8_t *tok_blen, int vocab,
                                        int max_new_tokens, int max_turns);
void lmrl_maze_tok_destroy(lmrl_maze_tok_ctx *c);
/* episode start (after lmrl_maze_reset): every env live, counters cleared */
int lmrl_maze_tok_begin(lmrl_maze_tok_ctx *c, const lmrl_maze_traj *tr, int n, void *stream);
/* turn start: obs_idx from the env state (pos, goal), position recorded, generation buffers cleared, gen_active = live */
int lmrl_maze_tok_turn(lmrl_maze_tok_ctx *c, const lmrl_maze_traj *tr, const void *state_d, int n, void *stream);
/* chunk j (chunk tokens per env) of the current prompts for a prefill forward: chunk_tok_d [N][chunk], chunk_cnt_d [N] */
int lmrl_maze_tok_prompt(lmrl_maze_tok_ctx *c, const lmrl_maze_traj *tr, int j, int chunk, int32_t *chunk_tok_d, int32_t *chunk_cnt_d,
                         int n, void *stream);
/* generated ids -> action code: text = concat(token bytes, special tokens skipped); `text.removesuffix('\n') + '\n'`
 * (the Maze scripts' out_str_process) compared with the four action strings; records ids and code */
int lmrl_maze_tok_action(lmrl_maze_tok_ctx *c, const lmrl_maze_traj *tr, int n, void *stream);
/* The finished episodes as PPO records (lmrl_ppo_records): one token trajectory per transition — observation ids ++ action ids, the step reward on the
 * action's last token.  Action ids: a legal action (its post-processed text is a key of the action dict) as the tokenizer's encoding of that key when
 * lmrl_maze_tok_set_actions was called; otherwise the generated ids minus special tokens (byte_ids != 0 — a tokenizer whose ids are the text's UTF-8
 * bytes: the bytes of every decoded token), + newline_tok when the decoded text does not end in a newline.  Chained per episode,
 * as the online scripts build them (llm_rl_scripts/maze/ppo/train_ppo_online.py:444-465 + LLM_RL/environment.py:359-370).  off_d [n + 1] =
 * exclusive scan of tr->n_turns (lmrl_exclusive_scan_i32); rows off_d[e] + t; outputs [off_d[n]][cap] / [off_d[n]], done_d / chain_total_d [n]
 * (a chain's concatenated length: the GAE row pitch is their maximum).  The first n of the engine's n_envs envs are exported. */
int lmrl_maze_tok_ppo_records(lmrl_maze_tok_ctx *c, const lmrl_maze_traj *tr, const void *state_d, int n, int n_envs, const int32_t *off_d, int newline_tok,
                              int byte_ids, int cap, int32_t *tokens_d, uint8_t *is_action_d, float *reward_d, int32_t *n_tok_d, int32_t *chain_d, int32_t *pos_d, uint8_t *last_d,
                              uint8_t *done_d, int32_t *chain_total_d, void *stream);
/* The same export for item windows (last_k > 1; run after an episode of the history-mode loop below), in the form of the partially observed online
 * script (llm_rl_scripts/maze/ppo/partially_observed_ppo_online.py:372-398): per transition ONE non-action text — the window's item texts joined by
 * single spaces — then the action text, reward on the action's last token; chained per episode.  The window of every turn is rebuilt from the record
 * (tr->pos / action / kind: maze/env/env.py:179-184).  Token ids: the first item's own encoding, every later item's encoding behind the joining space
 * (lmrl_maze_tok_set_spaced), a legal action as the encoding of its dict key (lmrl_maze_tok_set_actions), any other action string as its generated
 * ids (as above) — or, with byte_ids != 0 (a tokenizer whose ids are the text's UTF-8 bytes), as the bytes of its decoded text, i.e. exactly
 * tokenizer.encode(text) also when the policy spelled it with multi-byte tokens.  last_k <= 64.  tokens_d == NULL: first pass — only n_tok_d (untruncated lengths) .. chain_total_d are written, the caller sizes
 * `cap` by the longest row and calls again. */
int lmrl_maze_tok_ppo_records_hist(lmrl_maze_tok_ctx *c, const lmrl_maze_traj *tr, const void *state_d, int n, int n_envs, const int32_t *off_d, int last_k,
                                   int newline_tok, int byte_ids, int cap, int32_t *tokens_d, uint8_t *is_action_d, float *reward_d, int32_t *n_tok_d, int32_t *chain_d,
                                   int32_t *pos_d, uint8_t *last_d, uint8_t *done_d, int32_t *chain_total_d, void *stream);
/* host tables for the above: obs_sp_tok [n_obs][obs_sp_cap] / obs_sp_len [n_obs] = tokenizer.encode(' ' + observation text) of every row of the
 * observation table, act_sp_tok [4][act_sp_cap] = tokenizer.encode(' ' + action string), its length in the last slot */
int lmrl_maze_tok_set_spaced(lmrl_maze_tok_ctx *c, const int32_t *obs_sp_tok, const int32_t *obs_sp_len, int obs_sp_cap, const int32_t *act_sp_tok,
                             int act_sp_cap);
/* Histories of more than one item (MazeEnv(last_k > 1): `(history + [action] + [observation])[-last_k:]`, maze/env/env.py:182-184; the prompt is the
 * window's text, left-truncated to max_input_length tokens, ppo/gpt2/interface.py:519-524; partially_observed_bc.py:241 runs last_k = 40) on a
 * persistent per-env KV cache.  Per env: the episode's token history, the token offset of every item, and which part of the history is in the
 * cache.  "Append" turns (window still growing: prompt t + 1 = prompt t ++ action ++ observation token for token) forward only the action's
 * unforwarded tail + the new observation; "re-prefill" turns (window slid or truncated: every position shifts under GPT-2's absolute position
 * embeddings) forward the window again from position 0.  The caller schedules the kind per turn (`reprefill`) from static bounds and says how many
 * tokens per env the turn's chunk forwards cover (`feed_budget`); an env whose window moved in an append turn (e.g. restarted by an illegal action
 * string) is simply re-forwarded within that budget; flags[0] bit 0 = it did not fit, bit 1 = history buffer / item table overflow.  len0_d / len1_d: the cache-length arrays of the
 * policy's (and, optional, the value base's) KV session, kept in step by these calls. */
typedef struct {
    int32_t *hist;        /* [N][hcap] ids of every item of the episode, in order */
    int32_t *item_off;    /* [N][max_items + 1] token offset of item k (item_off[k + 1]: its end) */
    int32_t *n_items;     /* [N] */
    int32_t *feed_start;  /* [N] first history token that goes through the model this turn */
    int32_t *feed_len;    /* [N] */
    int32_t *cache_len;   /* [N] history tokens [base, base + cache_len) sit in the KV cache at positions [0, cache_len) */
    int32_t *base;        /* [N] */
    int32_t *prompt_len;  /* [N] tokens of the current prompt */
    int32_t *win_floor;   /* [N] first item of the window since the env last returned (observation,) alone (LMRL_MAZE_KIND_OBS_ONLY, env.py:179-180) */
    int32_t *flags;       /* [1] */
    int32_t hcap, max_items;
} lmrl_maze_hist;
int lmrl_maze_hist_begin(const lmrl_maze_hist *h, int32_t *len0_d, int32_t *len1_d, int n, void *stream);
/* after lmrl_maze_tok_turn: the current observation joins the history; window / truncation / feed range of every env */
int lmrl_maze_hist_observe(lmrl_maze_tok_ctx *c, const lmrl_maze_traj *tr, const lmrl_maze_hist *h, int last_k, int max_input_length, int reprefill,
                           int feed_budget, int32_t *len0_d, int32_t *len1_d, int n, void *stream);
/* chunk j of this turn's feed for a prefill forward: chunk_tok_d [N][chunk], chunk_cnt_d [N] */
int lmrl_maze_hist_chunk(const lmrl_maze_hist *h, int j, int chunk, int32_t *chunk_tok_d, int32_t *chunk_cnt_d, int n, void *stream);
/* act_tok (host) [4][act_cap]: the tokenizer's encoding of 'move left\n' / 'move right\n' / 'move up\n' / 'move down\n' (LMRL_MAZE_LEFT ..), the length of
 * each in its last slot — what a later prompt holds for a legal action, whatever ids the policy generated to spell it */
int lmrl_maze_tok_set_actions(lmrl_maze_tok_ctx *c, const int32_t *act_tok, int act_cap);
/* after lmrl_maze_tok_action: the action joins the history — a legal action as its encoding above, an illegal string (it never reaches a later
 * prompt: the env answers with the observation alone) as its generated ids; the cache is cut back to the prompt + the forwarded generated ids
 * that ARE the action's ids */
int lmrl_maze_hist_action(lmrl_maze_tok_ctx *c, const lmrl_maze_traj *tr, const lmrl_maze_hist *h, int32_t *len0_d, int32_t *len1_d, int n, void *stream);
/* outputs of lmrl_maze_step -> record (reward, kind), counters, live &= !done */
int lmrl_maze_tok_result(lmrl_maze_tok_ctx *c, const lmrl_maze_traj *tr, const float *reward_d, const uint8_t *done_d,
                         const uint8_t *kind_d, int n, void *stream);

/* ------------------------------------------------------------------------------------------
 * Per-kernel-class timing with HIP events on the launch stream (used by bench.py for the live roofline figure).
 * mask bit t enables tag t; `work` accumulates the algorithmic flops (GEMM-class tags) or bytes (others) per launch.
 * ------------------------------------------------------------------------------------------ */
void lmrl_prof_enable(unsigned mask);
void lmrl_prof_reset(void);
int lmrl_prof_n_tags(void);
const char *lmrl_prof_tag_name(int tag);
int lmrl_prof_read(int tag, double *total_ms, double *total_work, long long *launches);

/* ------------------------------------------------------------------------------------------
 * fp32 train-step building blocks (csrc/sgemm_f32.hip, csrc/train_ops.hip, csrc/losses.hip).
 * Together they restate GPT2PPOTrain._step / GPT2ILQLTrain._step (LLM_RL/algorithms/ppo/gpt2/interface.py:72-211,
 * LLM_RL/algorithms/ilql/gpt2/interface.py:88-367) in float32, the reference's default parameter/activation dtype.
 * ------------------------------------------------------------------------------------------ */
/* C[b] = alpha*op(A[b]).op(B[b]) + beta*C[b] (+bias[n]); row-major; op(A) MxK (trans_a: stored KxM), op(B) KxN
 * (trans_b: stored NxK); batch index = outer*nb_inner + inner with separate element strides. Exact fp32 (MFMA f32). */
int lmrl_sgemm(int trans_a, int trans_b, int m, int n, int k, float alpha, const float *a_d, int lda, long sa_outer,
               long sa_inner, const float *b_d, int ldb, long sb_outer, long sb_inner, float beta, float *c_d, int ldc,
               long sc_outer, long sc_inner, int nb_outer, int nb_inner, const float *bias_d, void *stream);
/* A/B hook: 0 = auto (128x128 tiles when m, n >= 128), 1 = always the 64x64-tile kernel */
void lmrl_sgemm_set_variant(int v);
/* TOOLS / TESTS ONLY: bit 0 = LayerNorm forward, bit 1 = fused LayerNorm backward on the strided 4-byte-access kernels (A/B of the 16-byte-access
 * register-row kernels) */
void lmrl_train_ops_set_variant(int v);
/* TOOLS / TESTS ONLY: 1 = top-k / top-p sampling on the strided-row warper kernel also where the register-row kernel applies (A/B, equality tests) */
void lmrl_sampler_set_variant(int v);
/* TOOLS / TESTS ONLY: 1 = lmrl_gae / lmrl_rtg on the round-1 LDS-compaction kernel and the scalar whitening kernels, 2 = the 64-lane register
 * kernel also for chains of <= 128 slots, 3 = one slot per lane in the 16-lane DPP-row kernel (A/B, cross-checks of the two-slots-per-lane form) */
void lmrl_rl_reduce_set_variant(int v);
/* TOOLS / TESTS ONLY: bit 0 = bf16 flash-attention forward, bit 1 = dQ, bit 2 = dK/dV on the round-3 kernels (A/B and equality tests of the
 * round-4 sweeps: 128 queries per workgroup, two query groups per wave, global_load_lds tile ring) */
void lmrl_flash_set_variant(int v);
/* x[r] = wte[ids[r]] + wpe[pos[r]].  vocab > 0: wte has `vocab` rows and an id outside [0, vocab) embeds as a ZERO row — the pad id of a tokenizer
 * whose `<|pad|>` is the first id after the model's vocabulary (train_ppo_gpt2.py:124-126; the reference resizes the embedding and forces those
 * logits to -inf, ppo/gpt2/interface.py:330, so that id is never sampled and only ever sits at masked positions). */
int lmrl_embed_fwd(const float *wte_d, const float *wpe_d, const int32_t *ids_d, const int32_t *pos_d, float *x_d, int rows, int d, int vocab,
                   void *stream);
/* dwte[ids[r]] += dx[r], dwpe[pos[r]] += dx[r] without atomics (one owner wave per distinct index, fixed order: bit-reproducible).
 * live_d (optional, uint8 [rows]): rows with flag 0 are skipped — the padded positions of a right-padded batch (attention_mask == 0), whose dx is
 * exactly zero and which would otherwise all pile onto the pad id's owner wave.  t_row > 0 (the T of a [B, T] batch): a row whose flag is 0
 * but whose NEXT position in the same sequence has flag 1 stays live (the PPO / BC losses read row t wherever attention_mask[t + 1] is set, so
 * with left padding or holes its dx is not zero); right-padded batches: exactly the flagged rows.  vocab > 0: ids outside [0, vocab) own no
 * wte row (lmrl_embed_fwd) and are skipped. */
int lmrl_embed_bwd(const float *dx_d, const int32_t *ids_d, const int32_t *pos_d, const uint8_t *live_d, float *dwte_d, float *dwpe_d, int rows, int d,
                   int vocab, int t_row, void *stream);
/* Row compaction for the vocabulary-wide heads of the train steps: the losses read the Q / policy logits only on rows whose mask is set
 * (should_take_action x attention mask: ilql/base_interface.py:22-119, ppo/base_interface.py:72-142), so a head runs on the gathered rows
 * dst[i] = src[idx[i]] and its input gradient goes back with dst[idx[i]] (=|+=) src[i].  idx_d holds DISTINCT rows (no atomics). */
int lmrl_gather_rows_f32(const float *src_d, const int32_t *idx_d, float *dst_d, int n, int d, void *stream);
int lmrl_scatter_rows_f32(const float *src_d, const int32_t *idx_d, float *dst_d, int n, int d, int accumulate, void *stream);
int lmrl_layernorm_fwd(const float *x_d, const float *g_d, const float *b_d, float *y_d, float *mean_d, float *rstd_d, int rows, int d,
                       float eps, void *stream);
/* LayerNorm / gelu forward that ALSO write the bf16 copy of their output (row pitch ldb elements) — the operand of the GEMM that consumes it in
 * the bf16-matmul train mode (saves the separate lmrl_cast_bf16 pass).  y_d may be NULL: only the bf16 copy is kept (it also serves, transposed,
 * as the operand of the layer's dW product). */
int lmrl_layernorm_fwd_staged(const float *x_d, const float *g_d, const float *b_d, float *y_d, float *mean_d, float *rstd_d, void *yb_d, long ldb,
                              int rows, int d, float eps, void *stream);
/* x[r] += resid[r] (resid_d NULL: no add) and then LayerNorm of the updated row, in one pass: y (fp32, optional) and / or yb (bf16 [rows][ldb], the
 * operand of the consuming GEMM), mean / rstd for the backward.  The residual adds of an HF GPT-2 block (x + attn(ln_1 x), x + mlp(ln_2 x)) folded
 * into the LayerNorm that follows each of them (ln_2, the next block's ln_1, ln_f): x_d is the projection's raw output on entry, the residual
 * stream on exit. */
int lmrl_layernorm_add_fwd(float *x_d, const float *resid_d, const float *g_d, const float *b_d, float *y_d, float *mean_d, float *rstd_d, void *yb_d,
                           long ldb, int rows, int d, float eps, void *stream);
/* the same with the output written only as the "bf16 x 3" operand [rows][3 d] = [hi | lo | hi] (lmrl_split3_bf16's row format) — LayerNorm + split in one
 * pass for GPT2EngineF32's bf16x3 matmul mode (d a multiple of 256, <= 1280); lmrl_gelu_split3: gelu_new + split likewise ([rows][3 cols]). */
int lmrl_layernorm_add_fwd_split3(float *x_d, const float *resid_d, const float *g_d, const float *b_d, float *mean_d, float *rstd_d, void *split_d,
                                  int rows, int d, float eps, void *stream);
int lmrl_gelu_split3(const float *x_d, int rows, int cols, void *split_d, void *stream);
int lmrl_gelu_fwd_staged(const float *x_d, float *y_d, void *yb_d, long ldb, int rows, int cols, void *stream);
/* dx (=|+=) LN backward; dy_xhat_d (optional [rows][d]) receives dy*xhat whose column sum is d gamma */
int lmrl_layernorm_bwd(const float *dy_d, const float *x_d, const float *g_d, const float *mean_d, const float *rstd_d, float *dx_d,
                       float *dy_xhat_d, int rows, int d, int accumulate_dx, void *stream);
/* LayerNorm backward with the gamma / beta gradients reduced in the same pass (row slabs -> per-slab partial rows -> fixed-order sum):
 * dx (+)= ..., dgamma (+)= sum_r dy*xhat, dbeta (+)= sum_r dy.  d_model in {128, 256, 768, 1024, 1280, 1600} (the per-lane column
 * registers are compile-time); other widths use lmrl_layernorm_bwd + lmrl_colsum. */
int lmrl_layernorm_bwd_fused_supported(int d);
size_t lmrl_layernorm_bwd_fused_ws_bytes(int rows, int d);
int lmrl_layernorm_bwd_fused(const float *dy_d, const float *x_d, const float *g_d, const float *mean_d, const float *rstd_d, float *dx_d,
                             float *dgamma_d, float *dbeta_d, int rows, int d, int accumulate_dx, int accumulate_dg, float *ws_d,
                             void *dxb_d /* optional: bf16 copy of the final dx, row pitch ldb — the dy operand of the next linear backward */,
                             long ldb, void *stream);
size_t lmrl_colsum_ws_bytes(int cols);
int lmrl_colsum(const float *x_d, int rows, int cols, int ld, float *out_d, int accumulate, float *ws_d, void *stream);
/* out[c] (=|+=) sum_r wrow[r * ldw] * x[r][c]: the weight gradient x^T . dy of a Dense layer with ONE output unit (heads/linear_head.py:112-119 as
 * the PPO value head, the V head of heads/mlp_head.py:139-148) — a matrix-vector product; ws_d as for lmrl_colsum. */
int lmrl_colsum_weighted(const float *x_d, int rows, int cols, int ld, const float *wrow_d, int ldw, float *out_d, int accumulate, float *ws_d,
                         void *stream);
/* elementwise ops: in-place calls are supported (y_d == x_d, dx_d == dy_d, out_d == x_d or y_d) */
int lmrl_gelu_fwd(const float *x_d, float *y_d, size_t n, void *stream);
int lmrl_gelu_bwd(const float *dy_d, const float *x_d, float *dx_d, size_t n, void *stream);
/* bf16-matmul mode: dx written only as the bf16 dy operand [rows_dst][ldb] of the next linear backward (padding zero-filled) */
int lmrl_gelu_bwd_bf16(const float *dy_d, const float *x_d, int rows, int cols, void *dst_d, long ldb, int rows_dst, void *stream);
int lmrl_relu_fwd(const float *x_d, float *y_d, size_t n, void *stream);
int lmrl_relu_bwd(const float *dy_d, const float *x_d, float *dx_d, size_t n, void *stream);
/* out = a*x + b*y (y may be NULL).  Polyak: optax.incremental_update(new, old, s) = axpby(s, new, 1-s, old). */
int lmrl_axpby(float a, const float *x_d, float b, const float *y_d, float *out_d, size_t n, void *stream);
/* optax.adamw(b1, b2, eps, weight_decay) with bias correction at `step` (1-based) */
int lmrl_adamw(float *p_d, const float *g_d, float *m_d, float *v_d, size_t n, float lr, float b1, float b2, float eps, float weight_decay,
               int step, void *stream);
/* ---- causal self-attention of the train step without the [B*H][T][T] tensors (csrc/flash_attn_train.hip): online-softmax tile sweeps on the
 * matrix cores, head dim 64.  qkv_d [B*T][3*H*64] fp32 (HF GPT-2 c_attn output: q | k | v), key_mask_d [B][T] uint8 or NULL, att_d / datt_d
 * [B*T][H*64]; lse_d [lmrl_flash_attn_lse_bytes] is written by fwd and read by bwd; ws_d [lmrl_flash_attn_ws_bytes] is scratch (re-staged by
 * each call).  bf16 = 0: exact fp32 operands (v_mfma_f32_16x16x4_f32); 1: bf16 operands, fp32 accumulation / softmax / outputs.
 * Same masking as lmrl_softmax_causal_fwd: keys c <= r with key_mask != 0; a query with no valid key gets a zero output row.
 * Restates the attention of FlaxGPT2 as differentiated by the reference's train steps (ppo/gpt2/interface.py:72-211). */
size_t lmrl_flash_attn_ws_bytes(int batch, int heads, int t, int bf16);
size_t lmrl_flash_attn_lse_bytes(int batch, int heads, int t);
int lmrl_flash_attn_fwd(const float *qkv_d, const uint8_t *key_mask_d, float *att_d, float *lse_d, void *ws_d, int batch, int heads, int t, int bf16,
                        void *stream);
/* forward that also writes the bf16 copy of att (row pitch ldb elements): the operand of the output projection in the bf16-matmul train mode */
int lmrl_flash_attn_fwd_staged(const float *qkv_d, const uint8_t *key_mask_d, float *att_d, float *lse_d, void *ws_d, void *att_bf16_d, long ldb, int batch,
                               int heads, int t, int bf16, void *stream);
/* bf16 workspace: where lmrl_gemm_bf16_qkv_heads writes (q matrix; k and v follow plane_elems apart); lmrl_flash_attn_finish_staging then adds the
 * transposed forms and zeroes the padded token rows, after which lmrl_flash_attn_fwd_staged(qkv_d = NULL, ...) and
 * lmrl_flash_attn_bwd_staged(qkv_d = NULL, ..., qkv_staged = 1) run on ws_d as if they had staged it themselves. */
int lmrl_flash_attn_stage_ptrs(void *ws_d, int batch, int heads, int t, void **q_heads_out, long *plane_elems_out);
int lmrl_flash_attn_finish_staging(void *ws_d, int batch, int heads, int t, void *stream);
int lmrl_flash_attn_bwd(const float *qkv_d, const uint8_t *key_mask_d, const float *att_d, const float *datt_d, const float *lse_d, float *dqkv_d,
                        void *ws_d, int batch, int heads, int t, int bf16,
                        int qkv_staged /* ws_d still holds what lmrl_flash_attn_fwd staged from these qkv: skip the re-staging */, void *stream);
/* bf16 kernels, d(qkv) written only as the bf16 dy operand [batch*t][ldb] of the c_attn backward products (bf16-matmul train mode) */
int lmrl_flash_attn_bwd_staged(const float *qkv_d, const uint8_t *key_mask_d, const float *att_d, const float *datt_d, const float *lse_d,
                               void *dqkv_bf16_d, long ldb, void *ws_d, int batch, int heads, int t, int qkv_staged, void *stream);
/* ... with D = rowsum(dO o O) from the bf16 attention output of lmrl_flash_attn_fwd_staged (att_bf16_d [batch * t][ld_att]); the forward may then be
 * called with att_d = NULL: no fp32 copy of the attention output exists in the bf16-matmul train mode. */
int lmrl_flash_attn_bwd_staged_attb(const float *qkv_d, const uint8_t *key_mask_d, const void *att_bf16_d, long ld_att, const float *datt_d, const float *lse_d,
                                    void *dqkv_bf16_d, long ldb, void *ws_d, int batch, int heads, int t, int qkv_staged, void *stream);
/* ---- bf16-MFMA matmul mode of the train step (csrc/train_bf16.hip): the reference's optional `bf16_activations`
 * (train_ilql_gpt2.py:193; model dtype bf16, fp32 parameters).  Operands are staged as K-major bf16 matrices for lmrl_gemm_bf16.
 * lmrl_cast_bf16: dst [rows_dst][ld_dst] bf16 := round-to-nearest-even of src [rows][cols] fp32 (transpose = 0) or of its transpose
 * (transpose = 1: dst[c][r] = src[r][c]); everything outside the source extent is zero-filled (K padding to multiples of 64). */
int lmrl_cast_bf16(const float *src_d, long ld_src, int rows, int cols, void *dst_d, long ld_dst, int rows_dst, int transpose, void *stream);
/* The bf16 staging of EVERY weight matrix of a parameter arena in one launch (per train step the fp32 masters move: the natural copy [rows][ld_nat]
 * feeds the dX products, the transposed copy [cols][ld_t] the forward products).  Offsets in elements: src_off into src_d (fp32, dense
 * [rows][cols]), nat_off / t_off into dst_d (bf16; < 0: that copy is not wanted); tile0 = number of 64 x 64 tiles of the segments before this one,
 * total_tiles = their sum over all segments.  Only source extents are written — zero-fill dst_d once. */
typedef struct {
    long src_off, nat_off, t_off;
    int rows, cols, ld_nat, ld_t, tile0, pad_;
} lmrl_cast_seg;
int lmrl_cast_bf16_segments(const float *src_d, const lmrl_cast_seg *segs_d, int nseg, int total_tiles, void *dst_d, void *stream);
/* "bf16 x 3" operand of an fp32 matrix: dst [rows][ld_dst >= 3 cols] bf16 := [hi(x) | lo(x) | hi(x)], hi = bf16(x), lo = bf16(x - hi).  Against
 * weight rows [hi(w) | hi(w) | lo(w)], lmrl_gemm_bf16 with K' = 3 cols accumulates hi.hi + lo.hi + hi.lo in fp32: ~16 mantissa bits per product at
 * 3x the bf16 MFMA cost (the f32-input MFMA costs 16x) — the matmul mode "bf16x3" of the fp32 rollout engine (GPT2EngineF32). */
int lmrl_split3_bf16(const float *src_d, long ld_src, int rows, int cols, void *dst_d, long ld_dst, void *stream);
/* transposed cast (as lmrl_cast_bf16 with transpose = 1) that also produces colsum_d[c] (=|+=) sum_r src[r][c] — the bias gradient of a
 * Dense layer falls out of staging dy^T for the dW product instead of a second pass over dy.  ws_d: lmrl_cast_bf16_t_colsum_ws_bytes. */
size_t lmrl_cast_bf16_t_colsum_ws_bytes(int rows, int rows_dst);
int lmrl_cast_bf16_t_colsum(const float *src_d, long ld_src, int rows, int cols, void *dst_d, long ld_dst, int rows_dst, float *colsum_d,
                            int accumulate, float *ws_d, void *stream);
/* bf16-matmul train mode, operands staged by their producer: dlogits of the CE / gather losses (lmrl_ce_bwd's formula) written directly as
 * the bf16 A operand [rows_dst][ld_dst] of the head's backward products (no fp32 dlogits, no cast pass), and the K-major transposed copy
 * (+ optional column sums = the bias gradient, from the bf16 values) made from such a bf16 operand. */
int lmrl_ce_bwd_bf16(const float *logits_d, int ld, int vocab, const float *lse_d, const int32_t *targets_d, const float *coef_ce_d,
                     const float *coef_gather_d, int rows, void *dst_d, long ld_dst, int rows_dst, void *stream);
int lmrl_transpose_bf16_colsum(const void *src_d, long ld_src, int rows, int cols, void *dst_d, long ld_dst, int rows_dst, float *colsum_d,
                               int accumulate, float *ws_d, void *stream);
/* dst[j][i] = beta*dst[j][i] + src[i][j]   (src [n][k] -> dst [k][n]): gradients produced in transposed form */
int lmrl_transpose_add_f32(const float *src_d, long ld_src, float *dst_d, long ld_dst, int n, int k, float beta, void *stream);
/* out[r] = sum_j a[r][j]*w[j][idx[r]] + bias[idx[r]]: the one column of a Dense layer that `take_along_axis(logits, token)` reads —
 * the ILQL target Q heads are only ever evaluated at the taken token (ilql/base_interface.py:57-66), exact fp32 */
int lmrl_gather_dot_f32(const float *a_d, long lda, const float *w_d, long ldw, const float *bias_d, const int32_t *idx_d, float *out_d, int rows,
                        int k, int n, void *stream);
/* lmrl_adamw over a whole parameter arena in one launch: seg_end_d[nseg] (exclusive element ends, ascending, last = n) and seg_wd_d[nseg] give
 * the weight-decay coefficient of each tensor (the optax mask: 0 for biases / LayerNorm).  Element for element the arithmetic of lmrl_adamw. */
int lmrl_adamw_segments(float *p_d, const float *g_d, float *m_d, float *v_d, long n, const long *seg_end_d, const float *seg_wd_d, int nseg, float lr,
                        float b1, float b2, float eps, int step, void *stream);
/* the same update and, in the same sweep, the Polyak target update of the reference's step (optax.incremental_update(new_params, target, alpha) right
 * after apply_gradients, ilql/gpt2/interface.py:327-347): target_d (nullable; same arena layout) = alpha * p_new + one_minus_alpha * target_d.
 * 16-byte accesses; arenas 16-byte aligned. */
int lmrl_adamw_segments_polyak(float *p_d, const float *g_d, float *m_d, float *v_d, long n, const long *seg_end_d, const float *seg_wd_d, int nseg, float lr,
                               float b1, float b2, float eps, int step, float *target_d, float alpha, float one_minus_alpha, void *stream);
/* P = causal (+ key padding mask [batch][t] uint8) softmax of S [batch*heads][t][t]; in place allowed */
int lmrl_softmax_causal_fwd(const float *s_d, const uint8_t *key_mask_d, float *p_d, int batch, int heads, int t, void *stream);
int lmrl_softmax_bwd(const float *p_d, float *dp_d, long rows, int t, void *stream);
/* per row: lse = logsumexp(logits[:vocab]), target logit, logprob = target - lse
 * (= -optax.softmax_cross_entropy_with_integer_labels; PPOInference.token_logprobs_from_logits, ppo/base_interface.py:396-403) */
int lmrl_lse_gather(const float *logits_d, int ld, int vocab, const int32_t *targets_d, float *logprob_d, float *lse_d,
                    float *target_logit_d, int rows, void *stream);
/* logits := coef_ce[r]*(softmax - onehot(t)) + coef_gather[r]*onehot(t)  (CE backward + take_along_axis backward) */
int lmrl_ce_bwd(float *logits_d, int ld, int vocab, const float *lse_d, const int32_t *targets_d, const float *coef_ce_d,
                const float *coef_gather_d, int rows, void *stream);
/* n = sum(should_take_action * attention_mask) as a device double */
int lmrl_mask_sum(const uint8_t *sta_d, const float *attn_d, size_t n, double *out_d, void *stream);
/* ppo_loss_fn forward+backward (ppo/base_interface.py:72-142). partials_d: [lmrl_ppo_loss_blocks(n)][lmrl_ppo_loss_nstats()]
 * doubles (layout in csrc/losses.hip), d_logprobs_d / d_values_d: gradient of the scalar loss. */
int lmrl_ppo_loss_blocks(size_t n);
int lmrl_ppo_loss_nstats(void);
int lmrl_ppo_loss(const float *attn_d, const float *logprobs_d, const float *values_d, const uint8_t *sta_d, const float *old_logprobs_d,
                  const float *old_values_d, const float *old_adv_d, const float *old_ret_d, size_t n, float cliprange_value,
                  float cliprange, float value_loss_coef, const double *n_mask_d, double *partials_d, float *d_logprobs_d,
                  float *d_values_d, void *stream);
/* ilql_loss forward+backward (ilql/base_interface.py:29-119) on gathered q1/q2/v/target-q [b][t1], v_final [b], per-token
 * CQL cross-entropies ce1/ce2 [b][t1]. partials_d: [b][lmrl_ilql_loss_nstats()] doubles. Outputs dq1,dq2,dv,coef_ce [b][t1]. */
int lmrl_ilql_loss_nstats(void);
int lmrl_ilql_loss(const float *q1_d, const float *q2_d, const float *v_d, const float *v_final_d, const float *tq1_d, const float *tq2_d,
                   const float *ce1_d, const float *ce2_d, const float *attn_d, const uint8_t *sta_d, const float *rewards_d, int b, int t1,
                   float gamma, float tau, float cql_weight, const double *n_mask_d, double *partials_d, float *dq1_d, float *dq2_d,
                   float *dv_d, float *coef_ce_d, void *stream);
/* mc_loss forward+backward (mc_returns/base_interface.py:19-60) */
int lmrl_mc_loss_blocks(size_t n);
int lmrl_mc_loss_nstats(void);
int lmrl_mc_loss(const float *q_d, const float *ce_d, const float *attn_d, const uint8_t *sta_d, const float *returns_d, size_t n,
                 float cql_weight, const double *n_mask_d, double *partials_d, float *dq_d, float *coef_ce_d, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LMRL_AMD_H */
