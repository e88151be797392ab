// maze_tokens.hip — on-device token <-> game bookkeeping for lock-step Maze rollouts with one-item histories (last_k = 1).
//
// The reference's act()/step() cycle (LLM_RL/environment.py:180-206, ppo/gpt2/interface.py:519-546) renders the observation text of
// the current cell, tokenises it, generates, decodes the ids, applies `out_str_process` and looks the string up in the action dict
// (maze/env/env.py:161-184).  With last_k = 1 the prompt is a pure function of (goal, cell), so the host does the text work ONCE per
// (tokenizer, maze): a table of observation token ids and a table of per-token bytes.  Here each turn
//   * picks the env's prompt row from its state (and the prompt's precomputed K/V rows via lmrl_gpt2_kv_gather),
//   * decodes the generated ids through the byte table exactly as `decode(ids, skip_special_tokens=True)` concatenates token strings,
//   * applies `x.removesuffix('\n') + '\n'` and compares with 'move left\n' | 'move right\n' | 'move up\n' | 'move down\n',
// and nothing leaves HBM during an episode.  One thread per env.
#include "../../include/lmrl_amd.h"
#include "common.h"

struct lmrl_maze_tok_ctx {
    int32_t *obs_tok_d = nullptr, *obs_len_d = nullptr, *goal_slot_d = nullptr;
    uint8_t *tok_bytes_d = nullptr, *tok_blen_d = nullptr;
    int n_obs = 0, obs_cap = 0, rows = 0, cols = 0, vocab = 0, max_new = 0, max_turns = 0;
};

namespace lmrl {

constexpr int kTokBytes = 16;     // byte-table pitch
constexpr int kTextCap = 24;      // longest text that can still be an action ('move right\n' = 11 bytes); longer -> OTHER

__global__ void maze_begin_kernel(lmrl_maze_traj tr, int n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    tr.n_turns[e] = 0;
    tr.live[e] = 1;
    tr.ep_reward[e] = 0.f;
}

__global__ void maze_turn_kernel(lmrl_maze_traj tr, const int32_t *__restrict__ state, const int32_t *__restrict__ goal_slot, int rows, int cols,
                                 int max_turns, int n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int t = tr.n_turns[e];
    int idx = -1;
    if (tr.live[e] && t < max_turns) {
        const int r = state[e], c = state[n + e], gr = state[2 * n + e], gc = state[3 * n + e];
        const int slot = goal_slot[gr * cols + gc];
        if (slot >= 0) idx = (slot * rows + r) * cols + c;
        tr.pos[(size_t)e * max_turns + t] = (r << 16) | c;
    } else {
        tr.live[e] = 0;                       // record full: the episode is cut here (the host sizes T = max_steps + 1, so this is never hit)
    }
    tr.obs_idx[e] = idx;
    tr.out_len[e] = 0;
    tr.gen_active[e] = idx >= 0 ? 1 : 0;
    tr.stepping[e] = idx >= 0 ? 1 : 0;
}

__global__ void maze_prompt_kernel(lmrl_maze_traj tr, const int32_t *__restrict__ obs_tok, const int32_t *__restrict__ obs_len, int obs_cap, int j,
                                   int chunk, int32_t *__restrict__ chunk_tok, int32_t *__restrict__ chunk_cnt, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * chunk) return;
    const int e = i / chunk, k = i - e * chunk;
    const int idx = tr.obs_idx[e];
    const int len = idx >= 0 ? obs_len[idx] : 0;
    const int p = j * chunk + k;
    chunk_tok[i] = p < len ? obs_tok[(size_t)idx * obs_cap + p] : 0;
    if (k == 0) chunk_cnt[e] = max(0, min(chunk, len - j * chunk));
}

__global__ void maze_action_kernel(lmrl_maze_traj tr, const uint8_t *__restrict__ tok_bytes, const uint8_t *__restrict__ tok_blen, int vocab,
                                   int max_new, int max_turns, int n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    if (tr.obs_idx[e] < 0) { tr.act[e] = LMRL_MAZE_OTHER; return; }
    const int t = tr.n_turns[e], len = tr.out_len[e];
    char text[kTextCap];
    int tl = 0;
    bool other = false;
    for (int k = 0; k < len; k++) {
        const int tok = tr.out_tok[(size_t)e * max_new + k];
        tr.gen[((size_t)e * max_turns + t) * max_new + k] = tok;
        const int bl = (tok >= 0 && tok < vocab) ? tok_blen[tok] : 255;
        if (bl == 255 || tl + bl > kTextCap) { other = true; continue; }
        for (int q = 0; q < bl; q++) text[tl + q] = (char)tok_bytes[(size_t)tok * kTokBytes + q];
        tl += bl;
    }
    tr.gen_len[(size_t)e * max_turns + t] = len;
    int code = LMRL_MAZE_OTHER;
    if (!other) {
        if (tl > 0 && text[tl - 1] == '\n') tl--;                    // removesuffix('\n'); the forced '\n' is implied below
        const char *names[4] = {"move left", "move right", "move up", "move down"};
        const int nl[4] = {9, 10, 7, 9};
#pragma unroll
        for (int a = 0; a < 4; a++) {
            bool eq = tl == nl[a];
            for (int q = 0; eq && q < nl[a]; q++) eq = text[q] == names[a][q];
            if (eq) code = a;
        }
    }
    tr.act[e] = (uint8_t)code;
    tr.action[(size_t)e * max_turns + t] = (uint8_t)code;
}

__global__ void maze_result_kernel(lmrl_maze_traj tr, const float *__restrict__ reward, const uint8_t *__restrict__ done,
                                   const uint8_t *__restrict__ kind, int max_turns, int n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n || tr.obs_idx[e] < 0) return;
    const int t = tr.n_turns[e];
    tr.reward[(size_t)e * max_turns + t] = reward[e];
    tr.kind[(size_t)e * max_turns + t] = kind[e];
    tr.ep_reward[e] += reward[e];
    tr.n_turns[e] = t + 1;
    if (done[e]) tr.live[e] = 0;
}

// ---- the finished episodes as PPO records (round 5): one token trajectory per TRANSITION, chained per episode — what the Maze / chess online
// scripts hand to get_ppo_data_from_text_trajectory_chain (llm_rl_scripts/maze/ppo/train_ppo_online.py:444-465: TextTrajectory(post_action_history,
// reward = [0, r], done) linked in episode order) after TokenTrajectory.from_text_trajectory (LLM_RL/environment.py:359-370): tokens = the
// observation's ids ++ the action's ids, is_action, the step reward on the action's last token.  The action's ids are the GENERATED ids with
// special tokens dropped and one newline id appended when the decoded text does not end in a newline — the ids of `removesuffix('\n') + '\n'`
// whenever the tokenizer's encoding of the decoded action is the generated sequence (byte-level tokenizers always).
// One wave per env; trajectory rows are compacted over the valid turns (row = off[env] + turn).
__global__ __launch_bounds__(256) void maze_ppo_records_kernel(lmrl_maze_traj tr, const int32_t *__restrict__ state, const int32_t *__restrict__ goal_slot,
                                                               const int32_t *__restrict__ obs_tok, const int32_t *__restrict__ obs_len, int obs_cap,
                                                               const uint8_t *__restrict__ tok_bytes, const uint8_t *__restrict__ tok_blen, int vocab,
                                                               int rows, int cols, int max_new, int max_turns, int n, int pitch, const int32_t *__restrict__ off,
                                                               int newline_tok, int cap, int32_t *__restrict__ tokens, uint8_t *__restrict__ is_action,
                                                               float *__restrict__ reward, int32_t *__restrict__ n_tok, int32_t *__restrict__ chain,
                                                               int32_t *__restrict__ pos, uint8_t *__restrict__ last, uint8_t *__restrict__ done,
                                                               int32_t *__restrict__ chain_total) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + wave;
    if (e >= n) return;
    const int nt = tr.n_turns[e];
    const int gr = state[2 * pitch + e], gc = state[3 * pitch + e];       // (state rows have the env batch's pitch; n <= pitch envs are exported)
    const int slot = goal_slot[gr * cols + gc];
    int offset = 0;
    for (int t = 0; t < nt; t++) {
        const size_t row = (size_t)off[e] + t;
        const int pc = tr.pos[(size_t)e * max_turns + t];
        const int idx = slot >= 0 ? (slot * rows + (pc >> 16)) * cols + (pc & 0xFFFF) : -1;
        const int ol = idx >= 0 ? min(obs_len[idx], cap) : 0;
        for (int k = lane; k < ol; k += 64) {
            tokens[row * cap + k] = obs_tok[(size_t)idx * obs_cap + k];
            is_action[row * cap + k] = 0;
            reward[row * cap + k] = 0.f;
        }
        // the action: generated ids minus special tokens (byte length 0), in order (wave-uniform walk: at most max_new ids)
        const int gl = tr.gen_len[(size_t)e * max_turns + t];
        int na = 0, last_byte = -1;
        for (int k = 0; k < gl; k++) {
            const int tok = tr.gen[((size_t)e * max_turns + t) * max_new + k];
            const int bl = (tok >= 0 && tok < vocab) ? tok_blen[tok] : 255;
            if (bl == 0) continue;                                   // skip_special_tokens
            if (lane == 0 && ol + na < cap) tokens[row * cap + ol + na] = tok;
            last_byte = bl == 255 ? -1 : tok_bytes[(size_t)tok * kTokBytes + bl - 1];
            na++;
        }
        if (last_byte != '\n') {                                     // removesuffix('\n') + '\n' on a text without a trailing newline
            if (lane == 0 && ol + na < cap) tokens[row * cap + ol + na] = newline_tok;
            na++;
        }
        const int total = min(ol + na, cap);
        for (int k = ol + lane; k < total; k += 64) {
            is_action[row * cap + k] = 1;
            reward[row * cap + k] = k == total - 1 ? tr.reward[(size_t)e * max_turns + t] : 0.f;
        }
        if (lane == 0) {
            n_tok[row] = total;
            chain[row] = e;
            pos[row] = offset;
            last[row] = t == nt - 1;
        }
        offset += total > 0 ? total - 1 : 0;
    }
    if (lane == 0) {
        const int k = nt > 0 ? tr.kind[(size_t)e * max_turns + nt - 1] : 0;
        done[e] = (k == 1 || k == 2) ? 1 : 0;                        // the episode's last transition ended in Failure / Success (maze/env/env.py:164-184)
        chain_total[e] = offset;
    }
}

}  // namespace lmrl

using namespace lmrl;

extern "C" {

lmrl_maze_tok_ctx *lmrl_maze_tok_create(const int32_t *obs_tok, const int32_t *obs_len, int n_obs, int obs_cap, const int32_t *goal_slot, int rows,
                                        int cols, const uint8_t *tok_bytes, const uint8_t *tok_blen, int vocab, int max_new_tokens, int max_turns) {
    if (!obs_tok || !obs_len || !goal_slot || !tok_bytes || !tok_blen || n_obs <= 0 || obs_cap <= 0 || rows <= 0 || cols <= 0 || vocab <= 0 ||
        max_new_tokens <= 0 || max_turns <= 0) {
        set_error("lmrl_maze_tok_create: bad argument");
        return nullptr;
    }
    lmrl_maze_tok_ctx *c = new lmrl_maze_tok_ctx();
    c->n_obs = n_obs; c->obs_cap = obs_cap; c->rows = rows; c->cols = cols; c->vocab = vocab; c->max_new = max_new_tokens; c->max_turns = max_turns;
    auto up = [](auto **dst, const void *src, size_t bytes) {
        return hipMalloc((void **)dst, bytes) == hipSuccess && hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
    };
    const bool ok = up(&c->obs_tok_d, obs_tok, sizeof(int32_t) * (size_t)n_obs * obs_cap) && up(&c->obs_len_d, obs_len, sizeof(int32_t) * n_obs) &&
                    up(&c->goal_slot_d, goal_slot, sizeof(int32_t) * (size_t)rows * cols) &&
                    up(&c->tok_bytes_d, tok_bytes, (size_t)vocab * kTokBytes) && up(&c->tok_blen_d, tok_blen, (size_t)vocab);
    if (!ok) {
        set_error("lmrl_maze_tok_create: device allocation/copy failed (is a GPU visible?)");
        lmrl_maze_tok_destroy(c);
        return nullptr;
    }
    return c;
}

void lmrl_maze_tok_destroy(lmrl_maze_tok_ctx *c) {
    if (!c) return;
    for (void *p : {(void *)c->obs_tok_d, (void *)c->obs_len_d, (void *)c->goal_slot_d, (void *)c->tok_bytes_d, (void *)c->tok_blen_d})
        if (p) (void)hipFree(p);
    delete c;
}

int lmrl_maze_tok_begin(lmrl_maze_tok_ctx *c, const lmrl_maze_traj *tr, int n, void *stream) {
    LMRL_REQUIRE(c && tr && n > 0, "lmrl_maze_tok_begin: bad argument");
    hipLaunchKernelGGL(maze_begin_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), *tr, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_maze_tok_turn(lmrl_maze_tok_ctx *c, const lmrl_maze_traj *tr, const void *state_d, int n, void *stream) {
    LMRL_REQUIRE(c && tr && state_d && n > 0, "lmrl_maze_tok_turn: bad argument");
    hipLaunchKernelGGL(maze_turn_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), *tr, (const int32_t *)state_d, c->goal_slot_d,
                       c->rows, c->cols, c->max_turns, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_maze_tok_prompt(lmrl_maze_tok_ctx *c, const lmrl_maze_traj *tr, int j, int chunk, int32_t *chunk_tok_d, int32_t *chunk_cnt_d, int n,
                         void *stream) {
    LMRL_REQUIRE(c && tr && chunk_tok_d && chunk_cnt_d && j >= 0 && chunk > 0 && n > 0, "lmrl_maze_tok_prompt: bad argument");
    hipLaunchKernelGGL(maze_prompt_kernel, dim3(ceil_div(n * chunk, 256)), dim3(256), 0, as_stream(stream), *tr, c->obs_tok_d, c->obs_len_d,
                       c->obs_cap, j, chunk, chunk_tok_d, chunk_cnt_d, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_maze_tok_action(lmrl_maze_tok_ctx *c, const lmrl_maze_traj *tr, int n, void *stream) {
    LMRL_REQUIRE(c && tr && n > 0, "lmrl_maze_tok_action: bad argument");
    hipLaunchKernelGGL(maze_action_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), *tr, c->tok_bytes_d, c->tok_blen_d, c->vocab,
                       c->max_new, c->max_turns, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_maze_tok_ppo_records(lmrl_maze_tok_ctx *c, const lmrl_maze_traj *tr, const void *state_d, int n, int n_envs, const int32_t *off_d, int newline_tok, int cap,
                              int32_t *tokens_d, uint8_t *is_action_d, float *reward_d, int32_t *n_tok_d, int32_t *chain_d, int32_t *pos_d, uint8_t *last_d,
                              uint8_t *done_d, int32_t *chain_total_d, void *stream) {
    LMRL_REQUIRE(c && tr && state_d && n > 0 && n <= n_envs && off_d && cap >= 2 && tokens_d && is_action_d && reward_d && n_tok_d && chain_d && pos_d && last_d && done_d &&
                     chain_total_d, "lmrl_maze_tok_ppo_records: bad argument");
    hipLaunchKernelGGL(maze_ppo_records_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, as_stream(stream), *tr, (const int32_t *)state_d, c->goal_slot_d,
                       c->obs_tok_d, c->obs_len_d, c->obs_cap, c->tok_bytes_d, c->tok_blen_d, c->vocab, c->rows, c->cols, c->max_new, c->max_turns, n, n_envs, off_d,
                       newline_tok, cap, tokens_d, is_action_d, reward_d, n_tok_d, chain_d, pos_d, last_d, done_d, chain_total_d);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_maze_tok_result(lmrl_maze_tok_ctx *c, const lmrl_maze_traj *tr, const float *reward_d, const uint8_t *done_d, const uint8_t *kind_d,
                         int n, void *stream) {
    LMRL_REQUIRE(c && tr && reward_d && done_d && kind_d && n > 0, "lmrl_maze_tok_result: bad argument");
    hipLaunchKernelGGL(maze_result_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), *tr, reward_d, done_d, kind_d, c->max_turns, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

}  // extern "C"
