// maze_tokens.hip — on-device token <-> game bookkeeping for lock-step Maze rollouts: one-item histories (last_k = 1: prompt table + prefix cache)
// and, round 6, item windows (last_k > 1: a persistent per-env KV cache that grows with the history, lmrl_maze_hist_*).
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
    int32_t *act_tok_d = nullptr;      // [4][act_cap] the tokenizer's encoding of the four action strings ('move left\n' ...), act_len in slot act_cap - 1 ... see lmrl_maze_tok_set_actions
    int act_cap = 0;
    // the same texts encoded behind a joining space (' ' + text): the items of a window after the first one in the partially observed PPO script's
    // state text (" ".join(...)), see lmrl_maze_tok_set_spaced
    int32_t *obs_sp_tok_d = nullptr, *obs_sp_len_d = nullptr, *act_sp_tok_d = nullptr;
    int obs_sp_cap = 0, act_sp_cap = 0;
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
                                                               const int32_t *__restrict__ act_tok, int act_cap, int byte_ids,
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
        // the action.  A legal one (its post-processed text IS a key of the action dict): the tokenizer's encoding of that key when the host supplied it
        // (lmrl_maze_tok_set_actions) — the reference's re-tokenisation for ANY tokenizer.  Else: generated ids minus special tokens (byte length 0), in
        // order (wave-uniform walk: at most max_new ids) — with byte_ids (ids = the text's UTF-8 bytes) the bytes of every decoded token instead
        const int gl = tr.gen_len[(size_t)e * max_turns + t];
        const int code = tr.action[(size_t)e * max_turns + t];
        int na = 0, last_byte = -1;
        if (act_tok && code < 4) {
            na = act_tok[code * act_cap + act_cap - 1];
            for (int k = lane; k < na; k += 64)
                if (ol + k < cap) tokens[row * cap + ol + k] = act_tok[code * act_cap + k];
            last_byte = '\n';
        } else {
            for (int k = 0; k < gl; k++) {
                const int tok = tr.gen[((size_t)e * max_turns + t) * max_new + k];
                const int bl = (tok >= 0 && tok < vocab) ? tok_blen[tok] : 255;
                if (bl == 0) continue;                               // skip_special_tokens
                if (byte_ids && bl != 255) {
                    if (lane == 0)
                        for (int b = 0; b < bl; b++)
                            if (ol + na + b < cap) tokens[row * cap + ol + na + b] = tok_bytes[(size_t)tok * kTokBytes + b];
                    last_byte = tok_bytes[(size_t)tok * kTokBytes + bl - 1];
                    na += bl;
                    continue;
                }
                if (lane == 0 && ol + na < cap) tokens[row * cap + ol + na] = tok;
                last_byte = bl == 255 ? -1 : tok_bytes[(size_t)tok * kTokBytes + bl - 1];
                na++;
            }
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

// ---- the same export for item windows (last_k > 1), in the form the partially observed online script builds (llm_rl_scripts/maze/ppo/
// partially_observed_ppo_online.py:372-398): per transition ONE non-action text — the window's item texts joined by single spaces — followed by the
// action text, reward on the action's last token.  The window of turn t is rebuilt from the record: (history + [action] + [observation])[-last_k:] per
// legal step, (observation,) alone after an action string outside the action dict (maze/env/env.py:179-184).  Token ids: the first item's own encoding,
// every later item's encoding behind the joining space (host tables: lmrl_maze_tok_set_spaced — equal to tokenizer.encode(joined text) for tokenizers
// whose encoding splits between an item's trailing newline and the space: checked on the host); the action: the tokenizer's encoding of a legal
// action string (its text IS the dict key), else the generated ids as in maze_ppo_records_kernel.  One wave per env; a window item is an observation
// table row (>= 0) or -(action code + 1), kept in a ring of last_k <= 64 entries.  count_only: lengths only (the host sizes `cap` by the longest row).
__global__ __launch_bounds__(256) void maze_ppo_records_hist_kernel(lmrl_maze_traj tr, const int32_t *__restrict__ state, const int32_t *__restrict__ goal_slot,
                                                                    const int32_t *__restrict__ obs_tok, const int32_t *__restrict__ obs_len, int obs_cap,
                                                                    const int32_t *__restrict__ obs_sp_tok, const int32_t *__restrict__ obs_sp_len, int obs_sp_cap,
                                                                    const int32_t *__restrict__ act_tok, int act_cap, const int32_t *__restrict__ act_sp_tok, int act_sp_cap,
                                                                    const uint8_t *__restrict__ tok_bytes, const uint8_t *__restrict__ tok_blen, int vocab,
                                                                    int rows, int cols, int max_new, int max_turns, int n, int pitch, const int32_t *__restrict__ off,
                                                                    int last_k, int newline_tok, int cap, int count_only, int byte_ids, int32_t *__restrict__ tokens,
                                                                    uint8_t *__restrict__ is_action, float *__restrict__ reward, int32_t *__restrict__ n_tok,
                                                                    int32_t *__restrict__ chain, int32_t *__restrict__ pos, uint8_t *__restrict__ last,
                                                                    uint8_t *__restrict__ done, int32_t *__restrict__ chain_total) {
    __shared__ int32_t ring_s[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + wave;
    if (e >= n) return;
    int32_t *ring = ring_s[wave];
    const int nt = tr.n_turns[e];
    const int gr = state[2 * pitch + e], gc = state[3 * pitch + e];
    const int slot = goal_slot[gr * cols + gc];
    auto obs_row = [&](int t) {
        const int pc = tr.pos[(size_t)e * max_turns + t];
        return slot >= 0 ? (slot * rows + (pc >> 16)) * cols + (pc & 0xFFFF) : -1;
    };
    int start = 0, count = 0, offset = 0;
    auto push = [&](int item) {                                      // wave-uniform; lane 0 writes, the wave reads after the fence below
        if (count == last_k) start = (start + 1) & 63; else count++;
        if (lane == 0) ring[(start + count - 1) & 63] = item;
    };
    for (int t = 0; t < nt; t++) {
        const size_t row = (size_t)off[e] + t;
        if (t == 0) push(obs_row(0));
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int len = 0;
        for (int j = 0; j < count; j++) {
            const int item = ring[(start + j) & 63];
            const int32_t *src = nullptr;
            int l = 0;
            if (item >= 0) {
                src = j == 0 ? obs_tok + (size_t)item * obs_cap : obs_sp_tok + (size_t)item * obs_sp_cap;
                l = j == 0 ? obs_len[item] : obs_sp_len[item];
            } else if (item > -6) {
                const int code = -item - 1;
                src = j == 0 ? act_tok + code * act_cap : act_sp_tok + code * act_sp_cap;
                l = j == 0 ? act_tok[code * act_cap + act_cap - 1] : act_sp_tok[code * act_sp_cap + act_sp_cap - 1];
            }
            if (!count_only)
                for (int k = lane; k < l; k += 64)
                    if (len + k < cap) {
                        tokens[row * cap + len + k] = src[k];
                        is_action[row * cap + len + k] = 0;
                        reward[row * cap + len + k] = 0.f;
                    }
            len += l;
        }
        const int code = tr.action[(size_t)e * max_turns + t];
        int na = 0;
        if (code < 4) {                                              // a legal action: its text is the dict key
            na = act_tok[code * act_cap + act_cap - 1];
            if (!count_only)
                for (int k = lane; k < na; k += 64)
                    if (len + k < cap) tokens[row * cap + len + k] = act_tok[code * act_cap + k];
        } else {
            const int gl = tr.gen_len[(size_t)e * max_turns + t];
            int last_byte = -1;
            for (int k = 0; k < gl; k++) {
                const int tok = tr.gen[((size_t)e * max_turns + t) * max_new + k];
                const int bl = (tok >= 0 && tok < vocab) ? tok_blen[tok] : 255;
                if (bl == 0) continue;                               // skip_special_tokens
                if (byte_ids && bl != 255) {                         // a tokenizer whose ids ARE the text's UTF-8 bytes: encode(decoded text), byte by byte
                    if (!count_only && lane == 0)
                        for (int b = 0; b < bl; b++)
                            if (len + na + b < cap) tokens[row * cap + len + na + b] = tok_bytes[(size_t)tok * kTokBytes + b];
                    last_byte = tok_bytes[(size_t)tok * kTokBytes + bl - 1];
                    na += bl;
                    continue;
                }
                if (!count_only && lane == 0 && len + na < cap) tokens[row * cap + len + na] = tok;
                last_byte = bl == 255 ? -1 : tok_bytes[(size_t)tok * kTokBytes + bl - 1];
                na++;
            }
            if (last_byte != '\n') {
                if (!count_only && lane == 0 && len + na < cap) tokens[row * cap + len + na] = newline_tok;
                na++;
            }
        }
        const int full = len + na, total = count_only ? full : min(full, cap);
        if (!count_only)
            for (int k = len + lane; k < total; k += 64) {
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
        const int kind = tr.kind[(size_t)e * max_turns + t];
        if (t + 1 < nt && kind != LMRL_MAZE_KIND_FAILURE && kind != LMRL_MAZE_KIND_SUCCESS) {
            __builtin_amdgcn_wave_barrier();                         // (every lane is done reading the ring)
            if (kind == LMRL_MAZE_KIND_OBS_ONLY) { start = 0; count = 0; }
            else push(-(code + 1));
            push(obs_row(t + 1));
        }
    }
    if (lane == 0) {
        const int k = nt > 0 ? tr.kind[(size_t)e * max_turns + nt - 1] : 0;
        done[e] = (k == 1 || k == 2) ? 1 : 0;
        chain_total[e] = offset;
    }
}

// ---- histories longer than one item (round 6): `MazeEnv.step` returns (history + [action] + [observation])[-last_k:] (maze/env/env.py:182-184) and the
// policy's prompt is the text of that window, left-truncated to max_input_length tokens (ppo/gpt2/interface.py:519-524; partially_observed_bc.py:241
// runs last_k = 40).  While the window only GROWS — fewer than last_k items and max_input_length tokens — prompt t + 1 = prompt t ++ action ++ new
// observation token for token (concatenative tokenizers: byte level), i.e. the K/V rows of prompt t stay valid at their positions: per turn only the
// action's unforwarded tail and the new observation go through the model ("append" turns, as the Wordle loop).  Once the window slides, every position
// shifts and GPT-2's learned absolute position embeddings enter every layer's K / V: the window is forwarded again from position 0 ("re-prefill"
// turns — exact, and priced in DESIGN.md).  The host picks the kind of turn from static bounds; a turn scheduled as "append" that would have needed a
// re-prefill of more tokens than its chunk budget raises flag bit 0.  An action string outside the action dict makes the env return (observation,)
// alone (env.py:179-180): that env's window restarts — one observation long, so it fits an append turn's budget.  Per env: the episode's token
// history and the token offset of every item.  One thread per env.
struct MazeHist {               // lmrl_maze_hist, device pointers
    int32_t *hist, *item_off, *n_items, *feed_start, *feed_len, *cache_len, *base, *prompt_len, *win_floor, *flags;
    int32_t hcap, max_items;
};

__global__ void maze_hist_begin_kernel(MazeHist h, int32_t *__restrict__ len0, int32_t *__restrict__ len1, int n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e == 0) h.flags[0] = 0;
    if (e >= n) return;
    h.n_items[e] = 0;
    h.item_off[(size_t)e * (h.max_items + 1)] = 0;
    h.cache_len[e] = 0; h.base[e] = 0; h.feed_start[e] = 0; h.feed_len[e] = 0; h.prompt_len[e] = 0; h.win_floor[e] = 0;
    len0[e] = 0;
    if (len1) len1[e] = 0;
}

// turn start (after maze_turn_kernel): the current observation becomes the window's newest item; which history tokens go through the model this turn
__global__ void maze_hist_observe_kernel(lmrl_maze_traj tr, MazeHist h, const int32_t *__restrict__ obs_tok, const int32_t *__restrict__ obs_len, int obs_cap,
                                         int last_k, int max_input, int reprefill, int feed_budget, int max_turns, int32_t *__restrict__ len0,
                                         int32_t *__restrict__ len1, int n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int idx = tr.obs_idx[e];
    if (idx < 0) { h.feed_len[e] = 0; return; }
    int32_t *off = h.item_off + (size_t)e * (h.max_items + 1);
    int32_t *hist = h.hist + (size_t)e * h.hcap;
    int ni = h.n_items[e], total = off[ni];
    const int ol = obs_len[idx];
    if (ni >= h.max_items || total + ol > h.hcap) { atomicOr(h.flags, 2); h.feed_len[e] = 0; tr.gen_active[e] = 0; return; }
    // an action string outside the action dict makes the env return (observation,) alone (env.py:179-180): the window restarts at this observation
    const int t = tr.n_turns[e];
    if (t > 0 && tr.kind[(size_t)e * max_turns + t - 1] == LMRL_MAZE_KIND_OBS_ONLY) h.win_floor[e] = ni;
    for (int k = 0; k < ol; k++) hist[total + k] = obs_tok[(size_t)idx * obs_cap + k];
    total += ol; ni += 1;
    off[ni] = total;
    h.n_items[e] = ni;
    const int floor_item = h.win_floor[e];
    const int first_item = max(floor_item, ni > last_k ? ni - last_k : 0);      // [-last_k:] of the item list since the last restart
    int start = off[first_item];
    if (total - start > max_input) start = total - max_input;                   // Truncation.LEFT on the token level
    h.prompt_len[e] = total - start;
    int base = h.base[e], cached = h.cache_len[e];
    if (reprefill || start != base) {
        // this env's window moved (slide, truncation, restart): its rows are forwarded again from position 0.  In a turn scheduled as "append" that
        // is fine as long as the window fits the turn's chunk budget (a restarted window is one observation long); otherwise the schedule was wrong
        if (!reprefill && total - start > feed_budget) atomicOr(h.flags, 1);
        base = start; cached = 0;
    }
    h.base[e] = base;
    h.feed_start[e] = base + cached;
    h.feed_len[e] = total - (base + cached);
    h.cache_len[e] = total - base;                                              // after this turn's chunk forwards
    len0[e] = cached;
    if (len1) len1[e] = cached;
}

__global__ void maze_hist_chunk_kernel(MazeHist h, int j, int chunk, int32_t *__restrict__ chunk_tok, int32_t *__restrict__ chunk_cnt, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * chunk) return;
    const int e = i / chunk, k = i - e * chunk;
    const int len = h.feed_len[e], p = j * chunk + k;
    chunk_tok[i] = p < len ? h.hist[(size_t)e * h.hcap + h.feed_start[e] + p] : 0;
    if (k == 0) chunk_cnt[e] = max(0, min(chunk, len - j * chunk));
}

// after maze_action_kernel: the action joins the window.  A LEGAL action's ids are the tokenizer's own encoding of its string (`act_tok`: the next
// prompt is tokenizer.encode(text of the window), whatever ids the policy generated to spell 'move left\n'); an illegal string never reaches a
// later prompt (the env answers with (observation,) alone, env.py:179-180) — its generated ids are kept only as a place holder item.  The K/V rows of
// the generated ids that were forwarded (all but the last one) stay valid as far as they ARE the action's ids: the cache is cut back to prompt +
// that common prefix.  act_tok [4][act_cap]: ids of action a, its length in the last slot.
__global__ void maze_hist_action_kernel(lmrl_maze_traj tr, MazeHist h, const uint8_t *__restrict__ tok_blen, int vocab, int max_new,
                                        const int32_t *__restrict__ act_tok, int act_cap, int32_t *__restrict__ len0, int32_t *__restrict__ len1, int n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n || tr.obs_idx[e] < 0) return;
    int32_t *off = h.item_off + (size_t)e * (h.max_items + 1);
    int32_t *hist = h.hist + (size_t)e * h.hcap;
    int ni = h.n_items[e], total = off[ni];
    const int gl = tr.out_len[e];
    const int code = tr.act[e];
    const int al = code < 4 ? act_tok[code * act_cap + act_cap - 1] : gl;
    if (ni >= h.max_items || total + al > h.hcap) { atomicOr(h.flags, 2); return; }
    const int fwd = gl > 0 ? gl - 1 : 0;                                        // generated ids that went through the model (lmrl_gen_accept)
    int lcp = 0;
    if (code < 4) {
        bool same = true;
        for (int k = 0; k < al; k++) {
            const int tok = act_tok[code * act_cap + k];
            hist[total + k] = tok;
            same = same && k < fwd && tr.out_tok[(size_t)e * max_new + k] == tok;
            if (same) lcp = k + 1;
        }
    } else {
        for (int k = 0; k < gl; k++) hist[total + k] = tr.out_tok[(size_t)e * max_new + k];
    }
    off[ni + 1] = total + al;
    h.n_items[e] = ni + 1;
    const int cached = (total - h.base[e]) + lcp;                                // prompt rows + the generated rows that equal the action's ids
    h.cache_len[e] = cached;
    len0[e] = cached;
    if (len1) len1[e] = cached;
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
    for (void *p : {(void *)c->obs_tok_d, (void *)c->obs_len_d, (void *)c->goal_slot_d, (void *)c->tok_bytes_d, (void *)c->tok_blen_d, (void *)c->act_tok_d,
                    (void *)c->obs_sp_tok_d, (void *)c->obs_sp_len_d, (void *)c->act_sp_tok_d})
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

static MazeHist hist_of(const lmrl_maze_hist *h) {
    return MazeHist{h->hist, h->item_off, h->n_items, h->feed_start, h->feed_len, h->cache_len, h->base, h->prompt_len, h->win_floor, h->flags, h->hcap, h->max_items};
}
static bool hist_ok(const lmrl_maze_hist *h) {
    return h && h->hist && h->item_off && h->n_items && h->feed_start && h->feed_len && h->cache_len && h->base && h->prompt_len && h->win_floor && h->flags &&
           h->hcap > 0 && h->max_items > 0;
}

int lmrl_maze_hist_begin(const lmrl_maze_hist *h, int32_t *len0_d, int32_t *len1_d, int n, void *stream) {
    LMRL_REQUIRE(hist_ok(h) && len0_d && n > 0, "lmrl_maze_hist_begin: bad argument");
    hipLaunchKernelGGL(maze_hist_begin_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), hist_of(h), len0_d, len1_d, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_maze_hist_observe(lmrl_maze_tok_ctx *c, const lmrl_maze_traj *tr, const lmrl_maze_hist *h, int last_k, int max_input_length, int reprefill,
                           int feed_budget, int32_t *len0_d, int32_t *len1_d, int n, void *stream) {
    LMRL_REQUIRE(c && tr && hist_ok(h) && last_k >= 1 && max_input_length > 0 && feed_budget > 0 && len0_d && n > 0, "lmrl_maze_hist_observe: bad argument");
    hipLaunchKernelGGL(maze_hist_observe_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), *tr, hist_of(h), c->obs_tok_d, c->obs_len_d, c->obs_cap,
                       last_k, max_input_length, reprefill, feed_budget, c->max_turns, len0_d, len1_d, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_maze_hist_chunk(const lmrl_maze_hist *h, int j, int chunk, int32_t *chunk_tok_d, int32_t *chunk_cnt_d, int n, void *stream) {
    LMRL_REQUIRE(hist_ok(h) && chunk_tok_d && chunk_cnt_d && j >= 0 && chunk > 0 && n > 0, "lmrl_maze_hist_chunk: bad argument");
    hipLaunchKernelGGL(maze_hist_chunk_kernel, dim3(ceil_div(n * chunk, 256)), dim3(256), 0, as_stream(stream), hist_of(h), j, chunk, chunk_tok_d, chunk_cnt_d, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_maze_tok_set_actions(lmrl_maze_tok_ctx *c, const int32_t *act_tok, int act_cap) {
    LMRL_REQUIRE(c && act_tok && act_cap >= 2, "lmrl_maze_tok_set_actions: bad argument");
    for (int a = 0; a < 4; a++)
        LMRL_REQUIRE(act_tok[a * act_cap + act_cap - 1] > 0 && act_tok[a * act_cap + act_cap - 1] < act_cap, "lmrl_maze_tok_set_actions: action length outside (0, act_cap)");
    if (c->act_tok_d) (void)hipFree(c->act_tok_d);
    c->act_tok_d = nullptr;
    LMRL_CHECK_HIP(hipMalloc((void **)&c->act_tok_d, sizeof(int32_t) * 4 * (size_t)act_cap));
    LMRL_CHECK_HIP(hipMemcpy(c->act_tok_d, act_tok, sizeof(int32_t) * 4 * (size_t)act_cap, hipMemcpyHostToDevice));
    c->act_cap = act_cap;
    return LMRL_OK;
}

int lmrl_maze_hist_action(lmrl_maze_tok_ctx *c, const lmrl_maze_traj *tr, const lmrl_maze_hist *h, int32_t *len0_d, int32_t *len1_d, int n, void *stream) {
    LMRL_REQUIRE(c && tr && hist_ok(h) && len0_d && n > 0, "lmrl_maze_hist_action: bad argument");
    LMRL_REQUIRE(c->act_tok_d, "lmrl_maze_hist_action: call lmrl_maze_tok_set_actions first");
    hipLaunchKernelGGL(maze_hist_action_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), *tr, hist_of(h), c->tok_blen_d, c->vocab, c->max_new,
                       c->act_tok_d, c->act_cap, len0_d, len1_d, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_maze_tok_ppo_records(lmrl_maze_tok_ctx *c, const lmrl_maze_traj *tr, const void *state_d, int n, int n_envs, const int32_t *off_d, int newline_tok,
                              int byte_ids, int cap, int32_t *tokens_d, uint8_t *is_action_d, float *reward_d, int32_t *n_tok_d, int32_t *chain_d, int32_t *pos_d, uint8_t *last_d,
                              uint8_t *done_d, int32_t *chain_total_d, void *stream) {
    LMRL_REQUIRE(c && tr && state_d && n > 0 && n <= n_envs && off_d && cap >= 2 && tokens_d && is_action_d && reward_d && n_tok_d && chain_d && pos_d && last_d && done_d &&
                     chain_total_d, "lmrl_maze_tok_ppo_records: bad argument");
    hipLaunchKernelGGL(maze_ppo_records_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, as_stream(stream), *tr, (const int32_t *)state_d, c->goal_slot_d,
                       c->obs_tok_d, c->obs_len_d, c->obs_cap, c->tok_bytes_d, c->tok_blen_d, c->vocab, c->rows, c->cols, c->max_new, c->max_turns, n, n_envs, off_d,
                       c->act_tok_d, c->act_cap, byte_ids, newline_tok, cap, tokens_d, is_action_d, reward_d, n_tok_d, chain_d, pos_d, last_d, done_d, chain_total_d);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_maze_tok_set_spaced(lmrl_maze_tok_ctx *c, const int32_t *obs_sp_tok, const int32_t *obs_sp_len, int obs_sp_cap, const int32_t *act_sp_tok, int act_sp_cap) {
    LMRL_REQUIRE(c && obs_sp_tok && obs_sp_len && obs_sp_cap > 0 && act_sp_tok && act_sp_cap >= 2, "lmrl_maze_tok_set_spaced: bad argument");
    for (int i = 0; i < c->n_obs; i++) LMRL_REQUIRE(obs_sp_len[i] >= 0 && obs_sp_len[i] <= obs_sp_cap, "lmrl_maze_tok_set_spaced: observation length outside [0, cap]");
    for (int a = 0; a < 4; a++)
        LMRL_REQUIRE(act_sp_tok[a * act_sp_cap + act_sp_cap - 1] > 0 && act_sp_tok[a * act_sp_cap + act_sp_cap - 1] < act_sp_cap, "lmrl_maze_tok_set_spaced: action length outside (0, cap)");
    for (void *p : {(void *)c->obs_sp_tok_d, (void *)c->obs_sp_len_d, (void *)c->act_sp_tok_d})
        if (p) (void)hipFree(p);
    c->obs_sp_tok_d = c->obs_sp_len_d = c->act_sp_tok_d = nullptr;
    LMRL_CHECK_HIP(hipMalloc((void **)&c->obs_sp_tok_d, sizeof(int32_t) * (size_t)c->n_obs * obs_sp_cap));
    LMRL_CHECK_HIP(hipMemcpy(c->obs_sp_tok_d, obs_sp_tok, sizeof(int32_t) * (size_t)c->n_obs * obs_sp_cap, hipMemcpyHostToDevice));
    LMRL_CHECK_HIP(hipMalloc((void **)&c->obs_sp_len_d, sizeof(int32_t) * (size_t)c->n_obs));
    LMRL_CHECK_HIP(hipMemcpy(c->obs_sp_len_d, obs_sp_len, sizeof(int32_t) * (size_t)c->n_obs, hipMemcpyHostToDevice));
    LMRL_CHECK_HIP(hipMalloc((void **)&c->act_sp_tok_d, sizeof(int32_t) * 4 * (size_t)act_sp_cap));
    LMRL_CHECK_HIP(hipMemcpy(c->act_sp_tok_d, act_sp_tok, sizeof(int32_t) * 4 * (size_t)act_sp_cap, hipMemcpyHostToDevice));
    c->obs_sp_cap = obs_sp_cap; c->act_sp_cap = act_sp_cap;
    return LMRL_OK;
}

int lmrl_maze_tok_ppo_records_hist(lmrl_maze_tok_ctx *c, const lmrl_maze_traj *tr, const void *state_d, int n, int n_envs, const int32_t *off_d, int last_k,
                                   int newline_tok, int byte_ids, int cap, int32_t *tokens_d, uint8_t *is_action_d, float *reward_d, int32_t *n_tok_d, int32_t *chain_d,
                                   int32_t *pos_d, uint8_t *last_d, uint8_t *done_d, int32_t *chain_total_d, void *stream) {
    const bool count_only = tokens_d == nullptr;                    // first pass: n_tok_d only (the longest row sizes `cap`)
    LMRL_REQUIRE(c && tr && state_d && n > 0 && n <= n_envs && off_d && last_k >= 1 && last_k <= 64 && n_tok_d && chain_d && pos_d && last_d && done_d && chain_total_d,
                 "lmrl_maze_tok_ppo_records_hist: bad argument (last_k must be <= 64)");
    LMRL_REQUIRE(count_only || (cap >= 2 && is_action_d && reward_d), "lmrl_maze_tok_ppo_records_hist: bad argument");
    LMRL_REQUIRE(c->act_tok_d && c->obs_sp_tok_d, "lmrl_maze_tok_ppo_records_hist: call lmrl_maze_tok_set_actions and lmrl_maze_tok_set_spaced first");
    hipLaunchKernelGGL(maze_ppo_records_hist_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, as_stream(stream), *tr, (const int32_t *)state_d, c->goal_slot_d,
                       c->obs_tok_d, c->obs_len_d, c->obs_cap, c->obs_sp_tok_d, c->obs_sp_len_d, c->obs_sp_cap, c->act_tok_d, c->act_cap, c->act_sp_tok_d, c->act_sp_cap,
                       c->tok_bytes_d, c->tok_blen_d, c->vocab, c->rows, c->cols, c->max_new, c->max_turns, n, n_envs, off_d, last_k, newline_tok, cap, count_only ? 1 : 0,
                       byte_ids, tokens_d, is_action_d, reward_d, n_tok_d, chain_d, pos_d, last_d, done_d, chain_total_d);
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
