// wordle_tokens.hip — on-device token <-> game bookkeeping for lock-step Wordle rollouts.
//
// Replaces the host round trip the reference makes on EVERY turn of interact_environment
// (LLM_RL/environment.py:180-206): decode generated ids to text, strip the prompt, `out_str_process`
// (ppo/gpt2/interface.py:538-541), deformat the action (wordle/env/env.py:19-26), format the observation
// (env.py:7-17) and re-tokenise the whole history.  Here the sampled token ids are classified through a
// per-vocabulary table, turned into a packed guess for lmrl_wordle_step, and the observation is appended as
// token ids; nothing leaves HBM during an episode.
//
// Text semantics implemented exactly:
//   action text  = concat(token strings up to and excluding eos) ; `.strip().replace(' ', '')`
//   valid guess  <=> the result is exactly 5 chars a-z                       (game.py:214)
//   stored text  = action text with one trailing '\n' forced                 (interface.py:541)
//   observation  = ' '.join(symbols) + '\n'   or   '\n' when there are none  (env.py:13-16)
// One thread per env; all arrays struct-of-arrays / row-per-env int32, tiny next to the model traffic.
#include "../../include/lmrl_amd.h"
#include "common.h"

namespace lmrl {

struct WordleTokCtx {
    lmrl_wordle_tokens t;
    uint32_t *cls_d = nullptr;   // [vocab] token classes
    int vocab = 0, max_new = 0, cap = 0;
};

// token class word: bits 0-24 up to five 5-bit letters, 25-27 letter count (7 = token makes the action invalid),
// bit 28 whitespace other than ' ' before the first letter (or anywhere if no letters), bit 29 ... after the last letter
__device__ __forceinline__ uint32_t cls_nl(uint32_t c) { return (c >> 25) & 7u; }

__global__ void tok_begin_kernel(lmrl_wordle_tokens t, lmrl_wordle_traj tr, int32_t *chunk_tok, int32_t *chunk_cnt, int cap,
                                 int n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    for (int k = 0; k < 8; k++) chunk_tok[e * 8 + k] = k < t.n_header ? t.header[k] : t.pad;
    chunk_cnt[e] = t.n_header;
    for (int k = 0; k < t.n_header && k < cap; k++) {
        tr.tokens[(size_t)e * cap + k] = t.header[k];
        tr.is_action[(size_t)e * cap + k] = 0;
        tr.reward[(size_t)e * cap + k] = 0.f;
    }
    tr.n_tok[e] = t.n_header;
    tr.gen_len[e] = 0;
    tr.gen_active[e] = 1;
    tr.env_done[e] = 0;
    tr.n_steps[e] = 0;
    tr.ep_reward[e] = 0.f;
}

// k-th sampled token of the current action
__global__ void tok_accept_kernel(lmrl_wordle_tokens t, lmrl_wordle_traj tr, const int32_t *sampled, int k, int max_new,
                                  int32_t *next_tok, int32_t *next_cnt, uint8_t *active_out, int n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    int cnt = 0;
    if (tr.gen_active[e]) {
        const int tok = sampled[e];
        tr.gen[(size_t)e * max_new + k] = tok;
        tr.gen_len[e] = k + 1;
        if (tok == t.newline || k + 1 >= max_new) tr.gen_active[e] = 0;   // eos reached or max_new_tokens
        else { cnt = 1; next_tok[e] = tok; }
    }
    next_cnt[e] = cnt;
    if (active_out) active_out[e] = (uint8_t)cnt;
}

// decode the generated ids into a packed guess; append the action tokens (+ forced '\n') to the trajectory
__global__ void tok_guess_kernel(lmrl_wordle_tokens t, lmrl_wordle_traj tr, const uint32_t *cls, int vocab, int max_new,
                                 int cap, uint32_t *guess, uint8_t *active, int n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const bool live = !tr.env_done[e];
    active[e] = live ? 1 : 0;
    if (!live) { guess[e] = 0xFFFFFFFFu; return; }
    const int len = tr.gen_len[e];
    uint32_t letters = 0, nl = 0;
    bool bad = false, started = false, pend_w = false, saw_eos = false;
    int nt = tr.n_tok[e];
    for (int k = 0; k < len; k++) {
        const int tok = tr.gen[(size_t)e * max_new + k];
        if (nt < cap) { tr.tokens[(size_t)e * cap + nt] = tok; tr.is_action[(size_t)e * cap + nt] = 1; tr.reward[(size_t)e * cap + nt] = 0.f; nt++; }
        if (tok == t.newline) { saw_eos = true; break; }
        const uint32_t c = (tok >= 0 && tok < vocab) ? cls[tok] : (7u << 25);
        const uint32_t cn = cls_nl(c);
        if (cn == 7u) { bad = true; continue; }
        if (cn > 0) {
            if (pend_w || (started && (c >> 28 & 1u))) bad = true;
            for (uint32_t i = 0; i < cn; i++) {
                if (nl < 5) letters |= ((c >> (5 * i)) & 31u) << (5 * nl);
                nl++;
            }
            started = true;
            if (c >> 29 & 1u) pend_w = true;
        } else if ((c >> 28 & 1u) && started) {
            pend_w = true;
        }
    }
    if (!saw_eos && nt < cap) {   // out_str_process: removesuffix('\n') + '\n'
        tr.tokens[(size_t)e * cap + nt] = t.newline; tr.is_action[(size_t)e * cap + nt] = 1; tr.reward[(size_t)e * cap + nt] = 0.f; nt++;
    }
    tr.n_tok[e] = nt;
    tr.pend_newline[e] = saw_eos ? 0 : 1;
    guess[e] = (!bad && nl == 5) ? letters : 0xFFFFFFFFu;
}

// after lmrl_wordle_step: reward onto the action's last token, observation tokens into the trajectory, next chunk
__global__ void tok_observe_kernel(lmrl_wordle_tokens t, lmrl_wordle_traj tr, const uint32_t *obs, const float *reward,
                                   const uint8_t *flags, int max_new, int cap, int32_t *chunk_tok, int32_t *chunk_cnt, int n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    for (int k = 0; k < 8; k++) chunk_tok[e * 8 + k] = t.pad;
    if (tr.env_done[e]) { chunk_cnt[e] = 0; return; }
    int nt = tr.n_tok[e];
    if (nt > 0) tr.reward[(size_t)e * cap + nt - 1] = reward[e];   // reward sits on the last token of the action Text (environment.py:370)
    tr.ep_reward[e] += reward[e];
    tr.n_steps[e] += 1;
    // tokens not yet forwarded through the model: last generated token (always), forced '\n' (if no eos was sampled)
    int c = 0;
    const int len = tr.gen_len[e];
    if (len > 0) chunk_tok[e * 8 + c++] = tr.gen[(size_t)e * max_new + len - 1];
    if (tr.pend_newline[e]) chunk_tok[e * 8 + c++] = t.newline;
    const uint32_t o = obs[e];
    const int ns = (int)((o >> 16) & 7u);
    for (int k = 0; k < ns; k++) {
        const int sym = (int)((o >> (3 * k)) & 7u) - 1;   // 0 g, 1 y, 2 b
        const int tok = k == 0 ? t.sym_first[sym] : t.sym_sp[sym];
        if (c < 8) chunk_tok[e * 8 + c++] = tok;
        if (nt < cap) { tr.tokens[(size_t)e * cap + nt] = tok; tr.is_action[(size_t)e * cap + nt] = 0; tr.reward[(size_t)e * cap + nt] = 0.f; nt++; }
    }
    if (c < 8) chunk_tok[e * 8 + c++] = t.newline;
    if (nt < cap) { tr.tokens[(size_t)e * cap + nt] = t.newline; tr.is_action[(size_t)e * cap + nt] = 0; tr.reward[(size_t)e * cap + nt] = 0.f; nt++; }
    tr.n_tok[e] = nt;
    const bool done = flags[e] & 1;
    tr.env_done[e] = done ? 1 : 0;
    tr.gen_active[e] = done ? 0 : 1;
    tr.gen_len[e] = 0;
    chunk_cnt[e] = done ? 0 : c;
}

// synthetic-workload helper: the token that spells letter k of a scripted guess (k == 5 -> newline)
__global__ void tok_steer_kernel(lmrl_wordle_tokens t, const uint32_t *scripted_guess, int k, int32_t *steer, int n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    if (k < 0) {   // all six positions at once: steer[j][e]
        const uint32_t g = scripted_guess[e];
        for (int j = 0; j < 5; j++) {
            const uint32_t c = (g >> (5 * j)) & 31u;
            steer[(size_t)j * n + e] = j == 0 ? t.letter_first[c % 26] : t.letter_sp[c % 26];
        }
        steer[(size_t)5 * n + e] = t.newline;
        return;
    }
    if (k >= 5) { steer[e] = t.newline; return; }
    const uint32_t c = (scripted_guess[e] >> (5 * k)) & 31u;
    steer[e] = k == 0 ? t.letter_first[c % 26] : t.letter_sp[c % 26];
}

}  // namespace lmrl

using namespace lmrl;

struct lmrl_wordle_tok_ctx : public WordleTokCtx {};

extern "C" {

lmrl_wordle_tok_ctx *lmrl_wordle_tok_create(const lmrl_wordle_tokens *tokens, const uint32_t *token_class, int vocab,
                                            int max_new_tokens, int traj_cap) {
    if (!tokens || !token_class || vocab <= 0 || max_new_tokens <= 0 || traj_cap <= 0 || tokens->n_header < 0 || tokens->n_header > 8) {
        set_error("lmrl_wordle_tok_create: bad argument");
        return nullptr;
    }
    lmrl_wordle_tok_ctx *c = new lmrl_wordle_tok_ctx();
    c->t = *tokens; c->vocab = vocab; c->max_new = max_new_tokens; c->cap = traj_cap;
    if (hipMalloc(&c->cls_d, sizeof(uint32_t) * vocab) != hipSuccess ||
        hipMemcpy(c->cls_d, token_class, sizeof(uint32_t) * vocab, hipMemcpyHostToDevice) != hipSuccess) {
        set_error("lmrl_wordle_tok_create: device allocation/copy failed");
        delete c;
        return nullptr;
    }
    return c;
}

void lmrl_wordle_tok_destroy(lmrl_wordle_tok_ctx *c) {
    if (!c) return;
    if (c->cls_d) (void)hipFree(c->cls_d);
    delete c;
}

#define TOK_GRID(n) dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream)

int lmrl_wordle_tok_begin(lmrl_wordle_tok_ctx *c, const lmrl_wordle_traj *tr, int32_t *chunk_tok_d, int32_t *chunk_cnt_d, int n,
                          void *stream) {
    LMRL_REQUIRE(c && tr && chunk_tok_d && chunk_cnt_d && n > 0, "lmrl_wordle_tok_begin: bad argument");
    hipLaunchKernelGGL(tok_begin_kernel, TOK_GRID(n), c->t, *tr, chunk_tok_d, chunk_cnt_d, c->cap, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_wordle_tok_accept(lmrl_wordle_tok_ctx *c, const lmrl_wordle_traj *tr, const int32_t *sampled_d, int k,
                           int32_t *next_tok_d, int32_t *next_cnt_d, uint8_t *active_d, int n, void *stream) {
    LMRL_REQUIRE(c && tr && sampled_d && next_tok_d && next_cnt_d && n > 0 && k >= 0 && k < c->max_new, "lmrl_wordle_tok_accept: bad argument");
    hipLaunchKernelGGL(tok_accept_kernel, TOK_GRID(n), c->t, *tr, sampled_d, k, c->max_new, next_tok_d, next_cnt_d, active_d, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_wordle_tok_guess(lmrl_wordle_tok_ctx *c, const lmrl_wordle_traj *tr, uint32_t *guess_d, uint8_t *active_d, int n,
                          void *stream) {
    LMRL_REQUIRE(c && tr && guess_d && active_d && n > 0, "lmrl_wordle_tok_guess: bad argument");
    hipLaunchKernelGGL(tok_guess_kernel, TOK_GRID(n), c->t, *tr, c->cls_d, c->vocab, c->max_new, c->cap, guess_d, active_d, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_wordle_tok_observe(lmrl_wordle_tok_ctx *c, const lmrl_wordle_traj *tr, const uint32_t *obs_d, const float *reward_d,
                            const uint8_t *flags_d, int32_t *chunk_tok_d, int32_t *chunk_cnt_d, int n, void *stream) {
    LMRL_REQUIRE(c && tr && obs_d && reward_d && flags_d && chunk_tok_d && chunk_cnt_d && n > 0, "lmrl_wordle_tok_observe: bad argument");
    hipLaunchKernelGGL(tok_observe_kernel, TOK_GRID(n), c->t, *tr, obs_d, reward_d, flags_d, c->max_new, c->cap, chunk_tok_d,
                       chunk_cnt_d, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_wordle_tok_steer(lmrl_wordle_tok_ctx *c, const uint32_t *scripted_guess_d, int k, int32_t *steer_d, int n, void *stream) {
    LMRL_REQUIRE(c && scripted_guess_d && steer_d && n > 0, "lmrl_wordle_tok_steer: bad argument");
    hipLaunchKernelGGL(tok_steer_kernel, TOK_GRID(n), c->t, scripted_guess_d, k, steer_d, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
}
