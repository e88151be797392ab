// wordle.hip — batched, bit-exact Wordle state-transition kernels for gfx950.
//
// Replaces the per-instance Python objects of the reference
//   WordleEnvironment.reset/step   llm_rl_scripts/wordle/env/env.py:39-55
//   WordleGame.next / reward / is_terminal / transition_sequence   wordle/env/game.py:213-296
//   Vocabulary filter (order preserving) + rng.choice             wordle/env/game.py:150-179
// with one launch that steps N envs in lock-step.
//
// Mapping to the hardware: ONE 64-lane wavefront per env (4 envs per 256-thread workgroup).
// The env's knowledge masks are wave-uniform (SGPR-resident after the uniform loads); the
// 64 lanes sweep the vocabulary 64 words per iteration (8 B/word, L2-resident table shared by
// every wave), `__ballot` + popcount give the order-preserving rank needed for
// `filtered_vocab[r]`.  HBM traffic per env-step is the 76 B of game state read+written plus
// one MT19937 word: the kernel is HBM/latency-bound, not ALU-bound.
#include "../../include/lmrl_amd.h"
#include "common.h"
#include "mt19937.h"
#include "wordle_core.h"

namespace lmrl {

struct WordleCtx {
    uint32_t *words_d = nullptr;   // [V] packed 5x5-bit letters, file order
    uint32_t *wmask_d = nullptr;   // [V] 26-bit letter-presence masks
    int V = 0;
    int require = 1;
    float bad_reward = -1.f;
    int step_variant = 0;          // 0: by batch size, 1: one wave per env, 2: one lane per env (lmrl_wordle_set_variant)
};

// rows of the SoA game-state buffer
enum { ROW_FORB = 0, ROW_MUST = 5, ROW_NFILT = 10, ROW_NACT = 11, ROW_HIST = 12 };

__global__ __launch_bounds__(kMtSeedLanes) void wordle_reset_kernel(uint32_t *st, void *mt, const uint64_t *seeds, const uint8_t *mask,
                                                                    const uint32_t *table, int V, int n) {
    extern __shared__ __attribute__((aligned(16))) uint32_t mt_lds[];
    const int e = blockIdx.x * kMtSeedLanes + threadIdx.x;
    const bool on = e < n && (!mask || mask[e]);
    if (on) {
        for (int k = 0; k < 10; k++) st[(size_t)k * n + e] = 0u;           // all POSSIBLE (game.py:72-74)
        st[(size_t)ROW_NFILT * n + e] = (uint32_t)V;                       // filtered_vocab = all_vocab
        st[(size_t)ROW_NACT * n + e] = 0u;
        for (int k = 0; k < kWordleTries; k++) st[(size_t)(ROW_HIST + k) * n + e] = kBadGuess;
    }
    mt_seed_and_twist_lds(mt_lds, mt, n, e, on, on ? seeds[e] : 0ull, table);   // env.py:53, state staged in LDS
}

__global__ __launch_bounds__(256) void wordle_step_kernel(const uint32_t *__restrict__ words,
                                                          const uint32_t *__restrict__ wmasks, int V, int require,
                                                          float bad_reward, uint32_t *st, void *mt,
                                                          const uint32_t *guess, const uint8_t *active, uint32_t *obs,
                                                          float *reward, uint8_t *flags, int n) {
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= n) return;                       // wave-uniform
    if (active && !active[e]) return;         // wave-uniform

    WordleMasks s;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        s.forb[i] = st[(size_t)(ROW_FORB + i) * n + e];
        s.must[i] = st[(size_t)(ROW_MUST + i) * n + e];
    }
    wordle_derive(s);
    const uint32_t nfilt = st[(size_t)ROW_NFILT * n + e];
    const uint32_t nact = st[(size_t)ROW_NACT * n + e];
    const uint32_t g = guess[e];
    const bool shaped = (g != kBadGuess);     // len == 5 and all a-z (game.py:214, first two clauses)

    // ---- `action in self.vocab` (game.py:187-188)
    bool member = false;
    if (shaped) {
        for (int base = 0; base < V; base += 64) {
            const int i = base + lane;
            const uint32_t w = i < V ? words[i] : kBadGuess;
            if (__any(w == g)) { member = true; break; }
        }
    }
    // game.py:214-219: a well-formed action (in the vocabulary when required) goes through `rng.choice(filtered_vocab)`.  On an EMPTY
    // filtered vocabulary the reference raises IndexError there (random.choice([]), game.py:178-179).  The state can not get there: the
    // sampled target always satisfies the state it produces (HERE only where it matches, NOT_HERE where it differs, all-NOT_HERE only for
    // letters it lacks), so the target itself stays in the filtered list.  Should a corrupted state arrive here anyway, the step does not
    // silently diverge: flag bit 8 is raised, the env is finished, and the host wrappers raise IndexError like the reference.
    const bool would_choose = shaped && (member || !require);
    const bool empty_choice = would_choose && nfilt == 0;
    const bool valid = would_choose && nfilt > 0;                      // goes through transition_state
    const bool bad_word = !(shaped && member);                        // reward() first clause (game.py:291-292)

    uint32_t new_nfilt = nfilt;
    uint32_t o = 0;            // observation symbols
    uint32_t uniq = kBadGuess; // filtered_vocab[0] after the transition

    if (valid) {
        // ---- word = rng.choice(filtered_vocab)  (game.py:219): r = _randbelow(len), then the r-th consistent word
        uint32_t r = 0;
        if (lane == 0) r = mt_randbelow(mt_ref(mt, n, e), nfilt);
        r = __shfl(r, 0);
        uint32_t cnt = 0, target = g;
        for (int base = 0; base < V; base += 64) {
            const int i = base + lane;
            const uint32_t w = i < V ? words[i] : 0u;
            const bool ok = i < V && wordle_consistent(s, w, wmasks[i]);
            const unsigned long long b = __ballot(ok);
            const uint32_t c = (uint32_t)__popcll(b);
            if (r < cnt + c) {
                const uint32_t want = r - cnt;
                const uint32_t myrank = (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
                const unsigned long long mb = __ballot(ok && myrank == want);
                target = __shfl(w, __ffsll((long long)mb) - 1);
                break;
            }
            cnt += c;
        }
        // ---- new_state = state.transition_state(action, word)  (game.py:220, 82-92)
        wordle_transition(s, g, target);
        // ---- vocab.update_vocab(new_state): recount + first element  (game.py:154)
        cnt = 0;
        for (int base = 0; base < V; base += 64) {
            const int i = base + lane;
            const uint32_t w = i < V ? words[i] : 0u;
            const bool ok = i < V && wordle_consistent(s, w, wmasks[i]);
            const unsigned long long b = __ballot(ok);
            if (b && cnt == 0) uniq = __shfl(w, __ffsll((long long)b) - 1);
            cnt += (uint32_t)__popcll(b);
        }
        new_nfilt = cnt;
        o = wordle_obs(s, g);
    }

    if (lane == 0) {
        // action_history + [action]
        if (nact < (uint32_t)kWordleTries) st[(size_t)(ROW_HIST + nact) * n + e] = g;
        const uint32_t new_nact = nact + 1;
        float rew;
        if (bad_word) {
            rew = bad_reward;
        } else {
            // int(filtered_vocab_size() == 1 and filtered_vocab[0] in action_history) - 1   (game.py:293)
            bool win = false;
            if (new_nfilt == 1) {
                win = (uniq == g);
                for (uint32_t k = 0; k < nact && k < (uint32_t)kWordleTries; k++)
                    win |= (st[(size_t)(ROW_HIST + k) * n + e] == uniq);
            }
            rew = win ? 0.f : -1.f;
        }
        const bool done = (new_nact == (uint32_t)kWordleTries) || (rew == 0.f);   // game.py:295-296
        if (valid) {
#pragma unroll
            for (int i = 0; i < 5; i++) {
                st[(size_t)(ROW_FORB + i) * n + e] = s.forb[i];
                st[(size_t)(ROW_MUST + i) * n + e] = s.must[i];
            }
            st[(size_t)ROW_NFILT * n + e] = new_nfilt;
        }
        st[(size_t)ROW_NACT * n + e] = new_nact;
        obs[e] = o;
        reward[e] = rew;
        flags[e] = (uint8_t)(((done || empty_choice) ? 1 : 0) | (valid ? 2 : 0) | (bad_word ? 4 : 0) | (empty_choice ? 8 : 0));
    }
}

// ---- lane-per-env form of the same step, for LARGE batches (env-only workloads, tens of thousands of envs): lane = env, so every state word
// is one coalesced 256-byte access per wave (the SoA layout of DESIGN.md section 3 taken literally), the vocabulary sits in LDS and is read at
// one (broadcast) address per iteration, and there are no cross-lane operations.  A lane walks the whole vocabulary serially (3 passes), so the
// kernel needs thousands of waves to fill the chip: lmrl_wordle_step picks it from 65 536 envs up (262 144 for the long word list), the
// wave-per-env kernel below that (1024 envs: 12 us there vs ~40 us here).  Same arithmetic, same RNG calls in the same order: bit-identical states and outputs.
__global__ __launch_bounds__(256) void wordle_step_lanes_kernel(const uint32_t *__restrict__ words, const uint32_t *__restrict__ wmasks, int V,
                                                                int require, float bad_reward, uint32_t *st, void *mt, const uint32_t *guess,
                                                                const uint8_t *active, uint32_t *obs, float *reward, uint8_t *flags, int n) {
    extern __shared__ uint32_t voc[];                  // [V] words, [V] letter masks
    for (int i = threadIdx.x; i < V; i += blockDim.x) { voc[i] = words[i]; voc[V + i] = wmasks[i]; }
    __syncthreads();
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n || (active && !active[e])) return;
    WordleMasks s;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        s.forb[i] = st[(size_t)(ROW_FORB + i) * n + e];
        s.must[i] = st[(size_t)(ROW_MUST + i) * n + e];
    }
    wordle_derive(s);
    const uint32_t nfilt = st[(size_t)ROW_NFILT * n + e];
    const uint32_t nact = st[(size_t)ROW_NACT * n + e];
    const uint32_t g = guess[e];
    const bool shaped = (g != kBadGuess);
    bool member = false;
    if (shaped)
        for (int i = 0; i < V; i++) member |= (voc[i] == g);
    const bool would_choose = shaped && (member || !require);
    const bool empty_choice = would_choose && nfilt == 0;
    const bool valid = would_choose && nfilt > 0;
    const bool bad_word = !(shaped && member);
    uint32_t new_nfilt = nfilt, o = 0, uniq = kBadGuess;
    if (valid) {
        const uint32_t r = mt_randbelow(mt_ref(mt, n, e), nfilt);      // rng.choice(filtered_vocab) (game.py:219)
        uint32_t cnt = 0, target = g;
        for (int i = 0; i < V; i++) {
            const uint32_t w = voc[i];
            const bool ok = wordle_consistent(s, w, voc[V + i]);
            target = (ok && cnt == r) ? w : target;
            cnt += ok ? 1u : 0u;
        }
        wordle_transition(s, g, target);
        cnt = 0;
        for (int i = 0; i < V; i++) {
            const uint32_t w = voc[i];
            const bool ok = wordle_consistent(s, w, voc[V + i]);
            uniq = (ok && cnt == 0) ? w : uniq;
            cnt += ok ? 1u : 0u;
        }
        new_nfilt = cnt;
        o = wordle_obs(s, g);
    }
    if (nact < (uint32_t)kWordleTries) st[(size_t)(ROW_HIST + nact) * n + e] = g;
    const uint32_t new_nact = nact + 1;
    float rew;
    if (bad_word) {
        rew = bad_reward;
    } else {
        bool win = false;
        if (new_nfilt == 1) {
            win = (uniq == g);
            for (uint32_t k = 0; k < nact && k < (uint32_t)kWordleTries; k++) win |= (st[(size_t)(ROW_HIST + k) * n + e] == uniq);
        }
        rew = win ? 0.f : -1.f;
    }
    const bool done = (new_nact == (uint32_t)kWordleTries) || (rew == 0.f);
    if (valid) {
#pragma unroll
        for (int i = 0; i < 5; i++) {
            st[(size_t)(ROW_FORB + i) * n + e] = s.forb[i];
            st[(size_t)(ROW_MUST + i) * n + e] = s.must[i];
        }
        st[(size_t)ROW_NFILT * n + e] = new_nfilt;
    }
    st[(size_t)ROW_NACT * n + e] = new_nact;
    obs[e] = o;
    reward[e] = rew;
    flags[e] = (uint8_t)(((done || empty_choice) ? 1 : 0) | (valid ? 2 : 0) | (bad_word ? 4 : 0) | (empty_choice ? 8 : 0));
}

__global__ void wordle_export_kernel(const uint32_t *st, uint8_t *trits, uint32_t *nf, uint32_t *na, int n) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    uint32_t forb[5], must[5];
    for (int i = 0; i < 5; i++) {
        forb[i] = st[(size_t)(ROW_FORB + i) * n + e];
        must[i] = st[(size_t)(ROW_MUST + i) * n + e];
    }
    for (int c = 0; c < 26; c++)
        for (int i = 0; i < 5; i++)
            trits[((size_t)e * 26 + c) * 5 + i] = (must[i] >> c & 1u) ? 2 : ((forb[i] >> c & 1u) ? 0 : 1);
    if (nf) nf[e] = st[(size_t)ROW_NFILT * n + e];
    if (na) na[e] = st[(size_t)ROW_NACT * n + e];
}

int mt_table(const uint32_t **out);  // mt19937.hip

}  // namespace lmrl

using namespace lmrl;

struct lmrl_wordle_ctx : public WordleCtx {};

extern "C" {

lmrl_wordle_ctx *lmrl_wordle_create(const char *words5, int n_words, int require_words_in_vocab, float bad_word_reward) {
    if (!words5 || n_words <= 0) {
        set_error("lmrl_wordle_create: empty vocabulary");
        return nullptr;
    }
    uint32_t *packed = new uint32_t[n_words];
    uint32_t *masks = new uint32_t[n_words];
    for (int w = 0; w < n_words; w++) {
        uint32_t p = 0;
        for (int i = 0; i < 5; i++) {
            char c = words5[(size_t)w * 5 + i];
            if (c < 'a' || c > 'z') {
                set_error("lmrl_wordle_create: word %d has a character outside a-z", w);
                delete[] packed; delete[] masks;
                return nullptr;
            }
            p |= (uint32_t)(c - 'a') << (5 * i);
        }
        packed[w] = p;
        masks[w] = letters_mask(p);
    }
    lmrl_wordle_ctx *ctx = new lmrl_wordle_ctx();
    ctx->V = n_words;
    ctx->require = require_words_in_vocab ? 1 : 0;
    ctx->bad_reward = bad_word_reward;
    bool ok = hipMalloc(&ctx->words_d, sizeof(uint32_t) * n_words) == hipSuccess &&
              hipMalloc(&ctx->wmask_d, sizeof(uint32_t) * n_words) == hipSuccess &&
              hipMemcpy(ctx->words_d, packed, sizeof(uint32_t) * n_words, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(ctx->wmask_d, masks, sizeof(uint32_t) * n_words, hipMemcpyHostToDevice) == hipSuccess;
    delete[] packed; delete[] masks;
    if (!ok) {
        set_error("lmrl_wordle_create: device allocation/copy failed (is a GPU visible?)");
        lmrl_wordle_destroy(ctx);
        return nullptr;
    }
    return ctx;
}

void lmrl_wordle_destroy(lmrl_wordle_ctx *ctx) {
    if (!ctx) return;
    if (ctx->words_d) (void)hipFree(ctx->words_d);
    if (ctx->wmask_d) (void)hipFree(ctx->wmask_d);
    delete ctx;
}

int lmrl_wordle_set_variant(lmrl_wordle_ctx *ctx, int variant) {
    LMRL_REQUIRE(ctx && variant >= 0 && variant <= 2, "lmrl_wordle_set_variant: 0 = by batch size, 1 = one wave per env, 2 = one lane per env");
    ctx->step_variant = variant;
    return LMRL_OK;
}

size_t lmrl_wordle_state_bytes(int n) { return (size_t)kWordleStateWords * (size_t)n * sizeof(uint32_t); }

int lmrl_wordle_reset(lmrl_wordle_ctx *ctx, void *state_d, void *mt_d, const uint64_t *seeds_d, const uint8_t *mask_d,
                      int n, void *stream) {
    LMRL_REQUIRE(ctx && state_d && mt_d && seeds_d && n >= 0, "lmrl_wordle_reset: null pointer or negative n");
    if (n == 0) return LMRL_OK;
    const uint32_t *table;
    int rc = mt_table(&table);
    if (rc) return rc;
    constexpr size_t lds = (size_t)kMtN * kMtSeedLanes * sizeof(uint32_t);
    static bool attr = false;
    if (!attr) {
        LMRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&wordle_reset_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    hipLaunchKernelGGL(wordle_reset_kernel, dim3(ceil_div(n, kMtSeedLanes)), dim3(kMtSeedLanes), lds, as_stream(stream),
                       (uint32_t *)state_d, mt_d, seeds_d, mask_d, table, ctx->V, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_wordle_step(lmrl_wordle_ctx *ctx, void *state_d, void *mt_d, const uint32_t *guess_d, const uint8_t *active_d,
                     uint32_t *obs_d, float *reward_d, uint8_t *flags_d, int n, void *stream) {
    LMRL_REQUIRE(ctx && state_d && mt_d && guess_d && obs_d && reward_d && flags_d && n >= 0,
                 "lmrl_wordle_step: null pointer or negative n");
    if (n == 0) return LMRL_OK;
    ProfScope ps(PROF_WORDLE_STEP, as_stream(stream), 96.0 * n);   // ~96 algorithmic bytes per env-step
    const size_t voc_lds = (size_t)ctx->V * 2 * sizeof(uint32_t);
    // measured (tools/bench_env.py, profiles/r03_env_only_microbench.txt; steps only, M env-steps/s, wave/env vs lane/env):
    //   V = 431:  65 536 envs 678 vs 746, 262 144 envs 711 vs 1549;   V = 2315:  65 536 envs 222 vs 151, 262 144 envs 233 vs 309
    const bool lanes = ctx->step_variant == 2 || (ctx->step_variant == 0 && (n >= 262144 || (n >= 65536 && ctx->V <= 1024)));
    if (lanes && voc_lds <= 64 * 1024) {      // lane = env: large batches (the vocabulary tables fit LDS for any Wordle word list: 18.5 KB at V = 2315)
        hipLaunchKernelGGL(wordle_step_lanes_kernel, dim3(ceil_div(n, 256)), dim3(256), voc_lds, as_stream(stream), ctx->words_d, ctx->wmask_d, ctx->V,
                           ctx->require, ctx->bad_reward, (uint32_t *)state_d, mt_d, guess_d, active_d, obs_d, reward_d, flags_d, n);
        LMRL_CHECK_LAUNCH();
        return LMRL_OK;
    }
    hipLaunchKernelGGL(wordle_step_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, as_stream(stream), ctx->words_d,
                       ctx->wmask_d, ctx->V, ctx->require, ctx->bad_reward, (uint32_t *)state_d, mt_d, guess_d, active_d,
                       obs_d, reward_d, flags_d, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_wordle_export_state(const void *state_d, uint8_t *trits_d, uint32_t *n_filtered_d, uint32_t *n_actions_d, int n,
                             void *stream) {
    LMRL_REQUIRE(state_d && trits_d && n >= 0, "lmrl_wordle_export_state: null pointer or negative n");
    if (n == 0) return LMRL_OK;
    hipLaunchKernelGGL(wordle_export_kernel, dim3(ceil_div(n, 64)), dim3(64), 0, as_stream(stream),
                       (const uint32_t *)state_d, trits_d, n_filtered_d, n_actions_d, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
}
