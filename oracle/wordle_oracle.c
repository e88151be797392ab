/*
 * oracle/wordle_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of the reference Wordle MDP, used only
 * as the checker in tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * The product path (lmrl-gym_amd/csrc/wordle.hip) never links or calls this file.
 *
 * Parity status: PINNED.  tests/test_oracle_wordle.py replays every episode of
 * tests/golden/wordle_traces_{v431,v2315}.json (produced by running the reference's own
 * Python code, see tests/golden/make_fixtures.py) and requires identical observation
 * strings, rewards, done flags, per-letter knowledge state and filtered-vocab sizes.
 *
 * Each function cites the reference code it follows (paths relative to /root/reference).
 * The structure deliberately mirrors the reference (per-letter CharState arrays, a
 * word_satisfies() per letter, order-preserving filter) instead of the bit-mask
 * formulation the HIP kernel uses, so the two are independent derivations.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ALPHA 26   /* llm_rl_scripts/wordle/env/game.py:13 ALPHA_SIZE */
#define NCH 5      /* game.py:14 N_CHARS */
#define NTRIES 6   /* game.py:15 N_TRIES */
enum { NOT_HERE = 0, POSSIBLE = 1, HERE = 2 }; /* game.py:17-20 CharKnowledge */

/* ------------------------------------------------------------------ MT19937 (CPython _randommodule.c) */
typedef struct { uint32_t mt[624]; int idx; } orc_mt;

static void mt_init_genrand(orc_mt *s, uint32_t seed) {
    s->mt[0] = seed;
    for (int i = 1; i < 624; i++)
        s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->idx = 624;
}

/* random.Random(int) -> init_by_array(32-bit little-endian limbs of |seed|); SURVEY Appendix A.6 */
void orc_mt_seed(orc_mt *s, const uint32_t *key, int klen) {
    mt_init_genrand(s, 19650218u);
    int i = 1, j = 0, k = (624 > klen ? 624 : klen);
    for (; k; k--) {
        s->mt[i] = (s->mt[i] ^ ((s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        i++; j++;
        if (i >= 624) { s->mt[0] = s->mt[623]; i = 1; }
        if (j >= klen) j = 0;
    }
    for (k = 623; k; k--) {
        s->mt[i] = (s->mt[i] ^ ((s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= 624) { s->mt[0] = s->mt[623]; i = 1; }
    }
    s->mt[0] = 0x80000000u;
    s->idx = 624;
}

uint32_t orc_mt_next(orc_mt *s) {
    if (s->idx >= 624) {
        uint32_t *mt = s->mt, y;
        int kk;
        for (kk = 0; kk < 624 - 397; kk++) {
            y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        for (; kk < 623; kk++) {
            y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        s->idx = 0;
    }
    uint32_t y = s->mt[s->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* random.Random._randbelow_with_getrandbits: k = n.bit_length(); r = getrandbits(k) until r < n */
uint32_t orc_mt_randbelow(orc_mt *s, uint32_t n) {
    int k = 0;
    for (uint32_t t = n; t; t >>= 1) k++;
    uint32_t r = orc_mt_next(s) >> (32 - k);
    while (r >= n) r = orc_mt_next(s) >> (32 - k);
    return r;
}

/* flat helpers so tests can pin the generator against the stdlib directly */
void orc_mt_stream(const uint32_t *key, int klen, uint32_t *out, int n) {
    orc_mt s; orc_mt_seed(&s, key, klen);
    for (int i = 0; i < n; i++) out[i] = orc_mt_next(&s);
}
void orc_mt_choices(const uint32_t *key, int klen, const uint32_t *ns, uint32_t *out, int n) {
    orc_mt s; orc_mt_seed(&s, key, klen);
    for (int i = 0; i < n; i++) out[i] = orc_mt_randbelow(&s, ns[i]);
}

/* ------------------------------------------------------------------ game */
typedef struct {
    /* Vocabulary (game.py:134-191): ordered word list + the shared rng */
    int V;
    char *words;            /* V x 5 */
    int require_in_vocab;   /* WordleGame.require_words_in_vocab */
    double bad_word_reward;
    orc_mt rng;
    /* WordleGame (game.py:193-211) */
    uint8_t state[ALPHA][NCH];
    int *filtered;          /* indices into words, order preserving (game.py:154) */
    int n_filtered;
    int n_actions;
    char actions[NTRIES + 2][8];   /* valid-shaped guesses only matter for `in action_history` */
    int action_is_word[NTRIES + 2]; /* 1 if that action was exactly 5 bytes */
    int last_valid;         /* last action went through transition_state */
    int last_in_vocab, last_len5;
    char last_action[8];
} orc_wordle;

/* CharState.word_satisfies (game.py:53-65) */
static int word_satisfies(const uint8_t k[NCH], char c, const char *word) {
    int all_possible = 1, all_not_here = 1, in_word = 0;
    for (int i = 0; i < NCH; i++) {
        if (k[i] != POSSIBLE) all_possible = 0;
        if (k[i] != NOT_HERE) all_not_here = 0;
        if (word[i] == c) in_word = 1;
    }
    if (all_possible) return 1;
    if (all_not_here) return !in_word;
    for (int i = 0; i < NCH; i++) {
        if (k[i] == HERE && c != word[i]) return 0;
        if (k[i] == NOT_HERE && c == word[i]) return 0;
    }
    return in_word;
}

/* WordleState.word_in_state (game.py:76-80) */
static int word_in_state(uint8_t st[ALPHA][NCH], const char *word) {
    for (int c = 0; c < ALPHA; c++)
        if (!word_satisfies(st[c], (char)('a' + c), word)) return 0;
    return 1;
}

/* Vocabulary.__init__ filter (game.py:150-158) */
static void refilter(orc_wordle *g) {
    g->n_filtered = 0;
    for (int w = 0; w < g->V; w++)
        if (word_in_state(g->state, g->words + (size_t)w * NCH)) g->filtered[g->n_filtered++] = w;
}

static int in_vocab(const orc_wordle *g, const char *a, int len) {
    if (len != NCH) return 0;
    for (int w = 0; w < g->V; w++)
        if (memcmp(g->words + (size_t)w * NCH, a, NCH) == 0) return 1;
    return 0;
}

orc_wordle *orc_wordle_create(const char *words5, int V, int require_in_vocab, double bad_word_reward) {
    orc_wordle *g = (orc_wordle *)calloc(1, sizeof(orc_wordle));
    g->V = V;
    g->words = (char *)malloc((size_t)V * NCH);
    memcpy(g->words, words5, (size_t)V * NCH);
    g->filtered = (int *)malloc(sizeof(int) * (size_t)V);
    g->require_in_vocab = require_in_vocab;
    g->bad_word_reward = bad_word_reward;
    return g;
}

void orc_wordle_destroy(orc_wordle *g) {
    if (!g) return;
    free(g->words); free(g->filtered); free(g);
}

/* WordleEnvironment.reset (wordle/env/env.py:52-55): rng = Random(seed); initial state; filtered = all */
void orc_wordle_reset(orc_wordle *g, const uint32_t *key, int klen) {
    orc_mt_seed(&g->rng, key, klen);
    memset(g->state, POSSIBLE, sizeof(g->state));
    refilter(g);
    g->n_actions = 0;
    g->last_valid = 0;
}

/* WordleGame.reward (game.py:290-293) */
static double reward_of(const orc_wordle *g, int *is_int) {
    if (g->n_actions > 0 && (!g->last_len5 || !g->last_in_vocab)) { *is_int = 0; return g->bad_word_reward; }
    *is_int = 1;
    if (g->n_filtered == 1) {
        const char *w = g->words + (size_t)g->filtered[0] * NCH;
        for (int i = 0; i < g->n_actions && i < NTRIES + 2; i++)
            if (g->action_is_word[i] && memcmp(g->actions[i], w, NCH) == 0) return 0.0;
    }
    return -1.0;
}

/*
 * One env.step: WordleEnvironment.step (env.py:46-50) -> WordleGame.next (game.py:213-222)
 * -> transition_sequence()[-1] (game.py:273-288), reward (290-293), is_terminal (295-296).
 * `action`/`len` is the de-formatted action text (env.py:19-26: strip + remove spaces, done by the caller).
 * obs_out receives up to 5 symbols from {g,y,b}; *obs_len = 0 for an invalid action (empty transition string).
 */
void orc_wordle_step(orc_wordle *g, const char *action, int len, char *obs_out, int *obs_len,
                     double *reward, int *reward_is_int, int *done) {
    int alpha = 1;
    for (int i = 0; i < len; i++)
        if (action[i] < 'a' || action[i] > 'z') alpha = 0;
    int inv = in_vocab(g, action, len);
    int slot = g->n_actions < NTRIES + 2 ? g->n_actions : NTRIES + 1;
    g->action_is_word[slot] = (len == NCH);
    if (len == NCH) memcpy(g->actions[slot], action, NCH);
    g->last_len5 = (len == NCH);
    g->last_in_vocab = inv;
    /* game.py:214 */
    int invalid = (len != NCH) || !alpha || (g->require_in_vocab && !inv);
    g->n_actions++;
    if (invalid) {
        g->last_valid = 0;
        *obs_len = 0;
    } else {
        /* game.py:219-221; random.choice([]) raises IndexError in the reference: reported as *obs_len = -1 (the Python face raises) */
        if (g->n_filtered == 0) { *obs_len = -1; *reward = 0.0; *reward_is_int = 1; *done = 1; return; }
        uint32_t r = orc_mt_randbelow(&g->rng, (uint32_t)g->n_filtered);
        const char *target = g->words + (size_t)g->filtered[r] * NCH;
        /* WordleState.transition_state (game.py:82-92) */
        for (int i = 0; i < NCH; i++) {
            char c = action[i];
            int ci = c - 'a', in_t = 0;
            for (int j = 0; j < NCH; j++) if (target[j] == c) in_t = 1;
            if (c == target[i]) g->state[ci][i] = HERE;
            else if (in_t) g->state[ci][i] = NOT_HERE;
            else memset(g->state[ci], NOT_HERE, NCH);
        }
        refilter(g);
        g->last_valid = 1;
        /* transition_sequence for the last action (game.py:280-287) */
        int n = 0;
        for (int i = 0; i < NCH; i++) {
            int ci = action[i] - 'a', all_nh = 1;
            for (int j = 0; j < NCH; j++) if (g->state[ci][j] != NOT_HERE) all_nh = 0;
            if (g->state[ci][i] == HERE) obs_out[n++] = 'g';
            else if (all_nh) obs_out[n++] = 'b';
            else if (g->state[ci][i] == NOT_HERE) obs_out[n++] = 'y';
        }
        *obs_len = n;
    }
    *reward = reward_of(g, reward_is_int);
    *done = (g->n_actions == NTRIES) || (*reward == 0.0);
}

void orc_wordle_get_state(const orc_wordle *g, uint8_t *out130, int *n_filtered) {
    memcpy(out130, g->state, ALPHA * NCH);
    *n_filtered = g->n_filtered;
}

/*
 * cpu_baseline driver: N independent envs, one scripted guess per env per step (word index, or <0 = invalid
 * 5-letter non-word), `steps` steps with auto-reset on done.  Returns the number of env.step calls made.
 */
long orc_wordle_run(orc_wordle **envs, int N, const int32_t *guess_idx /* [steps][N] */, int steps) {
    long count = 0;
    char obs[8]; int ol, rint, done; double rew;
    for (int s = 0; s < steps; s++) {
        for (int e = 0; e < N; e++) {
            orc_wordle *g = envs[e];
            int gi = guess_idx[(size_t)s * N + e];
            if (gi >= 0) orc_wordle_step(g, g->words + (size_t)gi * NCH, NCH, obs, &ol, &rew, &rint, &done);
            else orc_wordle_step(g, "qqqqq", NCH, obs, &ol, &rew, &rint, &done);
            count++;
            if (done) { uint32_t key = (uint32_t)(e + 1000003u * (uint32_t)(s + 1)); orc_wordle_reset(g, &key, 1); }
        }
    }
    return count;
}

/*
 * The same driver on all host cores (OpenMP, one env per loop iteration: envs are independent — SURVEY.md section 8(d)'s "all-cores env baseline").
 * Every env runs its `steps` scripted steps back to back; same per-env results as orc_wordle_run.  `threads` <= 0: the OpenMP default.
 */
#ifdef _OPENMP
#include <omp.h>
#endif
long orc_wordle_run_mt(orc_wordle **envs, int N, const int32_t *guess_idx /* [steps][N] */, int steps, int threads, int *threads_used) {
    long count = 0;
    int used = 1;
#ifdef _OPENMP
    omp_set_num_threads(threads > 0 ? threads : omp_get_num_procs());
#pragma omp parallel
    {
#pragma omp single
        used = omp_get_num_threads();
    }
#else
    (void)threads;
#endif
#pragma omp parallel for schedule(static) reduction(+ : count)
    for (int e = 0; e < N; e++) {
        char obs[8]; int ol, rint, done; double rew;
        orc_wordle *g = envs[e];
        for (int s = 0; s < steps; s++) {
            int gi = guess_idx[(size_t)s * N + e];
            if (gi >= 0) orc_wordle_step(g, g->words + (size_t)gi * NCH, NCH, obs, &ol, &rew, &rint, &done);
            else orc_wordle_step(g, "qqqqq", NCH, obs, &ol, &rew, &rint, &done);
            count++;
            if (done) { uint32_t key = (uint32_t)(e + 1000003u * (uint32_t)(s + 1)); orc_wordle_reset(g, &key, 1); }
        }
    }
    if (threads_used) *threads_used = used;
    return count;
}
