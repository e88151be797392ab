// Host build of the kernels' scalar cores (mt19937.h, wordle_core.h) so the CPU-only test tier can
// pin the bit-mask formulation against the golden traces without a GPU.  Control flow mirrors
// wordle_step_kernel with the 64-lane sweeps replaced by plain loops.
#define LMRL_HOST_ONLY 1
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../lmrl-gym_amd/csrc/mt19937.h"
#include "../lmrl-gym_amd/csrc/wordle_core.h"

using namespace lmrl;

extern "C" {

void host_mt_stream(uint64_t seed, int n_out, uint32_t *out) {
    uint32_t table[kMtN];
    mt_init_table(table);
    uint32_t *buf = (uint32_t *)calloc(kMtN + 1, sizeof(uint32_t));
    MtRef r = mt_ref(buf, 1, 0);
    mt_seed(r, seed, table);
    for (int k = 0; k < n_out; k++) out[k] = mt_next(r);
    free(buf);
}

void host_mt_randbelow(uint64_t seed, const uint32_t *bounds, int n, uint32_t *out) {
    uint32_t table[kMtN];
    mt_init_table(table);
    uint32_t *buf = (uint32_t *)calloc(kMtN + 1, sizeof(uint32_t));
    MtRef r = mt_ref(buf, 1, 0);
    mt_seed(r, seed, table);
    for (int k = 0; k < n; k++) out[k] = mt_randbelow(r, bounds[k]);
    free(buf);
}

struct HostWordle {
    int V, require;
    float bad;
    uint32_t *words, *wmask;
    WordleMasks s;
    uint32_t nfilt, nact, hist[6];
    uint32_t *mt;
};

HostWordle *host_wordle_create(const char *words5, int V, int require, float bad) {
    HostWordle *h = (HostWordle *)calloc(1, sizeof(HostWordle));
    h->V = V; h->require = require; h->bad = bad;
    h->words = (uint32_t *)malloc(4 * V); h->wmask = (uint32_t *)malloc(4 * V);
    for (int w = 0; w < V; w++) {
        uint32_t p = 0;
        for (int i = 0; i < 5; i++) p |= (uint32_t)(words5[w * 5 + i] - 'a') << (5 * i);
        h->words[w] = p; h->wmask[w] = letters_mask(p);
    }
    h->mt = (uint32_t *)calloc(kMtN + 1, 4);
    return h;
}
void host_wordle_destroy(HostWordle *h) { free(h->words); free(h->wmask); free(h->mt); free(h); }

void host_wordle_reset(HostWordle *h, uint64_t seed) {
    uint32_t table[kMtN];
    mt_init_table(table);
    memset(&h->s, 0, sizeof(h->s));
    h->nfilt = h->V; h->nact = 0;
    for (int k = 0; k < 6; k++) h->hist[k] = kBadGuess;
    MtRef r = mt_ref(h->mt, 1, 0);
    mt_seed(r, seed, table);
    mt_twist(r);
    r.idx[0] = 0;
}

void host_wordle_step(HostWordle *h, uint32_t g, uint32_t *obs, float *reward, uint8_t *flags) {
    WordleMasks &s = h->s;
    wordle_derive(s);
    const bool shaped = g != kBadGuess;
    bool member = false;
    if (shaped) for (int i = 0; i < h->V; i++) if (h->words[i] == g) { member = true; break; }
    const bool valid = shaped && (member || !h->require) && h->nfilt > 0;
    const bool bad_word = !(shaped && member);
    uint32_t new_nfilt = h->nfilt, o = 0, uniq = kBadGuess;
    if (valid) {
        uint32_t r = mt_randbelow(mt_ref(h->mt, 1, 0), h->nfilt);
        uint32_t cnt = 0, target = g;
        for (int i = 0; i < h->V; i++)
            if (wordle_consistent(s, h->words[i], h->wmask[i])) { if (cnt == r) { target = h->words[i]; break; } cnt++; }
        wordle_transition(s, g, target);
        cnt = 0;
        for (int i = 0; i < h->V; i++)
            if (wordle_consistent(s, h->words[i], h->wmask[i])) { if (cnt == 0) uniq = h->words[i]; cnt++; }
        new_nfilt = cnt;
        o = wordle_obs(s, g);
    }
    if (h->nact < 6) h->hist[h->nact] = g;
    float rew;
    if (bad_word) rew = h->bad;
    else {
        bool win = false;
        if (new_nfilt == 1) { win = uniq == g; for (uint32_t k = 0; k < h->nact && k < 6; k++) win |= h->hist[k] == uniq; }
        rew = win ? 0.f : -1.f;
    }
    h->nact++;
    const bool done = h->nact == 6 || rew == 0.f;
    h->nfilt = new_nfilt;
    *obs = o; *reward = rew;
    *flags = (uint8_t)((done ? 1 : 0) | (valid ? 2 : 0) | (bad_word ? 4 : 0));
}

void host_wordle_trits(HostWordle *h, uint8_t *out, uint32_t *nfilt) {
    for (int c = 0; c < 26; c++)
        for (int i = 0; i < 5; i++)
            out[c * 5 + i] = (h->s.must[i] >> c & 1u) ? 2 : ((h->s.forb[i] >> c & 1u) ? 0 : 1);
    *nfilt = h->nfilt;
}
}
