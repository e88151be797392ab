// wordle_core.h — scalar Wordle MDP logic on a bit-mask knowledge state (host + device).
//
// Knowledge state (reference: 26 letters x 5 positions x {NOT_HERE, POSSIBLE, HERE},
// llm_rl_scripts/wordle/env/game.py:17-65) is held as 10 x 26-bit masks:
//   forb[i] bit c  <=>  K[c][i] == NOT_HERE
//   must[i] bit c  <=>  K[c][i] == HERE            (neither => POSSIBLE)
// CharState.word_satisfies (game.py:53-65) then reduces, for a word w = l0..l4, to
//   for all i:  !(forb[i] >> l_i & 1)                      NOT_HERE at i  => w[i] != c
//               (must[i] & ~bit(l_i)) == 0                 HERE at i      => w[i] == c
//   (req & ~letters(w)) == 0,  req = (OR_i forb[i]|must[i]) & ~(AND_i forb[i])
// where `req` are the letters that are neither all-POSSIBLE nor all-NOT_HERE ("c in word").
// The all-NOT_HERE special case (c not in word) is the per-position NOT_HERE rule applied 5 times.
#pragma once
#include <stdint.h>

#ifdef LMRL_HOST_ONLY
#define LMRL_HD inline
#else
#include <hip/hip_runtime.h>
#define LMRL_HD __host__ __device__ __forceinline__
#endif

namespace lmrl {

constexpr uint32_t kBadGuess = 0xFFFFFFFFu;
constexpr int kWordleTries = 6;   // game.py:15 N_TRIES
constexpr int kWordleStateWords = 19;

struct WordleMasks {
    uint32_t forb[5];
    uint32_t must[5];
    uint32_t req;  // derived
};

LMRL_HD uint32_t letter_at(uint32_t packed, int i) { return (packed >> (5 * i)) & 31u; }

LMRL_HD uint32_t letters_mask(uint32_t packed) {
    uint32_t m = 0;
    for (int i = 0; i < 5; i++) m |= 1u << letter_at(packed, i);
    return m;
}

LMRL_HD void wordle_derive(WordleMasks &s) {
    uint32_t any = 0, all = 0x3FFFFFFu;
    for (int i = 0; i < 5; i++) {
        any |= s.forb[i] | s.must[i];
        all &= s.forb[i];
    }
    s.req = any & ~all;
}

// WordleState.word_in_state (game.py:76-80)
LMRL_HD bool wordle_consistent(const WordleMasks &s, uint32_t packed, uint32_t wmask) {
    uint32_t bad = s.req & ~wmask;
    for (int i = 0; i < 5; i++) {
        uint32_t bit = 1u << letter_at(packed, i);
        bad |= s.forb[i] & bit;
        bad |= s.must[i] & ~bit;
    }
    return bad == 0;
}

// WordleState.transition_state (game.py:82-92): guess letters processed in order, later letters overwrite.
LMRL_HD void wordle_transition(WordleMasks &s, uint32_t guess, uint32_t target) {
    const uint32_t tmask = letters_mask(target);
    for (int i = 0; i < 5; i++) {
        const uint32_t c = letter_at(guess, i);
        const uint32_t bit = 1u << c;
        if (c == letter_at(target, i)) {  // correct_pos(i)
            s.must[i] |= bit;
            s.forb[i] &= ~bit;
        } else if (tmask & bit) {         // wrong_pos(i)
            s.forb[i] |= bit;
            s.must[i] &= ~bit;
        } else {                          // not_used()
            for (int k = 0; k < 5; k++) {
                s.forb[k] |= bit;
                s.must[k] &= ~bit;
            }
        }
    }
    wordle_derive(s);
}

// WordleGame.transition_sequence()[-1] (game.py:280-287) for a guess that went through the transition.
// Returns bits [3k,3k+3) = k-th emitted symbol (1 g / 2 y / 3 b), bits [16,19) = number of symbols.
LMRL_HD uint32_t wordle_obs(const WordleMasks &s, uint32_t guess) {
    uint32_t all_nh = 0x3FFFFFFu;
    for (int i = 0; i < 5; i++) all_nh &= s.forb[i];
    uint32_t out = 0, n = 0;
    for (int i = 0; i < 5; i++) {
        const uint32_t bit = 1u << letter_at(guess, i);
        uint32_t sym = 0;
        if (s.must[i] & bit) sym = 1;
        else if (all_nh & bit) sym = 3;
        else if (s.forb[i] & bit) sym = 2;
        if (sym) { out |= sym << (3 * n); n++; }
    }
    return out | (n << 16);
}

}  // namespace lmrl
