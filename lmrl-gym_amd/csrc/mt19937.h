// mt19937.h — CPython-compatible MT19937 (random.Random) as host+device inline functions.
//
// Replaces the stdlib generator the reference envs draw from:
//   llm_rl_scripts/wordle/env/env.py:53        vocab.rng = random.Random(seed)
//   llm_rl_scripts/wordle/env/game.py:178-179  rng.choice(filtered_vocab)
//   llm_rl_scripts/maze/env/env.py:187-212     random.seed(seed) / random.choice(...)
//
// One stream per env, stored struct-of-arrays in HBM: word k of stream e lives at mt[k*n + e]
// so that a wave stepping 64 neighbouring envs reads consecutive dwords; idx[e] follows at
// offset 624*n.  The functions are written against that layout through `MtRef` and compile
// for the host as well (LMRL_HOST_ONLY) so the CPU test harness can pin them against CPython.
#pragma once
#include <stdint.h>

#ifdef LMRL_HOST_ONLY
#define LMRL_HD inline
#else
#include <hip/hip_runtime.h>
#define LMRL_HD __host__ __device__ __forceinline__
#endif

namespace lmrl {

constexpr int kMtN = 624;
constexpr int kMtM = 397;

struct MtRef {
    uint32_t *mt;   // [624][n]
    uint32_t *idx;  // [n]
    int n;
    int e;
    LMRL_HD uint32_t get(int k) const { return mt[(size_t)k * n + e]; }
    LMRL_HD void set(int k, uint32_t v) const { mt[(size_t)k * n + e] = v; }
};

LMRL_HD MtRef mt_ref(void *buf, int n, int e) {
    MtRef r;
    r.mt = reinterpret_cast<uint32_t *>(buf);
    r.idx = r.mt + (size_t)kMtN * n;
    r.n = n;
    r.e = e;
    return r;
}

// init_genrand(19650218): identical for every stream, so it is tabulated once (table[624]).
inline void mt_init_table(uint32_t *table) {
    table[0] = 19650218u;
    for (int i = 1; i < kMtN; i++) table[i] = 1812433253u * (table[i - 1] ^ (table[i - 1] >> 30)) + (uint32_t)i;
}

// genrand "twist": regenerate all 624 words in place.
LMRL_HD void mt_twist(const MtRef &r) {
    uint32_t y;
    uint32_t cur = r.get(0);
    const uint32_t first = cur;
    for (int kk = 0; kk < kMtN - 1; kk++) {
        uint32_t nxt = r.get(kk + 1);
        y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
        int src = kk + kMtM < kMtN ? kk + kMtM : kk + kMtM - kMtN;
        r.set(kk, r.get(src) ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u));
        cur = nxt;
    }
    // kk = 623 pairs with the NEW mt[0]; `first` is the old one, so re-read.
    (void)first;
    y = (cur & 0x80000000u) | (r.get(0) & 0x7fffffffu);
    r.set(kMtN - 1, r.get(kMtM - 1) ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u));
}

// random.Random(seed): init_by_array(key) with key = 32-bit little-endian limbs of |seed|
// (1 limb if < 2^32, else 2 — uint64 seeds), then idx = 624 (CPython random_seed()).
LMRL_HD void mt_seed(const MtRef &r, uint64_t seed, const uint32_t *table) {
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    const int klen = key[1] ? 2 : 1;
    // first loop: max(624, klen) = 624 iterations, i = 1..623, wrap, then i = 1 once more
    uint32_t prev = table[0];
    uint32_t mt1 = 0;
    int j = 0;
    for (int i = 1; i < kMtN; i++) {
        uint32_t cur = (table[i] ^ ((prev ^ (prev >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        r.set(i, cur);
        if (i == 1) mt1 = cur;
        prev = cur;
        j = (j + 1 == klen) ? 0 : j + 1;
    }
    // wrap: mt[0] = mt[623]; i = 1 (624th iteration)
    mt1 = (mt1 ^ ((prev ^ (prev >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
    prev = mt1;
    // second loop: 623 iterations starting at i = 2
    for (int i = 2; i < kMtN; i++) {
        uint32_t cur = (r.get(i) ^ ((prev ^ (prev >> 30)) * 1566083941u)) - (uint32_t)i;
        r.set(i, cur);
        prev = cur;
    }
    // wrap: mt[0] = mt[623]; last iteration at i = 1
    mt1 = (mt1 ^ ((prev ^ (prev >> 30)) * 1566083941u)) - 1u;
    r.set(1, mt1);
    r.set(0, 0x80000000u);
    *(r.idx + r.e) = kMtN;
}

#ifndef LMRL_HOST_ONLY
// Seeding + first regeneration with the 624-word state held in LDS (32 streams per 64-thread workgroup use 78 KiB):
// the 1870 dependent steps then pay LDS latency (~64 cycles) instead of an HBM round trip each, and the final state is
// written to the struct-of-arrays HBM buffer with coalesced stores.  `lds` = 624*32 uint32 of dynamic shared memory.
constexpr int kMtSeedLanes = 32;
__device__ __forceinline__ void mt_seed_and_twist_lds(uint32_t *lds, void *mt_buf, int n, int e, bool do_it, uint64_t seed,
                                                      const uint32_t *table) {
    const int slot = threadIdx.x % kMtSeedLanes;
    if (do_it) {
        MtRef l;
        l.mt = lds; l.idx = lds; l.n = kMtSeedLanes; l.e = slot;       // idx unused below
        const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
        const int klen = key[1] ? 2 : 1;
        uint32_t prev = table[0], mt1 = 0;
        int j = 0;
        for (int i = 1; i < kMtN; i++) {
            const uint32_t cur = (table[i] ^ ((prev ^ (prev >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
            l.set(i, cur);
            if (i == 1) mt1 = cur;
            prev = cur;
            j = (j + 1 == klen) ? 0 : j + 1;
        }
        mt1 = (mt1 ^ ((prev ^ (prev >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        prev = mt1;
        for (int i = 2; i < kMtN; i++) {
            const uint32_t cur = (l.get(i) ^ ((prev ^ (prev >> 30)) * 1566083941u)) - (uint32_t)i;
            l.set(i, cur);
            prev = cur;
        }
        mt1 = (mt1 ^ ((prev ^ (prev >> 30)) * 1566083941u)) - 1u;
        l.set(1, mt1);
        l.set(0, 0x80000000u);
        mt_twist(l);
        MtRef g = mt_ref(mt_buf, n, e);
        for (int k = 0; k < kMtN; k++) g.set(k, l.get(k));
        g.idx[e] = 0;
    }
}
#endif

LMRL_HD uint32_t mt_next(const MtRef &r) {
    uint32_t i = r.idx[r.e];
    if (i >= (uint32_t)kMtN) {
        mt_twist(r);
        i = 0;
    }
    uint32_t y = r.get((int)i);
    r.idx[r.e] = i + 1;
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// random.Random._randbelow_with_getrandbits(n): k = n.bit_length(); getrandbits(k) until < n.
LMRL_HD uint32_t mt_randbelow(const MtRef &r, uint32_t n) {
    int k = 0;
    for (uint32_t t = n; t; t >>= 1) k++;
    uint32_t v = mt_next(r) >> (32 - k);
    while (v >= n) v = mt_next(r) >> (32 - k);
    return v;
}

}  // namespace lmrl
