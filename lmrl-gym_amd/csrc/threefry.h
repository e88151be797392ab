// threefry.h — the random stream of `jax.random.categorical` (jax 0.4.7, default non-partitionable threefry PRNG), for the sampler's
// parity mode LMRL_RNG_JAX.
//
// The reference samples with HF-Flax `_sample` (transformers 4.26.1, 3rd party): per token `jax.random.split(key)` then
// `jax.random.categorical(key, logits[B, V])`, reached through GPT2PPOPolicy.act (LLM_RL/algorithms/ppo/gpt2/interface.py:524-535) /
// GPT2ValuePolicy.act (value_rl_base/gpt2/interface.py:298-310).  jax / jaxlib are not in the tree and not installable here; what
// follows restates their PUBLISHED algorithm (jax/_src/prng.py: threefry_2x32, threefry_split, threefry_random_bits; jax/_src/random.py:
// _uniform, gumbel, categorical):
//   * block function: Threefry-2x32, 20 rounds (Salmon et al., SC'11; pinned by the Random123 known-answer vectors in the tests);
//   * random_bits(key, n words): counts = iota(n) (one 0 appended when n is odd), split into halves x0 | x1 of h = ceil(n/2) words,
//     (y0, y1) = threefry(key, (x0[j], x1[j])), bits = concat(y0, y1)[:n]  =>  word i is  y0 of block (i, i + h)  for i < h  and
//     y1 of block (i - h, i)  for i >= h  (a counter i + h >= n is the appended 0);
//   * uniform(minval = tiny, maxval = 1): f = bitcast((bits >> 9) | 0x3F800000) - 1;  max(tiny, f * (1 - tiny) + tiny)  in fp32,
//     i.e. tiny for f == 0 and f otherwise;  gumbel = -log(-log(u));  categorical = argmax(logits + gumbel) over the last axis, with the
//     noise array shaped like `logits` (word index i = row * V + column, n = B * V).
// XLA's log is not bit-pinned (its polynomial differs per backend): draws agree except where two perturbed scores are within a few ulp.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define LMRL_TF_FN __host__ __device__ __forceinline__
#else
#define LMRL_TF_FN static inline
#endif

namespace lmrl {

LMRL_TF_FN uint32_t tf_rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

LMRL_TF_FN void threefry2x32_20(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t &o0, uint32_t &o1) {
    const uint32_t ks0 = k0, ks1 = k1, ks2 = k0 ^ k1 ^ 0x1BD11BDAu;
    uint32_t x0 = c0 + ks0, x1 = c1 + ks1;
#define LMRL_TF_R(r) x0 += x1; x1 = tf_rotl(x1, r); x1 ^= x0;
    LMRL_TF_R(13) LMRL_TF_R(15) LMRL_TF_R(26) LMRL_TF_R(6)
    x0 += ks1; x1 += ks2 + 1u;
    LMRL_TF_R(17) LMRL_TF_R(29) LMRL_TF_R(16) LMRL_TF_R(24)
    x0 += ks2; x1 += ks0 + 2u;
    LMRL_TF_R(13) LMRL_TF_R(15) LMRL_TF_R(26) LMRL_TF_R(6)
    x0 += ks0; x1 += ks1 + 3u;
    LMRL_TF_R(17) LMRL_TF_R(29) LMRL_TF_R(16) LMRL_TF_R(24)
    x0 += ks1; x1 += ks2 + 4u;
    LMRL_TF_R(13) LMRL_TF_R(15) LMRL_TF_R(26) LMRL_TF_R(6)
    x0 += ks2; x1 += ks0 + 5u;
#undef LMRL_TF_R
    o0 = x0; o1 = x1;
}

// word i of jax.random.bits(key, (n,), uint32)
LMRL_TF_FN uint32_t jax_random_word(uint32_t k0, uint32_t k1, uint32_t i, uint32_t n) {
    const uint32_t h = (n + 1u) >> 1;
    uint32_t y0, y1;
    if (i < h) {
        const uint32_t c1 = (i + h < n) ? i + h : 0u;
        threefry2x32_20(k0, k1, i, c1, y0, y1);
        return y0;
    }
    threefry2x32_20(k0, k1, i - h, i, y0, y1);
    return y1;
}

// jax.random.uniform(key, minval = finfo(float32).tiny, maxval = 1) for one word of random bits
LMRL_TF_FN float jax_uniform_open(uint32_t bits) {
    union { uint32_t u; float f; } v;
    v.u = (bits >> 9) | 0x3F800000u;
    const float f = v.f - 1.0f;
    const float tiny = 1.17549435e-38f;
    const float u = f * (1.0f - tiny) + tiny;          // fp32: (1 - tiny) == 1, f + tiny == f for f >= 2^-23
    return u > tiny ? u : tiny;
}

}  // namespace lmrl
