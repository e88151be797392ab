"""Host side of the sampler's `LMRL_RNG_JAX` mode: the KEY SCHEDULE of the reference's sampling path, in plain Python integers.

The reference draws tokens with `jax.random.categorical` under keys derived by `jax.random.split`:
  * `GPT2PPOPolicy.__init__(prng_key=...)`, then per `act()`:  `self.prng_key, new_key = jax.random.split(self.prng_key)`
    (LLM_RL/algorithms/ppo/gpt2/interface.py:524-526; value_rl_base/gpt2/interface.py:298-300); `new_key` goes to `generate`;
  * HF-Flax `_sample` (transformers 4.26.1 `generation/flax_utils.py`, 3rd party), per generated token:
    `prng_key, prng_key_next = jax.random.split(state.prng_key)`; `next_token = jax.random.categorical(prng_key, logits)`; the state
    keeps `prng_key_next`.
jax (0.4.7, default threefry PRNG, `jax_threefry_partitionable=False`) is absent from the tree and the image: this restates its
published algorithm (jax/_src/prng.py `threefry_seed`, `threefry_split`) — keys are two uint32 words, `split(key, n)` =
`threefry_2x32(key, iota(2n))` reshaped (n, 2).  The block function is pinned by the Random123 known-answer vectors
(tests/test_jax_prng.py); whether JaxSeq's `generate` wrapper splits the key once more before HF's loop cannot be checked here
("unverified vs the JAX path").  The per-element noise itself is generated on the DEVICE (csrc/threefry.h).
"""
from __future__ import annotations

from typing import List, Tuple

Key = Tuple[int, int]
_M = 0xFFFFFFFF


def _rotl(x: int, r: int) -> int:
    return ((x << r) | (x >> (32 - r))) & _M


def threefry2x32(key: Key, ctr: Tuple[int, int]) -> Tuple[int, int]:
    """Threefry-2x32, 20 rounds (Salmon et al. 2011) — jax/_src/prng.py `_threefry2x32_lowering` / `threefry2x32_p`."""
    k0, k1 = key[0] & _M, key[1] & _M
    ks = (k0, k1, k0 ^ k1 ^ 0x1BD11BDA)
    x0, x1 = (ctr[0] + ks[0]) & _M, (ctr[1] + ks[1]) & _M
    rot = ((13, 15, 26, 6), (17, 29, 16, 24))
    for g in range(5):
        for r in rot[g % 2]:
            x0 = (x0 + x1) & _M
            x1 = _rotl(x1, r) ^ x0
        x0 = (x0 + ks[(g + 1) % 3]) & _M
        x1 = (x1 + ks[(g + 2) % 3] + g + 1) & _M
    return x0, x1


def random_bits(key: Key, n: int) -> List[int]:
    """jax.random.bits(key, (n,), uint32) — `threefry_random_bits` for bit_width 32: counts = iota(n) (+ one 0 when n is odd) split in
    halves, blocks (x0[j], x1[j]), outputs concatenated."""
    h = (n + 1) // 2
    cnt = list(range(n)) + ([0] if n % 2 else [])
    ys = [threefry2x32(key, (cnt[j], cnt[j + h])) for j in range(h)]
    return ([y[0] for y in ys] + [y[1] for y in ys])[:n]


def prng_key(seed: int) -> Key:
    """jax.random.PRNGKey(seed) with 32-bit ints (`threefry_seed`, x64 disabled): key = [0, seed mod 2^32]; a seed wider than 32 bits
    (x64 enabled) gives [seed >> 32, seed & 0xffffffff]."""
    seed = int(seed)
    return ((seed >> 32) & _M if seed >= 0 else 0, seed & _M)


def split(key: Key, num: int = 2) -> List[Key]:
    """jax.random.split(key, num) — `threefry_split`: threefry_2x32(key, iota(2 * num)).reshape(num, 2)."""
    w = random_bits(key, 2 * num)
    return [(w[2 * i], w[2 * i + 1]) for i in range(num)]


def key_to_seed(key: Key) -> int:
    """The 64-bit `lmrl_sample_params.seed` carrying a key: (key[0] << 32) | key[1]."""
    return ((key[0] & _M) << 32) | (key[1] & _M)


class SampleKeys:
    """Key state of one `generate` call: `next()` returns the key of the next token's `categorical` and advances the carried key
    (HF-Flax `_sample`: `prng_key, prng_key_next = split(state.prng_key)`; sample with the first, carry the second)."""

    def __init__(self, key: Key):
        self.key = key

    def next(self) -> Key:
        k_sample, self.key = split(self.key)
        return k_sample
