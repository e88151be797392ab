"""TEST INFRASTRUCTURE — numpy restatement of the slice of jax.random the reference's sampling step uses.

jax==0.4.7 (requirements.txt:19-20 of the reference) is a third-party dependency absent from /root/reference and from this image, so
this follows its PUBLISHED algorithm: jax/_src/prng.py (`threefry_2x32`, `threefry_split`, `threefry_random_bits`, `threefry_seed`) and
jax/_src/random.py (`_uniform`, `gumbel`, `categorical`), default implementation ("threefry2x32", non-partitionable).
Pinned against: the Random123 Threefry-2x32-20 known-answer vectors (the same three jax's own tests/random_test.py::testThreefry2x32
checks) — tests/test_jax_prng.py.  NOT pinned: outputs of jax itself (parity unpinned vs the JAX path), XLA's float
log (its ulp-level rounding differs per backend; here numpy float32 log).
"""
from __future__ import annotations

import numpy as np

_U = np.uint32


def _rotl(x, r):
    return (x << _U(r)) | (x >> _U(32 - r))


def threefry_2x32(key, count):
    """jax/_src/prng.py::threefry_2x32(keypair, count): count (uint32 array, any length) is split in two halves (padded with one 0 when
    odd), Threefry-2x32-20 is applied to the pairs, outputs are concatenated and cut back to the input length."""
    key = np.asarray(key, dtype=_U)
    count = np.asarray(count, dtype=_U).ravel()
    odd = count.size % 2
    if odd:
        count = np.concatenate([count, np.zeros(1, _U)])
    x0, x1 = np.split(count.copy(), 2)
    ks = [key[0], key[1], key[0] ^ key[1] ^ _U(0x1BD11BDA)]
    rot = [(13, 15, 26, 6), (17, 29, 16, 24)]
    with np.errstate(over="ignore"):
        x0 = x0 + ks[0]; x1 = x1 + ks[1]
        for g in range(5):
            for r in rot[g % 2]:
                x0 = x0 + x1
                x1 = _rotl(x1, r)
                x1 = x1 ^ x0
            x0 = x0 + ks[(g + 1) % 3]
            x1 = x1 + ks[(g + 2) % 3] + _U(g + 1)
    out = np.concatenate([x0, x1])
    return out[:-1] if odd else out


def prng_key(seed: int):
    """threefry_seed with x64 disabled: [0, seed mod 2^32]."""
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF if seed >= 0 else 0, seed & 0xFFFFFFFF], dtype=_U)


def split(key, num: int = 2):
    return threefry_2x32(key, np.arange(num * 2, dtype=_U)).reshape(num, 2)


def random_bits(key, shape):
    size = int(np.prod(shape))
    return threefry_2x32(key, np.arange(size, dtype=_U)).reshape(shape)


def uniform(key, shape, minval=0.0, maxval=1.0):
    """_uniform for float32: 23 mantissa bits | exponent of 1.0, minus 1, scaled, clamped below."""
    bits = random_bits(key, shape)
    floats = ((bits >> _U(9)) | _U(0x3F800000)).view(np.float32) - np.float32(1.0)
    minval, maxval = np.float32(minval), np.float32(maxval)
    return np.maximum(minval, floats * (maxval - minval) + minval).astype(np.float32)


def gumbel(key, shape):
    u = uniform(key, shape, minval=np.finfo(np.float32).tiny, maxval=1.0)
    return (-np.log(-np.log(u))).astype(np.float32)


def categorical(key, logits):
    """jax.random.categorical(key, logits, axis=-1): argmax(logits + gumbel(key, logits.shape))."""
    logits = np.asarray(logits, dtype=np.float32)
    return np.argmax(gumbel(key, logits.shape) + logits, axis=-1)


def hf_flax_sample_keys(key, n_tokens: int):
    """Keys HF-Flax `_sample` hands to `categorical`, token by token (transformers 4.26.1 generation/flax_utils.py: per step
    `prng_key, prng_key_next = jax.random.split(state.prng_key)`; sample with prng_key, carry prng_key_next)."""
    out = []
    for _ in range(n_tokens):
        k, key = split(key)
        out.append(k)
    return out
