"""CPU tier: the `jax.random` restatements behind the sampler's LMRL_RNG_JAX mode.

Three independent implementations must agree and must reproduce the PUBLISHED known-answer vectors of Threefry-2x32 (20 rounds) from the
Random123 distribution (kat_vectors; the same three that jax's own tests/random_test.py::testThreefry2x32 asserts):
  * oracle/jax_random.py          numpy, vectorised (test infrastructure)
  * lmrl_gym_amd/jax_prng.py      Python ints (the host key schedule of the product path)
  * csrc/threefry.h               the device code, through its host faces in liblmrl_amd.so (no GPU needed)
jax itself is absent (third-party, jax==0.4.7 per the reference's requirements.txt): the key schedule is "unverified vs the JAX path"."""
import ctypes

import numpy as np

import lmrl_gym_amd  # noqa: F401
from lmrl_gym_amd import _lib, jax_prng as JP
from oracle import jax_random as JR

KAT = [((0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6B200159, 0x99BA4EFE)),
       ((0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF), (0x1CB996FC, 0xBB002BE7)),
       ((0x13198A2E, 0x03707344), (0x243F6A88, 0x85A308D3), (0xC4923A9C, 0x483DF7A0))]      # (key, counter, expected)


def _so_block(key, ctr):
    L = _lib.lib()
    k, c, o = (ctypes.c_uint32 * 2)(*key), (ctypes.c_uint32 * 2)(*ctr), (ctypes.c_uint32 * 2)()
    L.lmrl_threefry2x32(k, c, o)
    return o[0], o[1]


def test_threefry2x32_known_answer_vectors():
    for key, ctr, exp in KAT:
        assert tuple(int(x) for x in JR.threefry_2x32(np.array(key, np.uint32), np.array(ctr, np.uint32))) == exp
        assert JP.threefry2x32(key, ctr) == exp
        assert _so_block(key, ctr) == exp


def test_random_bits_layout_and_split_agree_across_implementations():
    L = _lib.lib()
    rng = np.random.RandomState(0)
    for n in (1, 2, 3, 8, 9, 1000, 1001):
        key = tuple(int(x) for x in rng.randint(0, 2 ** 32, size=2, dtype=np.uint64))
        ref = JR.random_bits(np.array(key, np.uint32), (n,))
        assert JP.random_bits(key, n) == [int(x) for x in ref]
        out = (ctypes.c_uint32 * n)()
        assert L.lmrl_jax_random_bits_host((ctypes.c_uint32 * 2)(*key), n, 0, n, out) == 0
        assert list(out) == [int(x) for x in ref]
        # the halves structure itself: word i < h comes from block (i, i + h), word i >= h from block (i - h, i)
        h = (n + 1) // 2
        for i in (0, n // 2, n - 1):
            blk = JP.threefry2x32(key, (i, i + h if i + h < n else 0)) if i < h else JP.threefry2x32(key, (i - h, i))
            assert int(ref[i]) == (blk[0] if i < h else blk[1])
    key = JP.prng_key(0)
    assert key == (0, 0) and JP.prng_key(42) == (0, 42) and JP.prng_key(-1) == (0, 0xFFFFFFFF)
    assert [tuple(int(x) for x in k) for k in JR.split(JR.prng_key(7), 3)] == JP.split(JP.prng_key(7), 3)
    # split(key) = the four words of threefry_2x32(key, [0, 1, 2, 3]) = blocks (0, 2) and (1, 3)
    a, b = JP.threefry2x32((0, 7), (0, 2)), JP.threefry2x32((0, 7), (1, 3))
    assert JP.split((0, 7)) == [(a[0], b[0]), (a[1], b[1])]
    # HF-Flax per-token schedule: sample with the first half, carry the second
    sk = JP.SampleKeys(JP.prng_key(3))
    assert [sk.next() for _ in range(4)] == [tuple(int(x) for x in k) for k in JR.hf_flax_sample_keys(JR.prng_key(3), 4)]


def test_uniform_gumbel_categorical_properties():
    key = JR.prng_key(11)
    u = JR.uniform(key, (200000,), minval=np.finfo(np.float32).tiny, maxval=1.0)
    assert u.dtype == np.float32 and u.min() >= np.finfo(np.float32).tiny and u.max() < 1.0
    assert abs(float(u.mean()) - 0.5) < 5e-3
    g = JR.gumbel(key, (200000,))
    assert np.isfinite(g).all() and abs(float(g.mean()) - 0.5772) < 1e-2
    # categorical draws follow softmax(logits): chi-square on 5 classes
    logits = np.log(np.array([0.5, 0.2, 0.15, 0.1, 0.05], dtype=np.float32))
    draws = JR.categorical(JR.prng_key(5), np.broadcast_to(logits, (60000, 5)))
    freq = np.bincount(draws, minlength=5) / 60000.0
    assert np.abs(freq - np.exp(logits)).max() < 8e-3
