"""Unit tests of tests/golden/_jnp_shim.py: every stand-in the reference's loss functions reach is checked against plain numpy
semantics (and the JAX rules it must reproduce: x64-off promotion, functional `.at[].set`, padded `argwhere`, zero rows of
`one_hot`, masked min/max/std, optax 0.1.3 losses), plus the complex-step derivative mode against central differences."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
import _jnp_shim as S  # noqa: E402

RNG = np.random.RandomState(0)
X = RNG.randn(4, 7).astype(np.float32)
Y = RNG.randn(4, 7).astype(np.float32)


def a(x):
    return S.asarray(x)


def test_promotion_is_jax_x64_off():
    i = a(np.arange(7, dtype=np.int64))
    assert i.dtype == np.int32
    assert (a(X) * i).dtype == np.float32 and (a(X) * a(np.ones(7, np.int32))).dtype == np.float32       # f32 * i32 -> f32 (numpy: f64)
    assert (a(X) * a(X > 0)).dtype == np.float32 and (a(X) * 2.5).dtype == np.float32 and (a(X) / 3).dtype == np.float32
    assert a(np.zeros(3, np.float64)).dtype == np.float32
    assert (a(X > 0).sum(axis=1)).dtype == np.int32 and (a(X > 0).astype(S.float32)).dtype == np.float32
    assert isinstance(a(X).sum(), S.Arr) and a(X).sum().dtype == np.float32 and a(X).size == 28
    np.testing.assert_array_equal(np.asarray(a(X) * i), X * np.arange(7, dtype=np.float32))


def test_elementwise_and_reductions_match_numpy():
    np.testing.assert_array_equal(S.clip(a(X), a(Y) - 0.2, a(Y) + 0.2), np.clip(X, Y - np.float32(0.2), Y + np.float32(0.2)))
    np.testing.assert_array_equal(S.clip(a(X), 0.8, 1.2), np.clip(X, np.float32(0.8), np.float32(1.2)))
    np.testing.assert_array_equal(S.maximum(a(X), a(Y)), np.maximum(X, Y)); np.testing.assert_array_equal(S.minimum(a(X), a(Y)), np.minimum(X, Y))
    np.testing.assert_array_equal(S.exp(a(X)), np.exp(X)); np.testing.assert_array_equal(S.sqrt(a(X * X)), np.sqrt(X * X))
    np.testing.assert_allclose(S.reciprocal(a(X)), 1 / X, rtol=1e-7)
    np.testing.assert_array_equal(S.sum_(a(X)), X.sum()); np.testing.assert_array_equal(S.sum_(a(X), axis=1), X.sum(axis=1))
    np.testing.assert_allclose(S.mean(a(X)), X.mean(), rtol=1e-6); np.testing.assert_allclose(S.var(a(X)), X.var(), rtol=1e-6)
    np.testing.assert_array_equal(S.argmax(a(X).astype(S.int32), axis=1), np.argmax(X.astype(np.int32), axis=1))
    np.testing.assert_array_equal(S.concatenate((a(X), a(Y[:, :1])), axis=1), np.concatenate((X, Y[:, :1]), axis=1))
    np.testing.assert_array_equal(S.cumprod(S.full((5,), 0.9, dtype=S.float32), axis=0), np.cumprod(np.full(5, 0.9, np.float32)))
    np.testing.assert_array_equal(S.triu(a(X)), np.triu(X)); np.testing.assert_array_equal(S.expand_dims(a(X), 0), X[None])
    np.testing.assert_array_equal(S.arange(0, 4, dtype=S.int32), np.arange(4, dtype=np.int32))
    np.testing.assert_array_equal(S.ones((2, 3), dtype=S.int32), np.ones((2, 3), np.int32))
    np.testing.assert_array_equal(a(X)[..., None], X[..., None]); np.testing.assert_array_equal(a(X).reshape(-1), X.reshape(-1))
    np.testing.assert_array_equal(S.where(1 - a(np.array([1, 1, 0, 1, 0], np.int32)))[0], [2, 4])
    np.testing.assert_array_equal(S.take_along_axis(a(X), a(np.array([[1], [0], [6], [3]], np.int32)), axis=1), np.take_along_axis(X, np.array([[1], [0], [6], [3]]), 1))
    np.testing.assert_array_equal(S.flip(a(X), axis=1), X[:, ::-1])


def test_masked_min_max_std():
    m = X > 0.1
    assert float(S.min(a(X), where=a(m), initial=float("inf"))) == X[m].min() and float(S.max(a(X), where=a(m), initial=float("-inf"))) == X[m].max()
    np.testing.assert_allclose(S.std(a(X), where=a(m)), X[m].std(), rtol=1e-6)
    none = np.zeros_like(m)
    assert float(S.min(a(X), where=a(none), initial=float("inf"))) == float("inf") and float(S.max(a(X), where=a(none), initial=float("-inf"))) == float("-inf")
    with np.errstate(all="ignore"):
        assert np.isnan(float(S.std(a(X), where=a(none))))
    np.testing.assert_array_equal(S.max(a(X), axis=-1, keepdims=True), X.max(axis=-1, keepdims=True))


def test_at_set_is_functional_and_argwhere_one_hot_follow_jax():
    m = a(np.array([[0, 1, 1, 0], [1, 0, 0, 1], [0, 0, 0, 0]], dtype=bool))
    first = S.argmax(m.astype(S.int32), axis=1)
    m2 = m.at[S.arange(0, 3, dtype=S.int32), first].set(False)
    np.testing.assert_array_equal(m2, [[0, 0, 1, 0], [0, 0, 0, 1], [0, 0, 0, 0]])
    np.testing.assert_array_equal(m, [[0, 1, 1, 0], [1, 0, 0, 1], [0, 0, 0, 0]])           # the source is untouched
    flat = m.reshape(-1)
    idx = S.argwhere(flat, size=flat.shape[0], fill_value=flat.shape[0])[:, 0]
    np.testing.assert_array_equal(idx, [1, 2, 4, 7] + [12] * 8)                             # padded with fill_value to `size`
    oh = S.one_hot(idx, num_classes=13, dtype=S.float32)[:, :-1]
    assert oh.shape == (12, 12) and oh.dtype == np.float32
    np.testing.assert_array_equal(np.asarray(oh).sum(1), [1, 1, 1, 1] + [0] * 8)
    np.testing.assert_array_equal(S.one_hot(a(np.array([0, 5, -1], np.int32)), 3), [[1, 0, 0], [0, 0, 0], [0, 0, 0]])   # out of range -> zero row


def test_optax_losses():
    np.testing.assert_allclose(S.l2_loss(a(X), a(Y)), 0.5 * (X - Y) ** 2, rtol=1e-6)
    lg = RNG.randn(3, 5, 11).astype(np.float32) * 3
    lb = RNG.randint(0, 11, size=(3, 5)).astype(np.int32)
    z = lg.astype(np.float64)
    ref = np.log(np.exp(z - z.max(-1, keepdims=True)).sum(-1)) + z.max(-1) - np.take_along_axis(z, lb[..., None].astype(np.int64), -1)[..., 0]
    got = S.softmax_cross_entropy_with_integer_labels(a(lg), a(lb))
    assert got.dtype == np.float32
    np.testing.assert_allclose(got, ref, rtol=2e-6, atol=2e-6)
    with pytest.raises(AssertionError):
        S.softmax_cross_entropy_with_integer_labels(a(lg), a(lb.astype(np.float32)))
    np.testing.assert_array_equal(S.stop_gradient(a(X)), X)


def test_complex_step_mode_gives_directional_derivatives_and_honours_stop_gradient():
    """f(x) = sum(max(x*y, clip(x, -0.5, 0.5)) * exp(stop_gradient(x)) + 0.5*l2(x, sg(y))) — derivative vs central differences of the
    same function in float64 with the stop_gradient arguments frozen."""
    x0 = RNG.randn(5, 6); y0 = RNG.randn(5, 6); d = RNG.randn(5, 6)

    def f(x, frozen):
        return S.sum_(S.maximum(x * a(y0), S.clip(x, -0.5, 0.5)) * S.exp(frozen) + S.l2_loss(x, S.stop_gradient(a(y0))))

    S.complex_step(True)
    try:
        xc = a(x0 + 1j * 1e-30 * d)
        val = f(xc, S.stop_gradient(xc))
        deriv = float(np.asarray(val).imag / 1e-30)
        assert (xc > 0).dtype == bool and S.stop_gradient(xc).imag.max() == 0.0
    finally:
        S.complex_step(False)

    def f64(x):
        return float((np.maximum(x * y0, np.clip(x, -0.5, 0.5)) * np.exp(x0) + 0.5 * (x - y0) ** 2).sum())
    h = 1e-6
    fd = (f64(x0 + h * d) - f64(x0 - h * d)) / (2 * h)
    assert abs(deriv - fd) <= 1e-6 * max(1.0, abs(fd)), (deriv, fd)
    with pytest.raises(TypeError):
        a(np.zeros(2, dtype=np.complex128))                 # complex arrays only exist in derivative mode


def test_round3_stand_ins_squeeze_matmul_cond_and_optax_updates():
    """The stand-ins added for executing the reference's train-step closures (tests/golden/make_step_fixtures.py): jnp.squeeze / matmul,
    jax.lax.cond with a concrete predicate, optax.incremental_update (Polyak: step_size * new + (1 - step_size) * old per leaf) and
    optax.periodic_update (new iff steps % period == 0) on nested dict pytrees."""
    a = S.asarray(np.arange(6, dtype=np.float32).reshape(2, 3, 1))
    assert S.squeeze(a, axis=-1).shape == (2, 3) and isinstance(S.squeeze(a, axis=-1), S.Arr)
    x, w = S.asarray(np.ones((2, 3), np.float32)), S.asarray(np.arange(12, dtype=np.float32).reshape(3, 4))
    np.testing.assert_array_equal(np.asarray(S.matmul(x, w)), np.ones((2, 3), np.float32) @ np.arange(12, dtype=np.float32).reshape(3, 4))
    assert S.cond(np.bool_(True), lambda p, q: p + q, lambda p, q: p - q, 5, 3) == 8 and S.cond(False, lambda p: p, lambda p: -p, 2) == -2
    new = {"l": {"k": np.array([1.0, 2.0], np.float32)}, "b": np.array([4.0], np.float32)}
    old = {"l": {"k": np.array([3.0, 0.0], np.float32)}, "b": np.array([0.0], np.float32)}
    upd = S.incremental_update(new, old, 0.25)
    np.testing.assert_allclose(upd["l"]["k"], [2.5, 0.5]); np.testing.assert_allclose(upd["b"], [1.0])
    assert S.periodic_update(new, old, 8, 4) is new and S.periodic_update(new, old, 7, 4) is old
    m = S.make_modules()
    assert m["jax.lax"].cond is S.cond and m["optax"].incremental_update is S.incremental_update and m["jax.numpy"].squeeze is S.squeeze


def test_flax_stand_ins_run_dense_modules():
    """tests/golden/_flax_shim.py: `Module.apply({'params': p}, x)` binds `nn.Dense` sub-modules created in `setup()` to p[<attribute name>]
    and `Dense` computes x @ kernel + bias (kernel [in, out]) — flax's published behaviour, enough for the reference's head modules."""
    import _flax_shim as F

    class Head(F.Module):
        width: int
        scale: float = 2.0

        def setup(self):
            self.dense1 = F.Dense(features=self.width)
            self.out = F.Dense(features=1, use_bias=False)

        def __call__(self, x, *, train):
            return self.out(F.relu(self.dense1(x))) * self.scale

    rng = np.random.RandomState(0)
    p = {"dense1": {"kernel": rng.randn(3, 4).astype(np.float32), "bias": rng.randn(4).astype(np.float32)}, "out": {"kernel": rng.randn(4, 1).astype(np.float32)}}
    x = rng.randn(5, 3).astype(np.float32)
    y = Head(4).apply({"params": p}, S.asarray(x), train=False, rngs=None)
    np.testing.assert_allclose(np.asarray(y), np.maximum(x @ p["dense1"]["kernel"] + p["dense1"]["bias"], 0) @ p["out"]["kernel"] * 2.0, rtol=1e-6)
    node = F.PyTreeNode(a=1, b=2).replace(b=3)
    assert (node.a, node.b) == (1, 3)


def test_log_softmax_and_empty():
    x = RNG.randn(3, 5).astype(np.float32) * 4
    ls = S.log_softmax(a(x), axis=-1)
    np.testing.assert_allclose(np.exp(np.asarray(ls)).sum(-1), 1.0, rtol=1e-6)
    np.testing.assert_allclose(np.asarray(ls), x - np.log(np.exp(x.astype(np.float64)).sum(-1, keepdims=True)), rtol=1e-5, atol=1e-6)
    e = S.empty((4,), dtype=np.float32)
    assert e.shape == (4,) and e.dtype == np.float32 and isinstance(e.at[1].set(2.0), S.Arr)
