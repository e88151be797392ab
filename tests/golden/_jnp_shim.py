"""numpy-backed stand-ins for the handful of `jax.numpy` / `jax.nn` / `jax.lax` / `optax` calls the reference's loss
functions make — FIXTURE-GENERATION AND TEST INFRASTRUCTURE ONLY.

jax / optax cannot be installed in the build container (no network), yet the reference's loss functions are short pure
array programs:  ppo_loss_fn, whiten (LLM_RL/algorithms/ppo/base_interface.py:72-142,245-251), get_query_indicators,
ilql_loss (ilql/base_interface.py:22-119), mc_loss (mc_returns/base_interface.py:19-60), get_rtg (mc_returns/data.py:10-14),
bc_loss (bc/interface.py:28-43), get_tensor_stats / unpad_array (utils.py:12-38), token_logprobs_from_logits
(ppo/base_interface.py:396-403).  With this shim registered as `jax` / `optax` (tests/golden/_ref_import.py) those functions
are imported from /root/reference and executed UNMODIFIED; tests/golden/make_loss_fixtures.py commits their outputs.

Semantics reproduced (each has a unit test in tests/test_jnp_shim.py):
  * JAX type promotion with x64 disabled: float64 -> float32, int64 -> int32; int/bool (x) float32 -> float32; Python scalars
    are weakly typed (numpy >= 2 NEP 50 already behaves that way).  Reductions run in numpy's float32 pairwise order — XLA's
    order differs, i.e. the reference itself is only defined up to float32 reduction order (~1e-7 relative).
  * `x.at[idx].set(v)` functional update, `jnp.argwhere(size=, fill_value=)`, `jax.nn.one_hot` (out-of-range -> zero row),
    `jnp.min/max(where=, initial=)`, `jnp.std/var(where=)`.
  * optax 0.1.3 (requirements.txt:2) `l2_loss` = 0.5 (x - y)^2 and `softmax_cross_entropy_with_integer_labels`
    (max-shifted log-sum-exp minus the label logit) restated from the published implementation.
  * `jax.lax.stop_gradient`: identity on values.  In DERIVATIVE MODE (`complex_step(True)`) arrays are complex128, every function
    here is the analytic continuation of its real version (comparisons / max / min / clip / where decide on the real part) and
    stop_gradient drops the imaginary part — so f(x + i h d).imag / h is the directional derivative of the reference's own code
    INCLUDING its stop_gradient placement (complex-step differentiation, exact to rounding for h = 1e-30).
"""
from __future__ import annotations

import types

import numpy as np

_COMPLEX = False


def complex_step(on: bool) -> None:
    global _COMPLEX
    _COMPLEX = bool(on)


def _narrow(a: np.ndarray) -> np.ndarray:
    if _COMPLEX:
        return a
    if a.dtype == np.float64:
        return a.astype(np.float32)
    if a.dtype == np.int64:
        return a.astype(np.int32)
    if a.dtype == np.complex128 or a.dtype == np.complex64:
        raise TypeError("complex array outside derivative mode")
    return a


_ORDER_UFUNCS = {np.greater, np.greater_equal, np.less, np.less_equal}
_SELECT_UFUNCS = {np.maximum: np.greater_equal, np.minimum: np.less_equal}


class Arr(np.ndarray):
    """ndarray with JAX's promotion rules (x64 off) and the `.at[...]` functional-update helper."""

    def __array_ufunc__(self, ufunc, method, *inputs, out=None, **kwargs):
        ins = []
        for x in inputs:
            if isinstance(x, np.ndarray):
                x = _narrow(np.asarray(x))
            ins.append(x)
        has_float = any(isinstance(x, np.ndarray) and x.dtype.kind in "fc" for x in ins)
        if has_float and ufunc not in (np.logical_and, np.logical_or, np.logical_not, np.logical_xor):
            ft = np.complex128 if _COMPLEX else np.float32
            ins = [x.astype(ft) if isinstance(x, np.ndarray) and x.dtype.kind in "iub" else x for x in ins]
        if _COMPLEX and method == "__call__" and any(isinstance(x, np.ndarray) and x.dtype.kind == "c" for x in ins):
            if ufunc in _ORDER_UFUNCS:                       # order relations decide on the real part
                ins = [x.real if isinstance(x, np.ndarray) else (x.real if isinstance(x, complex) else x) for x in ins]
            elif ufunc in _SELECT_UFUNCS:
                a, b = (np.asarray(x, dtype=np.complex128) for x in ins)
                a, b = np.broadcast_arrays(a, b)
                return _wrap(np.where(_SELECT_UFUNCS[ufunc](a.real, b.real), a, b))
        if out is not None:
            kwargs["out"] = tuple(np.asarray(o) if isinstance(o, Arr) else o for o in out)
        res = getattr(ufunc, method)(*ins, **kwargs)
        if isinstance(res, tuple):
            return tuple(_wrap(r) for r in res)
        return _wrap(res)

    # --- jnp array API used by the reference
    @property
    def at(self):
        return _At(self)

    def astype(self, dtype, *a, **k):
        if _COMPLEX and np.dtype(dtype).kind == "f" and self.dtype.kind == "c":
            return self
        if _COMPLEX and np.dtype(dtype).kind == "f":
            return _wrap(np.asarray(self).astype(np.complex128))
        if np.dtype(dtype).kind in "biu" and self.dtype.kind == "c":
            return _wrap(np.asarray(self).real.astype(dtype))
        return _wrap(np.asarray(self).astype(dtype, *a, **k))

    def sum(self, axis=None, **k):
        return sum_(self, axis=axis, **k)

    def reshape(self, *shape, **k):
        return _wrap(np.asarray(self).reshape(*shape, **k))

    def copy(self, *a, **k):
        return _wrap(np.asarray(self).copy())

    def __getitem__(self, idx):
        if isinstance(idx, tuple):
            idx = tuple(np.asarray(i) if isinstance(i, Arr) else i for i in idx)
        elif isinstance(idx, Arr):
            idx = np.asarray(idx)
        return _wrap(np.asarray(self)[idx])


class _At:
    def __init__(self, arr):
        self.arr = arr

    def __getitem__(self, idx):
        return _AtIdx(self.arr, idx)


class _AtIdx:
    def __init__(self, arr, idx):
        self.arr, self.idx = arr, idx

    def set(self, value):
        out = np.asarray(self.arr).copy()
        idx = self.idx
        if isinstance(idx, tuple):
            idx = tuple(np.asarray(i) if isinstance(i, np.ndarray) else i for i in idx)
        out[idx] = value
        return _wrap(out)

    def add(self, value):
        out = np.asarray(self.arr).copy()
        np.add.at(out, self.idx, value)
        return _wrap(out)


def _wrap(x) -> Arr:
    return _narrow(np.asarray(x)).view(Arr)


def asarray(x, dtype=None):
    a = np.asarray(x, dtype=dtype)
    if _COMPLEX and a.dtype.kind == "f":
        a = a.astype(np.complex128)
    return _wrap(a)


array = asarray
float32, int32, bool_, uint8 = np.float32, np.int32, np.bool_, np.uint8
ndarray = Arr
inf = np.inf


def _raw(x):
    return np.asarray(x)


def _real_key(a):
    return a.real if a.dtype.kind == "c" else a


def sum_(x, axis=None, keepdims=False, where=None):
    a = _raw(x)
    if where is not None:
        a = np.where(_raw(where), a, 0)
    if a.dtype.kind == "b":
        a = a.astype(np.int32)
    return _wrap(np.sum(a, axis=axis, keepdims=keepdims))


def mean(x, axis=None, where=None):
    a = _raw(x)
    if where is None:
        return _wrap(np.mean(a, axis=axis))
    w = _raw(where).astype(bool)
    return _wrap(np.sum(np.where(w, a, 0), axis=axis) / np.sum(w, axis=axis))


def var(x, axis=None, where=None):
    a = _raw(x)
    m = np.asarray(mean(a, axis=axis, where=where))
    if axis is not None:
        m = np.expand_dims(m, axis)
    d = a - m
    sq = d * d if a.dtype.kind == "c" else np.abs(d) ** 2       # analytic continuation in derivative mode
    return mean(sq, axis=axis, where=where)


def std(x, axis=None, where=None):
    return sqrt(var(x, axis=axis, where=where))


def _minmax(x, axis, where, initial, is_min):
    """Full reduction with optional `where` mask and `initial` (as jnp.min/max): decides on the real part."""
    if axis is not None:
        raise NotImplementedError("axis reductions with where= are not used by the reference losses")
    a = _raw(x).reshape(-1)
    key = _real_key(a)
    valid = np.ones(a.shape, dtype=bool) if where is None else np.broadcast_to(_raw(where).astype(bool), _raw(x).shape).reshape(-1)
    assert where is None or initial is not None, "jnp.min/max(where=) requires initial="
    ft = np.complex128 if _COMPLEX else np.float32
    if not valid.any():
        return _wrap(np.asarray(initial, dtype=ft))
    cand = np.nonzero(valid)[0]
    i = cand[np.argmin(key[cand]) if is_min else np.argmax(key[cand])]
    if initial is not None and ((is_min and initial < key[i]) or (not is_min and initial > key[i])):
        return _wrap(np.asarray(initial, dtype=ft))
    return _wrap(a[i])


def min(x, axis=None, where=None, initial=None):   # noqa: A001 (mirrors the jnp name)
    return _minmax(x, axis, where, initial, True)


def max(x, axis=None, where=None, initial=None, keepdims=False):   # noqa: A001
    if axis is not None and where is None:
        a = _raw(x)
        idx = np.argmax(_real_key(a), axis=axis)
        out = np.take_along_axis(a, np.expand_dims(idx, axis), axis=axis)
        return _wrap(out if keepdims else np.squeeze(out, axis=axis))
    return _minmax(x, axis, where, initial, False)


def clip(x, a_min=None, a_max=None):
    out = _wrap(_raw(x))
    if a_min is not None:
        out = maximum(out, a_min)
    if a_max is not None:
        out = minimum(out, a_max)
    return out


def maximum(a, b):
    return np.maximum(_wrap(_raw(a)), b if not isinstance(b, np.ndarray) else _wrap(_raw(b)))


def minimum(a, b):
    return np.minimum(_wrap(_raw(a)), b if not isinstance(b, np.ndarray) else _wrap(_raw(b)))


def exp(x):
    return np.exp(_wrap(_raw(x)))


def log(x):
    return np.log(_wrap(_raw(x)))


def sqrt(x):
    return np.sqrt(_wrap(_raw(x)))


def reciprocal(x):
    a = _wrap(_raw(x))
    return 1.0 / a


def where(cond, x=None, y=None):
    if x is None and y is None:
        return tuple(_wrap(i) for i in np.where(_real_key(_raw(cond))))
    return _wrap(np.where(_real_key(_raw(cond)).astype(bool), _raw(x), _raw(y)))


def argwhere(a, size=None, fill_value=None):
    idx = np.argwhere(_real_key(_raw(a)))
    if size is not None:
        pad = np.full((size, idx.shape[1]), 0 if fill_value is None else fill_value, dtype=idx.dtype)
        n = builtins_min(size, idx.shape[0])
        pad[:n] = idx[:n]
        idx = pad
    return _wrap(idx)


import builtins  # noqa: E402

builtins_min = builtins.min


def argmax(a, axis=None):
    return _wrap(np.argmax(_real_key(_raw(a)), axis=axis))


def arange(*a, dtype=None, **k):
    return _wrap(np.arange(*a, dtype=dtype, **k))


def ones(shape, dtype=np.float32):
    return _wrap(np.ones(shape, dtype=dtype))


def zeros(shape, dtype=np.float32):
    return _wrap(np.zeros(shape, dtype=dtype))


def full(shape, fill_value, dtype=None):
    return _wrap(np.full(shape, fill_value, dtype=dtype if dtype is not None else np.float32))


def concatenate(arrs, axis=0):
    arrs = [_raw(a) for a in arrs]
    if any(a.dtype.kind == "c" for a in arrs):
        arrs = [a.astype(np.complex128) for a in arrs]
    elif any(a.dtype.kind == "f" for a in arrs):
        arrs = [a.astype(np.float32) if a.dtype.kind in "iub" else a for a in arrs]
    return _wrap(np.concatenate(arrs, axis=axis))


def expand_dims(a, axis):
    return _wrap(np.expand_dims(_raw(a), axis))


def cumprod(a, axis=None):
    return _wrap(np.cumprod(_raw(a), axis=axis, dtype=_raw(a).dtype))


def triu(a, k=0):
    return _wrap(np.triu(_raw(a), k=k))


def take_along_axis(a, idx, axis):
    return _wrap(np.take_along_axis(_raw(a), _raw(idx).astype(np.int64), axis=axis))


def flip(a, axis=None):
    return _wrap(np.flip(_raw(a), axis=axis))


# ---------------------------------------------------------------------------- jax.nn / jax.lax / optax
def one_hot(x, num_classes, dtype=np.float32):
    idx = _raw(x).astype(np.int64)
    out = np.zeros(idx.shape + (num_classes,), dtype=np.float32)
    ok = (idx >= 0) & (idx < num_classes)                      # out-of-range indices give an all-zero row (jax.nn.one_hot)
    flat = out.reshape(-1, num_classes)
    rows = np.nonzero(ok.reshape(-1))[0]
    flat[rows, idx.reshape(-1)[rows]] = 1.0
    if _COMPLEX:
        out = out.astype(np.complex128)
    elif np.dtype(dtype) != np.float32:
        out = out.astype(dtype)
    return _wrap(out)


def stop_gradient(x):
    a = _raw(x)
    if a.dtype.kind == "c":
        return _wrap(a.real.astype(np.complex128))
    return _wrap(a)


def l2_loss(predictions, targets=None):
    """optax 0.1.3 `l2_loss`: 0.5 * (predictions - targets)^2."""
    e = _wrap(_raw(predictions)) - (0 if targets is None else _wrap(_raw(targets)))
    return 0.5 * e * e                                         # e*e (not |e|^2): analytic in derivative mode


def softmax_cross_entropy_with_integer_labels(logits, labels):
    """optax 0.1.3: logits -= stop_gradient(max); log(sum(exp(logits))) - logits[label]."""
    lg = _wrap(_raw(logits))
    lb = _raw(labels)
    assert lb.dtype.kind in "iu", "labels must be integers (chex.assert_type in optax)"
    m = stop_gradient(max(lg, axis=-1, keepdims=True))
    lg = lg - m
    label_logits = take_along_axis(lg, lb[..., None], axis=-1)[..., 0]
    log_norm = log(sum_(exp(lg), axis=-1))
    return log_norm - label_logits


def log_softmax(x, axis=-1):
    """jax.nn.log_softmax: x - logsumexp(x), max-shifted, float32."""
    x = _raw(x)
    m = np.max(x, axis=axis, keepdims=True)
    return _wrap((x - m) - np.log(np.sum(np.exp(x - m), axis=axis, keepdims=True)))


def empty(shape, dtype=np.float32):
    return _wrap(np.zeros(shape, dtype=dtype))


def squeeze(a, axis=None):
    return _wrap(np.squeeze(_raw(a), axis=axis))


def matmul(a, b):
    """jnp.matmul / `@` on Arr operands (flax Dense: x @ kernel), float32 arithmetic."""
    return _wrap(np.matmul(_raw(a), _raw(b)))


def cond(pred, true_fun, false_fun, *operands):
    """jax.lax.cond with a concrete predicate."""
    return true_fun(*operands) if bool(np.asarray(pred)) else false_fun(*operands)


def _tree_map2(f, a, b):
    if isinstance(a, dict):
        return {k: _tree_map2(f, a[k], b[k]) for k in a}
    return f(a, b)


def incremental_update(new_tensors, old_tensors, step_size):
    """optax.incremental_update (optax 0.1.3 `_src/update.py`): step_size * new + (1 - step_size) * old on every leaf (Polyak averaging)."""
    return _tree_map2(lambda n, o: step_size * n + (1.0 - step_size) * o, new_tensors, old_tensors)


def periodic_update(new_tensors, old_tensors, steps, update_period):
    """optax.periodic_update: `new` when steps % update_period == 0, else `old`."""
    return new_tensors if int(np.asarray(steps)) % int(update_period) == 0 else old_tensors


def make_modules():
    """`jax`, `jax.numpy`, `jax.nn`, `jax.lax`, `optax` module objects backed by this file."""
    g = globals()
    jnp = types.ModuleType("jax.numpy")
    for name in ("asarray", "array", "float32", "int32", "bool_", "uint8", "ndarray", "inf", "mean", "var", "std", "min", "max",
                 "clip", "maximum", "minimum", "exp", "log", "sqrt", "reciprocal", "where", "argwhere", "argmax", "arange", "ones",
                 "zeros", "full", "concatenate", "expand_dims", "cumprod", "triu", "take_along_axis", "flip"):
        setattr(jnp, name, g[name])
    jnp.sum = sum_
    jnp.matmul = jnp.dot = matmul
    jnp.squeeze = squeeze
    jnp.empty = empty
    nn = types.ModuleType("jax.nn")
    nn.one_hot = one_hot
    nn.log_softmax = log_softmax
    lax = types.ModuleType("jax.lax")
    lax.stop_gradient = stop_gradient
    lax.cond = cond
    optax = types.ModuleType("optax")
    optax.l2_loss = l2_loss
    optax.softmax_cross_entropy_with_integer_labels = softmax_cross_entropy_with_integer_labels
    optax.incremental_update = incremental_update
    optax.periodic_update = periodic_update
    return {"jax.numpy": jnp, "jax.nn": nn, "jax.lax": lax, "optax": optax}
