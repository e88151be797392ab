"""Minimal stand-ins for `flax.linen` and `flax.struct` — fixture generators only (see _ref_import.py).

flax==0.7.2 (requirements.txt of the reference) is not installable here.  With these stand-ins the reference's OWN head modules
(`LLM_RL/heads/mlp_head.py::MLPHead`, `linear_head.py::LinearHead`: `setup()` building `nn.Dense` sub-modules, `__call__` chaining
them) and its train-state containers execute unmodified on numpy arrays.  What is restated is only flax's published behaviour:
  * `nn.Module`: dataclass-style fields from the class annotations; `apply({'params': p}, *args, **kw)` runs `setup()` and calls the
    module with every sub-module bound to `p[<attribute name>]`;
  * `nn.Dense(features, use_bias)`: `y = x @ params['kernel'] + params['bias']`, kernel [in, out];
  * `nn.relu`;  `struct.PyTreeNode`: keyword-constructed immutable record with `replace(**changes)`;  `struct.field(...)`: a marker.
"""
from __future__ import annotations

import types

import _jnp_shim as S


class Dense:
    def __init__(self, features, use_bias=True, dtype=None, param_dtype=None, precision=None, kernel_init=None, bias_init=None, name=None):
        self.features, self.use_bias = features, use_bias
        self._params = None

    def __call__(self, x):
        assert self._params is not None, "Dense used outside Module.apply"
        k = S.asarray(self._params["kernel"])
        assert k.shape[-1] == self.features and k.shape[0] == x.shape[-1]
        y = S.matmul(x, k)
        if self.use_bias:
            y = y + S.asarray(self._params["bias"])
        return y


class Module:
    def __init__(self, *args, **kwargs):
        names = [n for klass in reversed(type(self).__mro__) for n in getattr(klass, "__annotations__", {})]
        for n in names:                                   # class-level defaults first
            if hasattr(type(self), n):
                object.__setattr__(self, n, getattr(type(self), n))
        for n, v in zip(names, args):
            object.__setattr__(self, n, v)
        for n, v in kwargs.items():
            object.__setattr__(self, n, v)

    def setup(self):
        pass

    def apply(self, variables, *args, rngs=None, **kwargs):
        params = variables["params"]
        self.setup()
        for name, sub in list(vars(self).items()):
            if isinstance(sub, Dense):
                sub._params = params[name]
        return self(*args, **kwargs)


def relu(x):
    return S.maximum(x, 0.0)


class PyTreeNode:
    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            object.__setattr__(self, k, v)

    def replace(self, **changes):
        new = type(self)(**{**vars(self), **changes})
        return new


def field(*args, **kwargs):
    return None


def make_modules():
    linen = types.ModuleType("flax.linen")
    linen.Module, linen.Dense, linen.relu = Module, Dense, relu
    linen.compact = lambda f: f
    struct = types.ModuleType("flax.struct")
    struct.PyTreeNode, struct.field = PyTreeNode, field
    core = types.ModuleType("flax.core")                      # FrozenDict is only an immutability wrapper: identity on plain dicts
    core.freeze = core.unfreeze = lambda d: d
    frozen = types.ModuleType("flax.core.frozen_dict")
    frozen.freeze = frozen.unfreeze = core.freeze
    return {"flax.linen": linen, "flax.struct": struct, "flax.core": core, "flax.core.frozen_dict": frozen}
