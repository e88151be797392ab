"""Import shim used ONLY by the fixture generators in this directory.

It makes the pure-Python / numpy-only parts of the reference importable in the
build container (where jax / flax / optax / JaxSeq are not installed) by
registering permissive stand-in modules for those packages.  Nothing here is
shipped or used at run time: fixtures are generated once, committed as data,
and the tests only read the data files.
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import sys
import types

REFERENCE_ROOT = "/root/reference"

_STUB_ROOTS = (
    "jax", "jaxlib", "flax", "optax", "chex", "JaxSeq", "jaxtyping", "gcsfs", "wandb",
    "tyro", "transformers", "termcolor", "IPython", "skimage", "tiktoken", "openai", "nltk", "jax_models",
)


class _AnyMeta(type):
    def __getattr__(cls, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Any

    def __getitem__(cls, item):
        return _Any

    def __or__(cls, other):
        return _Any

    def __ror__(cls, other):
        return _Any


class _Any(metaclass=_AnyMeta):
    """Callable / subscriptable / subclassable placeholder."""
    _is_placeholder = True

    def __new__(cls, *args, **kwargs):
        # decorator use: @pjit, @partial-like (placeholders only — a reference class that merely SUBCLASSES a placeholder constructs normally)
        if cls.__dict__.get("_is_placeholder") and len(args) == 1 and not kwargs and callable(args[0]) and not isinstance(args[0], type):
            return args[0]
        return super().__new__(cls)

    def __init__(self, *args, **kwargs):
        pass

    def __init_subclass__(cls, **kwargs):
        pass

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Any

    def __call__(self, *args, **kwargs):
        if len(args) == 1 and callable(args[0]):
            return args[0]
        return _Any()

    def __getitem__(self, item):
        return _Any

    def __iter__(self):
        return iter(())


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        # one distinct placeholder class per module attribute, so that `class X(mod.A, mod.B)` has two different bases
        ph = type(name, (_Any,), {"_is_placeholder": True})
        self.__dict__[name] = ph
        return ph


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        if spec.name in _SHIM_MODULES:      # numpy-backed jax.numpy / jax.nn / jax.lax / optax (tests/golden/_jnp_shim.py)
            real = _SHIM_MODULES[spec.name]
            m = _StubModule(spec.name)
            m.__path__ = []
            m.__dict__.update({k: v for k, v in real.__dict__.items() if not k.startswith("__")})
            return m
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        if module.__name__ == "termcolor":
            module.colored = lambda s, *a, **k: s
        if module.__name__ == "IPython":
            module.embed = lambda *a, **k: None
        if module.__name__ == "jax" and _SHIM_MODULES:      # `jax.nn.one_hot` / `jax.lax.stop_gradient` are reached by attribute
            import importlib
            for sub in ("numpy", "nn", "lax"):
                setattr(module, sub, importlib.import_module("jax." + sub))
        # stand-in sub-modules registered by `install(extra=...)` (e.g. flax.linen, flax.struct) are also reachable as attributes of their parent
        import importlib
        for key in list(_SHIM_MODULES):
            parent, _, child = key.rpartition(".")
            if parent == module.__name__ and child not in module.__dict__ and key not in ("jax.numpy", "jax.nn", "jax.lax"):
                setattr(module, child, importlib.import_module(key))


_SHIM_MODULES = {}


def install(jnp_shim: bool = False, extra=None):
    """jnp_shim=True additionally backs `jax.numpy`, `jax.nn`, `jax.lax` and `optax` with the numpy restatements of
    tests/golden/_jnp_shim.py so that the reference's loss functions execute (every other jax/flax name stays a placeholder).
    extra: {module name: module object} of further stand-ins (tests/golden/_flax_shim.py: `flax.linen`, `flax.struct`)."""
    if jnp_shim and "jax.numpy" not in _SHIM_MODULES:
        import os
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import _jnp_shim
        _SHIM_MODULES.update(_jnp_shim.make_modules())
        assert not any(k in sys.modules for k in _SHIM_MODULES), "install(jnp_shim=True) must run before the reference is imported"
    if extra:
        assert not any(k in sys.modules for k in extra), "install(extra=...) must run before the reference is imported"
        _SHIM_MODULES.update(extra)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    if not any(isinstance(f, _StubFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _StubFinder())
