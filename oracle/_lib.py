"""ctypes loader for oracle/_build/liboracle.so (test infrastructure only)."""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "wordle_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return _SO


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        u32p = ctypes.POINTER(ctypes.c_uint32)
        L.orc_mt_stream.argtypes = [u32p, ctypes.c_int, u32p, ctypes.c_int]
        L.orc_mt_choices.argtypes = [u32p, ctypes.c_int, u32p, u32p, ctypes.c_int]
        L.orc_wordle_create.restype = ctypes.c_void_p
        L.orc_wordle_create.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_double]
        L.orc_wordle_destroy.argtypes = [ctypes.c_void_p]
        L.orc_wordle_reset.argtypes = [ctypes.c_void_p, u32p, ctypes.c_int]
        L.orc_wordle_step.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p,
                                      ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double),
                                      ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        L.orc_wordle_get_state.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_int)]
        L.orc_wordle_run.restype = ctypes.c_long
        L.orc_wordle_run.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.POINTER(ctypes.c_int32), ctypes.c_int]
        L.orc_wordle_run_mt.restype = ctypes.c_long
        L.orc_wordle_run_mt.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.POINTER(ctypes.c_int32), ctypes.c_int, ctypes.c_int,
                                        ctypes.POINTER(ctypes.c_int)]
        _lib = L
    return _lib


def seed_key(seed: int):
    """|seed| as little-endian 32-bit limbs — CPython random_seed(): at least one limb."""
    n = abs(int(seed))
    limbs = []
    while n:
        limbs.append(n & 0xFFFFFFFF)
        n >>= 32
    if not limbs:
        limbs = [0]
    arr = (ctypes.c_uint32 * len(limbs))(*limbs)
    return arr, len(limbs)
