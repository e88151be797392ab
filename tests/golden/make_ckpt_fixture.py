"""Generates tests/golden/ckpt_stream.json: a checkpoint file written by the REFERENCE's own streaming writer.

`llm_rl_scripts/twenty_questions/env/convert_checkpoints.py:36-47` (`save_pytree`, "adapted from EasyLM": the layout JaxSeq's `save_pytree`
streams as well) is imported from /root/reference and executed unmodified: the file is a concatenation of msgpack records
`(key path tuple, flax.serialization.to_bytes(leaf))` over `flax.traverse_util.flatten_dict(to_state_dict(tree), keep_empty_nodes=True)`.
flax / jax are not installable here, so the four library calls it makes run on stand-ins defined below — `to_state_dict` (identity on nested dicts),
`flatten_dict` (restated), `jax.device_get` (identity) and `to_bytes` = flax's published `msgpack_serialize` leaf encoding (ExtType 1 =
msgpack((shape, dtype name, raw bytes))).  What this pins: the record framing and key paths come from the reference's code; the leaf encoding is
a restatement of the dependency's format (stated as such in lmrl-gym_amd/checkpoints.py).

The fixture holds the file bytes (hex) and every leaf's values; tests/test_checkpoints_stream.py reads the bytes back with
`lmrl_gym_amd.checkpoints.load_msgpack_tree` and compares.  Run from the repo root:  python tests/golden/make_ckpt_fixture.py
"""
import io
import json
import os
import sys
import types

import msgpack
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402


def _leaf_bytes(x):
    """flax.serialization.to_bytes of one leaf (msgpack_serialize: `_msgpack_ext_pack`)."""
    def ext(v):
        if isinstance(v, np.ndarray):
            return msgpack.ExtType(1, msgpack.packb((v.shape, v.dtype.name, v.tobytes()), use_bin_type=True))
        if isinstance(v, np.generic):
            return msgpack.ExtType(3, msgpack.packb(((), v.dtype.name, v.tobytes()), use_bin_type=True))
        return v
    return msgpack.packb(x, default=ext, strict_types=True)


def _flatten(tree, keep_empty_nodes=False, prefix=()):
    out = {}
    for k, v in tree.items():
        if isinstance(v, dict) and v:
            out.update(_flatten(v, keep_empty_nodes, prefix + (k,)))
        elif isinstance(v, dict):
            if keep_empty_nodes:
                out[prefix + (k,)] = v            # flax keeps an `empty_node` sentinel; none occur in these trees
        else:
            out[prefix + (k,)] = v
    return out


ser = types.ModuleType("flax.serialization")
ser.to_state_dict = lambda t: t
ser.to_bytes = _leaf_bytes
trv = types.ModuleType("flax.traverse_util")
trv.flatten_dict = _flatten
_ref_import.install(extra={"flax.serialization": ser, "flax.traverse_util": trv})
import jax  # noqa: E402  (stand-in)
jax.device_get = lambda x: x
from llm_rl_scripts.twenty_questions.env.convert_checkpoints import save_pytree  # noqa: E402


def main():
    rng = np.random.RandomState(5)
    d, ff, V, P, L = 8, 16, 11, 6, 2
    f = lambda *s: rng.randn(*s).astype(np.float32)
    tr = {"wte": {"embedding": f(V, d)}, "wpe": {"embedding": f(P, d)}, "ln_f": {"scale": f(d), "bias": f(d)}, "h": {}}
    for l in range(L):
        tr["h"][str(l)] = {"ln_1": {"scale": f(d), "bias": f(d)}, "ln_2": {"scale": f(d), "bias": f(d)},
                           "attn": {"c_attn": {"kernel": f(3 * d, d), "bias": f(3 * d)}, "c_proj": {"kernel": f(d, d), "bias": f(d)}},
                           "mlp": {"c_fc": {"kernel": f(ff, d), "bias": f(ff)}, "c_proj": {"kernel": f(d, ff), "bias": f(d)}}}
    tree = {"transformer": tr, "head": {"dense1": {"kernel": f(d, d), "bias": f(d)}, "dense2": {"kernel": f(d, 1), "bias": f(1)}},
            "step": np.int32(7)}
    bufs = {}

    class _F(io.BytesIO):
        def __init__(self, path):
            super().__init__()
            self.path = path

        def close(self):
            bufs[self.path] = self.getvalue()
            super().close()

    save_pytree(lambda path, mode: _F(path), tree, "params.msgpack")
    raw = bufs["params.msgpack"]
    leaves = {"/".join(k): {"shape": list(np.shape(v)), "dtype": np.asarray(v).dtype.name, "values": np.asarray(v).reshape(-1).tolist()}
              for k, v in _flatten(tree).items()}
    out = {"source": "llm_rl_scripts/twenty_questions/env/convert_checkpoints.py:36-47 save_pytree, executed (flax / jax calls on stand-ins)",
           "file_hex": raw.hex(), "leaves": leaves}
    with open(os.path.join(HERE, "ckpt_stream.json"), "w") as fh:
        json.dump(out, fh)
    print("wrote ckpt_stream.json:", len(raw), "bytes,", len(leaves), "leaves")


if __name__ == "__main__":
    main()
