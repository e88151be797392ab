"""Checkpoint import/export: the directory layout the reference's train loops write
(`LLM_RL/algorithms/ppo/train.py:28-85`, `ilql/train.py:25-175`: `<dir>/policy/{config.json, params.msgpack |
train_state.msgpack}`, `<dir>/value_head/...`, `<dir>/q1_head/...` ...) so that a BC / ILQL / PPO checkpoint trained with
the reference drives the native engine, and weights trained here can be handed back.

The tensors are flax `msgpack` pytrees (`flax.serialization.msgpack_serialize`, called by JaxSeq `save_pytree`): nested
string-keyed maps whose leaves are msgpack ExtType 1 = packed `(shape, dtype name, raw bytes)`, ExtType 3 = numpy scalar,
and — for arrays above 2^30 bytes — a `{"__msgpack_chunked_array__": True, "shape": ..., "chunks": {...}}` map.  flax and
jax are not installable here, so the leaf encoding follows the published format: **unpinned** against a real flax blob.  The file LAYOUT has two
forms: one msgpack map (`flax.serialization.to_bytes(tree)`), or the streaming form — a concatenation of `(key path, to_bytes(leaf))` records —
which the reference's own `save_pytree` writes (llm_rl_scripts/twenty_questions/env/convert_checkpoints.py:36-47, adapted from EasyLM as JaxSeq's
is); that one IS pinned: tests/golden/ckpt_stream.json holds a file written by that reference function (executed with its flax / jax calls on
stand-ins), `load_msgpack_tree` reads it back leaf for leaf and `save_msgpack_tree(streaming=True)` reproduces its bytes.

Parameter naming: HF-Flax GPT-2 (`transformer/{wte,wpe}/embedding`, `transformer/h/<l>/{ln_1,ln_2}/{scale,bias}`,
`.../attn/{c_attn,c_proj}/{kernel,bias}`, `.../mlp/{c_fc,c_proj}/{kernel,bias}`, `transformer/ln_f/{scale,bias}`) with
`FlaxConv1D` kernels stored `[out, in]` — the transpose of the PyTorch `Conv1D` `[in, out]` layout our state dicts use.
Heads: `LinearHead` = `{"dense": {"kernel", "bias"}}`, `MLPHead` = `{"dense1": {...}, "dense2": {...}}`
(`LLM_RL/heads/linear_head.py:112-123`, `mlp_head.py:139-148`).
"""
from __future__ import annotations

import json
import os
from typing import Any, Dict, Optional, Tuple

import numpy as np

_CHUNK_KEY = "__msgpack_chunked_array__"
_MAX_CHUNK = 2 ** 30


# ---------------------------------------------------------------------------------------------- flax msgpack pytrees
def _ext_hook(code: int, data: bytes):
    import msgpack
    if code == 1:                                   # ndarray: (shape, dtype name, buffer)
        shape, dtype, buf = msgpack.unpackb(data, raw=False)
        return np.frombuffer(buf, dtype=_np_dtype(dtype)).reshape(shape).copy()
    if code == 2:                                   # native complex
        re, im = msgpack.unpackb(data, raw=False)
        return complex(re, im)
    if code == 3:                                   # numpy scalar
        shape, dtype, buf = msgpack.unpackb(data, raw=False)
        return np.frombuffer(buf, dtype=_np_dtype(dtype)).reshape(shape)[()]
    return msgpack.ExtType(code, data)


def _np_dtype(name: str):
    if name == "bfloat16":                          # numpy has no bfloat16: widen to float32 on load
        return np.dtype("uint16")
    return np.dtype(name)


def _unchunk(tree):
    if isinstance(tree, dict):
        if tree.get(_CHUNK_KEY):
            chunks, shape = tree["chunks"], tree["shape"]        # both are {"0": ..., "1": ...} maps (flax `_tuple_to_dict`)
            flat = np.concatenate([np.asarray(chunks[str(i)]).reshape(-1) for i in range(len(chunks))])
            if isinstance(shape, dict):
                shape = [shape[str(i)] for i in range(len(shape))]
            return flat.reshape(shape)
        return {k: _unchunk(v) for k, v in tree.items()}
    return tree


def load_msgpack_tree(path: str) -> Dict[str, Any]:
    import msgpack
    with open(path, "rb") as f:
        raw = f.read()
    # bfloat16 leaves need their dtype name: decode ExtType 1 by hand to widen them
    def hook(code, data):
        if code in (1, 3):
            shape, dtype, buf = msgpack.unpackb(data, raw=False)
            if dtype == "bfloat16":
                u = np.frombuffer(buf, dtype=np.uint16).astype(np.uint32) << 16
                arr = u.view(np.float32).reshape(shape)
                return arr.copy() if code == 1 else arr[()]
        return _ext_hook(code, data)
    # sniff the layout before parsing anything: a streamed file starts with a 2-element array (0x92: the first `(key path, bytes)` record), a
    # one-map file (flax `to_bytes`, this module's default writer) with a map header (fixmap 0x8X / map16 0xde / map32 0xdf) — decoding a whole
    # one-map checkpoint through the record reader first cost 3x the load time and 2x the peak memory
    if raw[:1] == b"\x92":
        streamed = _load_streamed(raw, hook)
        if streamed is not None:
            return streamed
    return _unchunk(msgpack.unpackb(raw, ext_hook=hook, raw=False, strict_map_key=False))


def _load_streamed(raw: bytes, hook) -> Optional[Dict[str, Any]]:
    """The streaming layout (`save_pytree` of llm_rl_scripts/twenty_questions/env/convert_checkpoints.py:36-47, adapted from EasyLM like JaxSeq's):
    a concatenation of msgpack records `(key path, flax to_bytes(leaf))` over the flattened state dict, instead of ONE msgpack map.  Returns None
    when `raw` is not in that layout.  Pinned to a file written by the reference's own function (tests/golden/ckpt_stream.json)."""
    import io
    import msgpack
    unp = msgpack.Unpacker(io.BytesIO(raw), raw=False, strict_map_key=False, ext_hook=hook, max_buffer_size=max(len(raw), 1))
    tree: Dict[str, Any] = {}
    n = 0
    for rec in unp:
        if not (isinstance(rec, (list, tuple)) and len(rec) == 2 and isinstance(rec[0], (list, tuple)) and len(rec[0]) > 0
                and all(isinstance(k, str) for k in rec[0]) and isinstance(rec[1], (bytes, bytearray))):
            return None
        leaf = msgpack.unpackb(rec[1], ext_hook=hook, raw=False, strict_map_key=False) if len(rec[1]) else {}
        node = tree
        for k in rec[0][:-1]:
            node = node.setdefault(k, {})
        node[rec[0][-1]] = _unchunk(leaf)
        n += 1
    return tree if n else None


def _pack_leaf(x):
    import msgpack
    if isinstance(x, np.ndarray):
        if x.nbytes > _MAX_CHUNK:
            flat = x.reshape(-1)
            per = max(1, _MAX_CHUNK // x.dtype.itemsize)
            chunks = {str(i): _pack_leaf(np.ascontiguousarray(flat[o:o + per])) for i, o in enumerate(range(0, flat.size, per))}
            return {_CHUNK_KEY: True, "shape": {str(i): int(n) for i, n in enumerate(x.shape)}, "chunks": chunks}
        return msgpack.ExtType(1, msgpack.packb((list(x.shape), x.dtype.name, np.ascontiguousarray(x).tobytes()), use_bin_type=True))
    if isinstance(x, np.generic):
        return msgpack.ExtType(3, msgpack.packb(([], x.dtype.name, x.tobytes()), use_bin_type=True))
    return x


def save_msgpack_tree(path: str, tree: Dict[str, Any], streaming: bool = False) -> None:
    """streaming=False: one flax `to_bytes` map; True: the record-per-leaf layout of the reference's streaming `save_pytree` (see `_load_streamed`)."""
    import msgpack
    conv = lambda t: {k: conv(v) for k, v in t.items()} if isinstance(t, dict) else _pack_leaf(t)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        if not streaming:
            f.write(msgpack.packb(conv(tree), use_bin_type=True))
            return
        packer = msgpack.Packer()

        def walk(t, prefix):
            for k, v in t.items():
                if isinstance(v, dict) and v:
                    walk(v, prefix + (k,))
                else:
                    f.write(packer.pack((prefix + (k,), msgpack.packb(conv(v) if isinstance(v, dict) else _pack_leaf(v), use_bin_type=True))))
        walk(tree, ())


def _np(x) -> np.ndarray:
    if hasattr(x, "detach"):
        x = x.detach().cpu().float().numpy()
    return np.asarray(x)


# ---------------------------------------------------------------------------------------------- GPT-2 name maps
def flax_gpt2_params_to_state_dict(params: Dict[str, Any]) -> Dict[str, np.ndarray]:
    """HF-Flax GPT-2 param tree -> HF PyTorch-style names ([in, out] Conv1D kernels) as float32 numpy arrays."""
    tr = params["transformer"] if "transformer" in params else params
    f = lambda a: np.asarray(a, dtype=np.float32)
    sd = {"wte.weight": f(tr["wte"]["embedding"]), "wpe.weight": f(tr["wpe"]["embedding"]),
          "ln_f.weight": f(tr["ln_f"]["scale"]), "ln_f.bias": f(tr["ln_f"]["bias"])}
    for l in sorted(tr["h"], key=int):
        blk, p = tr["h"][l], f"h.{int(l)}."
        sd[p + "ln_1.weight"], sd[p + "ln_1.bias"] = f(blk["ln_1"]["scale"]), f(blk["ln_1"]["bias"])
        sd[p + "ln_2.weight"], sd[p + "ln_2.bias"] = f(blk["ln_2"]["scale"]), f(blk["ln_2"]["bias"])
        for mod, name in (("attn", "c_attn"), ("attn", "c_proj"), ("mlp", "c_fc"), ("mlp", "c_proj")):
            sd[p + f"{mod}.{name}.weight"] = np.ascontiguousarray(f(blk[mod][name]["kernel"]).T)      # [out,in] -> [in,out]
            sd[p + f"{mod}.{name}.bias"] = f(blk[mod][name]["bias"])
    return sd


def state_dict_to_flax_gpt2_params(sd: Dict[str, Any]) -> Dict[str, Any]:
    sd = {(k[len("transformer."):] if k.startswith("transformer.") else k): _np(v).astype(np.float32) for k, v in sd.items()}
    n_layer = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("h."))
    tr: Dict[str, Any] = {"wte": {"embedding": sd["wte.weight"]}, "wpe": {"embedding": sd["wpe.weight"]},
                          "ln_f": {"scale": sd["ln_f.weight"], "bias": sd["ln_f.bias"]}, "h": {}}
    for l in range(n_layer):
        p = f"h.{l}."
        blk = {"ln_1": {"scale": sd[p + "ln_1.weight"], "bias": sd[p + "ln_1.bias"]},
               "ln_2": {"scale": sd[p + "ln_2.weight"], "bias": sd[p + "ln_2.bias"]}, "attn": {}, "mlp": {}}
        for mod, name in (("attn", "c_attn"), ("attn", "c_proj"), ("mlp", "c_fc"), ("mlp", "c_proj")):
            blk[mod][name] = {"kernel": np.ascontiguousarray(sd[p + f"{mod}.{name}.weight"].T), "bias": sd[p + f"{mod}.{name}.bias"]}
        tr["h"][str(l)] = blk
    return {"transformer": tr}


def head_params_from_flax(tree: Dict[str, Any]) -> Dict[str, np.ndarray]:
    """`{"dense": {...}}` -> `{"kernel", "bias"}` ; `{"dense1": ..., "dense2": ...}` -> `{"dense1.kernel", ...}`."""
    f = lambda a: np.asarray(a, dtype=np.float32)
    if "dense" in tree:
        return {"kernel": f(tree["dense"]["kernel"]), "bias": f(tree["dense"]["bias"])}
    return {f"{d}.{n}": f(tree[d][n]) for d in ("dense1", "dense2") for n in ("kernel", "bias")}


def head_params_to_flax(p: Dict[str, Any]) -> Dict[str, Any]:
    if "kernel" in p:
        return {"dense": {"kernel": _np(p["kernel"]), "bias": _np(p["bias"])}}
    return {d: {n: _np(p[f"{d}.{n}"]) for n in ("kernel", "bias")} for d in ("dense1", "dense2")}


# ---------------------------------------------------------------------------------------------- directory layout
def _load_params_dir(path: str) -> Tuple[Dict[str, Any], Dict[str, Any]]:
    cfg = {}
    cj = os.path.join(path, "config.json")
    if os.path.exists(cj):
        with open(cj) as f:
            cfg = json.load(f)
    pm, ts = os.path.join(path, "params.msgpack"), os.path.join(path, "train_state.msgpack")
    if os.path.exists(pm):
        return cfg, load_msgpack_tree(pm)
    if os.path.exists(ts):                                    # TrainState: {'step', 'params', 'opt_state', ...}
        return cfg, load_msgpack_tree(ts)["params"]
    raise FileNotFoundError(f"neither params.msgpack nor train_state.msgpack under {path}")


def load_gpt2_checkpoint(path: str):
    """-> (GPT2Config, state_dict of float32 numpy arrays) from `<path>/{config.json, params.msgpack|train_state.msgpack}`."""
    from .gpt2 import GPT2Config
    cfg_json, tree = _load_params_dir(path)
    sd = flax_gpt2_params_to_state_dict(tree)
    d = sd["wte.weight"].shape[1]
    if "n_head" not in cfg_json:
        # the head count is not recoverable from the tensors (c_attn is [d, 3d] for any head count): refuse instead of guessing
        raise ValueError(f"{path}/config.json has no 'n_head': cannot reconstruct the GPT-2 configuration")
    cfg = GPT2Config(n_layer=1 + max(int(k.split(".")[1]) for k in sd if k.startswith("h.")),
                     n_head=int(cfg_json["n_head"]), d_model=d, d_ff=sd["h.0.mlp.c_fc.weight"].shape[1],
                     vocab=sd["wte.weight"].shape[0], n_pos=sd["wpe.weight"].shape[0],
                     ln_eps=float(cfg_json.get("layer_norm_epsilon", 1e-5)))
    return cfg, sd


def save_gpt2_checkpoint(path: str, cfg, state_dict: Dict[str, Any], extra_config: Optional[Dict[str, Any]] = None) -> None:
    os.makedirs(path, exist_ok=True)
    cj = dict(model_type="gpt2", n_layer=cfg.n_layer, n_head=cfg.n_head, n_embd=cfg.d_model, n_inner=cfg.d_ff, vocab_size=cfg.vocab,
              n_positions=cfg.n_pos, layer_norm_epsilon=cfg.ln_eps, activation_function="gelu_new")
    cj.update(extra_config or {})
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cj, f, indent=2)
    save_msgpack_tree(os.path.join(path, "params.msgpack"), state_dict_to_flax_gpt2_params(state_dict))


def load_head_checkpoint(path: str) -> Tuple[Dict[str, Any], Dict[str, np.ndarray]]:
    cfg_json, tree = _load_params_dir(path)
    return cfg_json, head_params_from_flax(tree)


def save_head_checkpoint(path: str, params: Dict[str, Any], config: Optional[Dict[str, Any]] = None) -> None:
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(config or {}, f, indent=2)
    save_msgpack_tree(os.path.join(path, "params.msgpack"), head_params_to_flax(params))


def load_hf_pytorch_gpt2(path: str):
    """HF PyTorch GPT-2 directory (`config.json` + `model.safetensors` | `pytorch_model.bin`) -> (GPT2Config, state dict)."""
    from .gpt2 import GPT2Config
    with open(os.path.join(path, "config.json")) as f:
        cj = json.load(f)
    st = os.path.join(path, "model.safetensors")
    if os.path.exists(st):
        from safetensors.numpy import load_file
        raw = load_file(st)
    else:
        import torch
        raw = {k: v.float().numpy() for k, v in torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu").items()}
    sd = {}
    for k, v in raw.items():
        k = k[len("transformer."):] if k.startswith("transformer.") else k
        if k.endswith(".attn.bias") or k.endswith(".attn.masked_bias") or k == "lm_head.weight":
            continue
        sd[k] = np.asarray(v, dtype=np.float32)
    d = sd["wte.weight"].shape[1]
    cfg = GPT2Config(n_layer=cj["n_layer"], n_head=cj["n_head"], d_model=d, d_ff=cj.get("n_inner") or 4 * d, vocab=sd["wte.weight"].shape[0],
                     n_pos=sd["wpe.weight"].shape[0], ln_eps=cj.get("layer_norm_epsilon", 1e-5))
    return cfg, sd
