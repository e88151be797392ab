"""CPU tier: the streaming checkpoint layout, pinned to a file WRITTEN BY THE REFERENCE's own `save_pytree`
(llm_rl_scripts/twenty_questions/env/convert_checkpoints.py:36-47, executed by tests/golden/make_ckpt_fixture.py with the flax / jax library calls on
stand-ins): record framing and key paths are the reference's; the leaf encoding is flax's published `msgpack_serialize` format, restated."""
import json
import os

import numpy as np

import lmrl_gym_amd  # noqa: F401
from lmrl_gym_amd import checkpoints as C

HERE = os.path.dirname(os.path.abspath(__file__))


def _fixture():
    with open(os.path.join(HERE, "golden", "ckpt_stream.json")) as f:
        return json.load(f)


def _leaves(tree, prefix=()):
    for k, v in tree.items():
        if isinstance(v, dict):
            yield from _leaves(v, prefix + (k,))
        else:
            yield "/".join(prefix + (k,)), v


def test_reads_the_file_the_reference_writer_produced(tmp_path):
    fx = _fixture()
    p = tmp_path / "params.msgpack"
    p.write_bytes(bytes.fromhex(fx["file_hex"]))
    tree = C.load_msgpack_tree(str(p))
    got = dict(_leaves(tree))
    assert set(got) == set(fx["leaves"])
    for name, want in fx["leaves"].items():
        a = np.asarray(got[name])
        assert list(a.shape) == want["shape"] and a.dtype.name == want["dtype"], name
        assert np.array_equal(a.reshape(-1), np.asarray(want["values"], dtype=a.dtype)), name
    # and the GPT-2 / head name maps run on it (FlaxConv1D kernels are [out, in])
    sd = C.flax_gpt2_params_to_state_dict(tree)
    assert sd["h.1.mlp.c_fc.weight"].shape == (8, 16) and sd["wte.weight"].shape == (11, 8)
    assert np.array_equal(sd["h.0.attn.c_attn.weight"], np.asarray(got["transformer/h/0/attn/c_attn/kernel"]).T)
    hp = C.head_params_from_flax(tree["head"])
    assert hp["dense2.kernel"].shape == (8, 1)


def test_streaming_writer_reproduces_the_reference_file_bytes(tmp_path):
    fx = _fixture()
    p = tmp_path / "a.msgpack"
    p.write_bytes(bytes.fromhex(fx["file_hex"]))
    tree = C.load_msgpack_tree(str(p))
    q = tmp_path / "b.msgpack"
    C.save_msgpack_tree(str(q), tree, streaming=True)
    assert q.read_bytes() == bytes.fromhex(fx["file_hex"])
    C.save_msgpack_tree(str(q), tree)                       # the one-map layout still round-trips
    back = dict(_leaves(C.load_msgpack_tree(str(q))))
    for name, want in fx["leaves"].items():
        assert np.array_equal(np.asarray(back[name]).reshape(-1), np.asarray(want["values"], dtype=want["dtype"])), name
