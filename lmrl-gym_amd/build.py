"""In-tree build of liblmrl_amd.so for gfx950 (hipcc cross-compiles without a GPU).

    python lmrl-gym_amd/build.py [--force] [--verbose] [--tools]

`--tools` builds a SECOND library, liblmrl_amd_tools.so (objects under csrc/_obj_tools/, `-DLMRL_TOOLS`): the same sources plus the timing-only
launch ablations of csrc/ablate_tools.h.  Only tools/bench_ablate_decode.py loads it; the package, the tests and bench.py never do.

Every csrc/*.hip is compiled to csrc/_obj/*.o and linked into lmrl-gym_amd/liblmrl_amd.so.  Staleness is decided by
CONTENT, not mtime: csrc/_obj/manifest.json records, per object, the sha256 of its source + every header under csrc/ and
include/ + the compiler flags + `hipcc --version`; an object (and the .so) is reused only when that digest matches, so a
shipped `_obj/` can never be linked against changed sources.  The .so is git-ignored but travels to the GPU box with the tree.
"""
from __future__ import annotations

import concurrent.futures as cf
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
SO = os.path.join(HERE, "liblmrl_amd.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-ffp-contract=off"]


# per-source flags.  flash_attn_train: the sweeps interleave MFMAs with per-element VALU work on their results; with the default AGPR form of the
# MFMA destination the compiler copies every score through v_accvgpr_read/_write (a quarter of the loop's VALU instructions) — keep C/D in VGPRs
EXTRA_FLAGS = {"flash_attn_train": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "gpt2": ["-mllvm", "-amdgpu-mfma-vgpr-form"] + os.environ.get("LMRL_GPT2_EXTRA", "").split(),
               "train_bf16": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "sampler": os.environ.get("LMRL_SAMPLER_EXTRA", "").split()}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build liblmrl_amd.so)")


def _sha(paths, extra: str = "") -> str:
    h = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode() + b"\0" + f.read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False, tools: bool = False) -> str:
    hipcc = _hipcc()
    OBJ = os.path.join(CSRC, "_obj_tools" if tools else "_obj")
    SO = os.path.join(HERE, "liblmrl_amd_tools.so" if tools else "liblmrl_amd.so")
    FLAGS = globals()["FLAGS"] + (["-DLMRL_TOOLS"] if tools else [])
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    headers = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "lmrl_amd.h")]
    ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout
    common = _sha(headers, extra=" ".join(FLAGS) + ver)
    man_path = os.path.join(OBJ, "manifest.json")
    try:
        with open(man_path) as f:
            manifest = json.load(f)
    except Exception:
        manifest = {}
    jobs = []
    objs = []
    digests = {}
    for s in srcs:
        name = os.path.basename(s)[:-4]
        o = os.path.join(OBJ, name + ".o")
        objs.append(o)
        digests[name] = _sha([s], extra=common + " ".join(EXTRA_FLAGS.get(name, [])))
        if force or not os.path.exists(o) or manifest.get(name) != digests[name]:
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(s)[:-4], []) + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        return o

    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    link_digest = hashlib.sha256("".join(digests[k] for k in sorted(digests)).encode()).hexdigest()
    if jobs or force or not os.path.exists(SO) or manifest.get("__so__") != link_digest:
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", SO] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    digests["__so__"] = link_digest
    with open(man_path, "w") as f:
        json.dump(digests, f, indent=1, sort_keys=True)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv, tools="--tools" in sys.argv))
