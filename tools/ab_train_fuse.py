"""A/B of the train step's bf16-matmul mode on one box: fused epilogues on / off x tile policy (0 = current, 108 = round 2).
usage: python tools/ab_train_fuse.py [ilql-step|ppo-step]"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from lmrl_gym_amd import _lib
from lmrl_gym_amd.train import ops

mode = sys.argv[1] if len(sys.argv) > 1 else "ilql-step"
world, rank, dev, backend, use_dist = bench._dist_setup(torch)
for rep in range(2):
    for fuse, variant, add_ln, km in ((True, 0, True, True), (True, 0, True, False), (True, 0, False, False), (False, 0, False, False), (False, 108, False, False)):
        ops.FUSE_EPILOGUES = 7 if fuse else 0
        ops.FUSE_ADD_LN = add_ln
        ops.FUSE_KMAJOR_DW = km
        _lib.lib().lmrl_gemm_set_variant(variant)
        r = bench.run_train_step(mode, "bf16", 32, 6, 2, dev, 0, 1, False, "nccl")
        print(f"kmajor_dw={km!s:5s} add_ln={add_ln!s:5s} fused={fuse!s:5s} tiles={'r3' if variant == 0 else 'r2'}  {r['ms_per_step']:7.2f} ms  loss {r['last_loss']}", flush=True)
_lib.lib().lmrl_gemm_set_variant(0)
