import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from lmrl_gym_amd.train import ops
world, rank, dev, backend, use_dist = bench._dist_setup(torch)
for rep in range(2):
    for fuse in (True, False):
        ops.FUSE_RESIDUAL = fuse
        r = bench.run_train_step("ilql-step", "bf16", 32, 6, 2, dev, 0, 1, False, "nccl")
        print("fuse" if fuse else "two-launch", r["ms_per_step"], r["last_loss"], flush=True)
