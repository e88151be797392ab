"""M5 micro-benchmark: the GAE / reward-to-go / whitening kernels on token chains of the rollout's shape — the `rl_reduce` leg of
bench.py (`bench.run_rl_reduce`: rotating buffer sets, one hipGraph replay of 40 launches per kernel, algorithmic bytes) at 4096 / 65536 / 1 M chains.
Counters + rocprofv3 summary of the same leg: tools/prof_rl_reduce.sh -> profiles/r05_rl_reduce*.  Round 1's table: profiles/r01_rl_reduce_microbench.txt."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    out = bench.run_rl_reduce(torch.device("cuda", 0), chains=(4096, 65536, 1048576), budget_bytes=900 << 20)
    for b, r in out["chains"].items():
        print("chains %8s: " % b + " | ".join("%s %7.2f us %7.1f GB/s (%.3f)" % (k, v["avg_launch_us"], v["achieved"], v["frac"]) for k, v in r.items() if isinstance(v, dict)))
    print(json.dumps(out["chains"]))
