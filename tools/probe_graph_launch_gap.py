#!/usr/bin/env python
"""How long does one kernel node of a hipGraph take when the kernel does (almost) nothing?  A chain of N dependent one-element adds, and a chain
of N dependent adds over 256 x 512 x 4 floats (one wave per SIMD on every CU), replayed from a graph: µs per node = the floor a dependent launch
chain pays per kernel (dispatch + wave start + drain), which is what the rollout's ~3 400-launch episode is made of.

    python tools/probe_graph_launch_gap.py [--n 2000]
"""
import argparse
import json
import time

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2000)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    out = {}
    for name, numel in (("one_element", 1), ("one_wave_per_simd", 256 * 4 * 64 * 4), ("eight_waves_per_cu_x4", 256 * 512 * 4 * 4)):
        x = torch.zeros(numel, device=dev)
        for _ in range(3):
            x.add_(1.0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(args.n):
                x.add_(1.0)
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        out[name] = round(best * 1e6 / args.n, 3)
    print(json.dumps(dict(probe="hipGraph dependent-chain node cost, us per kernel", n=args.n, **out)))


if __name__ == "__main__":
    main()
