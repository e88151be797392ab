#!/usr/bin/env python
"""RCCL all-reduce probe for the ONE data-path collective of the train step (ilql/gpt2/interface.py:292-324):
an in-place SUM all-reduce of the ILQL step's fp32 gradient arenas (815 MB per rank: base 498 MB + two Q heads 157 MB each + V head), handed
to RCCL in >= 64 MB slices exactly as `lmrl_gym_amd.dist.GradReducer` does.  Prints ONE JSON line (rank 0):

    python tools/rccl_allreduce_probe.py --gpus 8 [--mb 815] [--slice-mb 64] [--iters 10] [--wire f32|bf16] [--debug]

  * `algbw_GBs`  = buffer bytes / time;  `busbw_GBs` = algbw x 2 (n - 1) / n  (the rccl-tests convention: per-link traffic of a ring)
  * per slice size (64 MB slices, async, one wait at the end — what the train step does — and the whole buffer as ONE call)
  * `--wire bf16`: the optional compressed wire format of dist.set_grad_compression("bf16") (cast + all-reduce + widen, all inside the timing)
  * `--debug`: NCCL_DEBUG=INFO with the TUNING / COLL subsystems into gpurun_out/rccl_probe_rank<r>.log; the lines naming the chosen
    algorithm / protocol / channel count are echoed in the JSON (`rccl_log_excerpt`)
xGMI on MI355X is point-to-point (7 links x ~153 GB/s per GPU): a ring is per-link bound (busbw <= ~153 GB/s), direct reduce-scatter +
all-gather over all 7 links can reach several times that; this probe says which one the installed RCCL picks for this message size.
Refuses to run N ranks on fewer than N GPUs (same rule as bench.py).  With --gpus 1 it runs a 1-rank group (identity collective): plumbing check.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--mb", type=float, default=815.0, help="fp32 gradient bytes per rank, MB (ILQL GPT-2-small step: 815)")
    ap.add_argument("--slice-mb", type=float, default=64.0)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--wire", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--debug", action="store_true")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ:
        import torch
        if torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"[rccl probe] {args.gpus} ranks need {args.gpus} GPUs, {torch.cuda.device_count()} visible")
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.debug:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            env.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,COLL,TUNING,GRAPH", NCCL_DEBUG_FILE=os.path.join(ROOT, "gpurun_out", "rccl_probe_rank%h_%p.log"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    import torch
    import torch.distributed as dist
    import lmrl_gym_amd  # noqa: F401
    from lmrl_gym_amd import dist as D
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    n = int(args.mb * 1e6 / 4)
    flat = torch.full((n,), float(rank + 1), dtype=torch.float32, device=dev)
    step = max(1, int(args.slice_mb * 2 ** 20 / 4))
    D.set_grad_compression(args.wire)

    def sliced():
        works = []
        for lo in range(0, n, step):
            D._reduce_grad_slice(flat[lo:lo + step], None, works)
        D._finish_works(works)

    def whole():
        works = []
        D._reduce_grad_slice(flat, None, works)
        D._finish_works(works)

    def timed(fn):
        fn(); fn()
        dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            fn()
        torch.cuda.synchronize(); dist.barrier()
        t = torch.tensor([(time.perf_counter() - t0) / args.iters], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # correctness first: one reduction of rank-constant buffers must give sum(1..world)
    flat.fill_(float(rank + 1)); whole()
    expect = world * (world + 1) / 2
    ok = bool((flat == expect).all())
    res = {}
    for name, fn in (("sliced", sliced), ("whole", whole)):
        flat.fill_(1e-3)
        s = timed(fn)
        alg = n * 4 / s / 1e9
        res[name] = dict(ms=round(s * 1e3, 3), algbw_GBs=round(alg, 1), busbw_GBs=round(alg * 2 * (world - 1) / max(world, 1), 1))
    if rank == 0:
        excerpt = None
        if args.debug:
            import glob
            lines = []
            for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "rccl_probe_rank*.log")))[:1]:
                for ln in open(f, errors="replace"):
                    if any(k in ln for k in ("Algo", "algo", "Proto", "proto", "Channel", "channels", "Ring", "Tree", "xGMI", "XGMI", "P2P")):
                        lines.append(ln.strip()[:240])
            excerpt = lines[:40]
        try:
            ver = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:
            ver = None
        print(json.dumps(dict(probe="rccl all-reduce, in place, fp32 gradient arena", world_size=dist.get_world_size(), backend=dist.get_backend(),
                              rccl_version=ver, bytes_per_rank=n * 4, slice_bytes=step * 4, wire=args.wire, iters=args.iters, sum_correct=ok,
                              sliced_async_like_the_train_step=res["sliced"], whole_buffer_one_call=res["whole"], rccl_log_excerpt=excerpt)), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
