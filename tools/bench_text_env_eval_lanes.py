#!/usr/bin/env python
"""`WordleRolloutEngine.text_env_eval(..., concurrent=n)` end to end on the bench workload (GPT-2-small, 1024 envs per batch, steered scripted
guesses, hipGraph replays): env steps returned as host InteractionTransition lists / wall time, for n = 1, 2, 3 episode batches in flight.

    python tools/bench_text_env_eval_lanes.py [--batch 1024] [--batches 12] [--lanes 1 2 3] [--engine bf16|bf16x3|f32]

Every batch stays one lock-step batch of `--batch` envs (the reference's `bsize`); lanes only overlap independent batches (rollout.py:_eval_lanes).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--batches", type=int, default=12)
    ap.add_argument("--lanes", type=int, nargs="+", default=[1, 2, 3])
    ap.add_argument("--engine", default="bf16", choices=["bf16", "bf16x3", "f32"], help="bf16: GPT2Engine (the headline path); bf16x3 / f32: GPT2EngineF32")
    args = ap.parse_args()
    import numpy as np
    import torch
    import lmrl_gym_amd  # noqa: F401
    import bench as BN
    from lmrl_gym_amd.envs import wordle as W
    from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine
    from lmrl_gym_amd.rollout import WordleRolloutEngine
    dev = torch.device("cuda", 0)
    vocab = W.Vocabulary.builtin("wordle_official_400.txt")
    if args.engine == "bf16":
        eng = GPT2Engine.random_init(GPT2Config.gpt2_small(), seed=0, device=dev)
    else:
        from lmrl_gym_amd.gpt2_f32_engine import GPT2EngineF32
        eng = GPT2EngineF32.random_init(GPT2Config.gpt2_small(), seed=0, device=dev, matmul=args.engine)
    B = args.batch
    ro = WordleRolloutEngine(eng, vocab, B, max_new_tokens=6, bad_word_reward=-10.0)
    guesses = torch.from_numpy(BN.scripted_guesses(vocab.all_vocab, args.batches, W.N_TRIES, B, seed=12345).view(np.int32)).to(dev)
    kw = dict(scripted_guesses_fn=lambda bid: guesses[bid % args.batches], steer_strength=30.0, temperature=1.0, sample_seed=9, use_graph=True)
    seeds = iter(range(10 ** 6, 10 ** 9))
    out = {}
    for n in args.lanes:
        ro.text_env_eval(max(n, 1) * B, seed_generator=seeds, concurrent=n, **kw)       # warm: twins, graphs, pinned buffers
        best = None
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            inter, _ = ro.text_env_eval(args.batches * B, seed_generator=seeds, concurrent=n, **kw)
            dt = time.perf_counter() - t0
            steps = sum(len(ep) for ep in inter)
            del inter
            best = dt if best is None else min(best, dt)
        out[f"lanes_{n}"] = dict(env_steps_per_s=round(steps / best, 1), ms_per_batch=round(best * 1e3 / args.batches, 2))
    print(json.dumps(dict(tool="text_env_eval lanes", engine=args.engine, batch=B, batches=args.batches, **out)))
    ro.close()


if __name__ == "__main__":
    main()
