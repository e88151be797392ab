"""What the HF logits warpers cost on the device loop: the bench workload (GPT-2-small, 1024 lock-step Wordle envs, steered sampling) under hipGraph replay
with no warper (fused LM-head Gumbel-max, no logits in HBM), top_k = 40, top_p = 0.95 and both (materialised fp32 logits + radix select per sampled token).
Round 5: warper episodes are graph-capturable (before: eager launches only).  -> profiles/r05_warpers_on_graph.txt
Round 6: every warper form on the candidate path (top-p alone, top_k up to 256), each beside its materialised form (`lmrl_sampler_set_variant(2)`)
-> profiles/r06_warpers_on_graph.txt"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import lmrl_gym_amd  # noqa: E402,F401
from lmrl_gym_amd import _lib  # noqa: E402
from lmrl_gym_amd.envs import wordle as W  # noqa: E402
from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine  # noqa: E402
from lmrl_gym_amd.rollout import WordleRolloutEngine  # noqa: E402

dev = _lib.require_gpu()
vocab = W.Vocabulary.builtin("wordle_official_400.txt")
eng = GPT2Engine.random_init(GPT2Config.gpt2_small(), seed=0, device=dev)
B, n_eps = 1024, 6
g = torch.from_numpy(bench.scripted_guesses(vocab.all_vocab, n_eps, 6, B).view(np.int32)).to(dev)
seeds = torch.arange(n_eps * B, dtype=torch.int64, device=dev).view(n_eps, B)
only = sys.argv[1] if len(sys.argv) > 1 else None          # e.g. "top_k=40": that configuration alone, graph replay only (profiling)
L = _lib.lib()
cases = (("no warper", {}), ("top_k=40", dict(top_k=40)), ("top_p=0.95", dict(top_p=0.95)), ("top_k=40 top_p=0.95", dict(top_k=40, top_p=0.95)),
         ("top_k=128", dict(top_k=128)), ("top_k=256 top_p=0.9", dict(top_k=256, top_p=0.9)))
for name, kw in cases:
    if only is not None and name != only:
        continue
    for graph, variant in (((True, 0),) if only is not None else ((True, 0), (True, 2), (False, 0))):
        if variant == 2 and not kw:
            continue
        L.lmrl_sampler_set_variant(variant)
        ro = WordleRolloutEngine(eng, vocab, B, max_new_tokens=6, bad_word_reward=-10.0)
        kws = dict(temperature=1.0, sample_seed=7, steer_strength=30.0, **kw)
        if graph:
            ro.capture_episode(scripted=True, **kws)
            run = lambda i: ro.replay_episode(seeds[i], g[i])
        else:
            run = lambda i: ro.run_episode(seeds[i], scripted_guesses=g[i], **kws)
        run(0); torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 0
        for i in range(1, n_eps):
            run(i); n += int(ro.traj["n_steps"].sum().item())
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        form = "candidate path" if variant == 0 and kw else ("materialised   " if kw else "fused Gumbel   ")
        print(f"{name:22s} {form} {'hipGraph replay' if graph else 'eager launches ':16s} {dt * 1e3 / (n_eps - 1):7.2f} ms per 1024-env episode  {n / dt / 1e3:7.1f} k env-steps/s", flush=True)
        ro.close()
        L.lmrl_sampler_set_variant(0)
