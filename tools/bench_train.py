"""Train-step micro-benchmark at the sizes SURVEY.md §8(d) names:
M3 ILQL step  GPT-2-small, B=32, T=512   (train_ilql_gpt2.py:58,65)
M4 PPO  step  GPT-2-small, B=32, T=1024  (train_ppo_gpt2.py:74-75)
Synthetic ids ~ U[0, 50257), 6-on/6-off action pattern after a 4-token header, random-init weights.
Prints ms per optimizer step and the model-flop rate (6·N·tokens for fwd+bwd, +2·N·tokens per extra forward).
usage: python tools/bench_train.py [ilql|ppo|both] [--steps K] [--bsize B]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import lmrl_gym_amd  # noqa: F401
from lmrl_gym_amd import _lib
from lmrl_gym_amd.algorithms import ilql, ppo
from lmrl_gym_amd.gpt2 import GPT2Config, init_hf_style_state_dict
from lmrl_gym_amd.train.gpt2_f32 import GPT2F32, LinearHeadF32, MLPHeadF32


def make_batch(rng, B, T, vocab):
    ids = rng.randint(0, vocab, size=(B, T)).astype(np.int32)
    sta = np.zeros((B, T - 1), dtype=bool)
    t = np.arange(T - 1)
    sta[:, :] = ((t >= 4) & (((t - 4) // 6) % 2 == 0))[None, :]
    return ids, sta


def timed(fn, steps, warmup=1):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="?", default="both")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--bsize", type=int, default=32)
    a = ap.parse_args()
    dev = _lib.require_gpu()
    cfg = GPT2Config.gpt2_small(50258)
    pad = 50257
    rng = np.random.RandomState(0)
    n_params = 124.4e6
    if a.which in ("ilql", "both"):
        B, T = a.bsize, 512
        sd = init_hf_style_state_dict(cfg, seed=0)
        base, tbase = GPT2F32(sd, cfg.n_head, device=dev), GPT2F32(sd, cfg.n_head, device=dev)
        d, V = cfg.d_model, cfg.vocab
        g = torch.Generator().manual_seed(1)
        mk = lambda out, b2: MLPHeadF32({"dense1.kernel": torch.randn(d, d, generator=g) * 0.02, "dense1.bias": torch.zeros(d),
                                         "dense2.kernel": torch.zeros(d, out), "dense2.bias": torch.full((out,), b2)}, dev)
        tr = ilql.GPT2ILQLTrain(base, mk(V, -4.4), mk(V, -4.4), mk(1, -4.4), pad, dict(gamma=0.99, tau=0.7, cql_weight=0.01),
                                target_base=tbase, lr=3e-5)
        ids, sta = make_batch(rng, B, T, 50257)
        rewards = np.where(sta & ~np.roll(sta, -1, axis=1), -1.0, 0.0).astype(np.float32)
        dones = (rng.rand(B) < 0.5).astype(np.float32)
        dt = timed(lambda: tr.step(ids, sta, rewards, dones), a.steps)
        tok = B * T
        head_flops = 2 * (d * d + d * V) * tok * (3 * 2 + 2)            # q1,q2 fwd+bwd (3x) and two target-head forwards
        flops = (6 + 2) * n_params * tok + head_flops
        print("ILQL step  B=%d T=%d : %.1f ms/step  %.1f k tokens/s  %.1f TFLOP/s (fp32 MFMA peak 157)" %
              (B, T, dt * 1e3, tok / dt / 1e3, flops / dt / 1e12))
        del tr, base, tbase
        torch.cuda.empty_cache()
    if a.which in ("ppo", "both"):
        B, T = a.bsize, 1024
        sd = init_hf_style_state_dict(cfg, seed=0)
        pol = GPT2F32(sd, cfg.n_head, device=dev)
        head = LinearHeadF32(dict(kernel=torch.randn(cfg.d_model, 1) * 0.01, bias=torch.tensor([-4.1])), dev)
        tr = ppo.GPT2PPOTrain(pol, head, pad, dict(cliprange_value=0.2, cliprange=0.2, value_loss_coef=1.0), lr=1e-5)
        ids, sta = make_batch(rng, B, T, 50257)
        f = lambda s: (rng.randn(B, T - 1) * s).astype(np.float32)
        olp, ov, oa, orr = f(0.1) - 10.8, f(1), f(1), f(1)
        dt = timed(lambda: tr.step(ids, sta, olp, ov, oa, orr), a.steps)
        tok = B * T
        flops = 6 * n_params * tok + 12 * 6 * 2 * T * cfg.d_model * tok   # + attention score/PV matmuls
        print("PPO  step  B=%d T=%d : %.1f ms/step  %.1f k tokens/s  %.1f TFLOP/s (fp32 MFMA peak 157)" %
              (B, T, dt * 1e3, tok / dt / 1e3, flops / dt / 1e12))


if __name__ == "__main__":
    main()
