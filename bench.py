#!/usr/bin/env python
"""bench.py — headline metric of BASELINE.json: env-steps/sec, GPT-2-small Wordle rollouts, 1024 envs per GPU.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one lock-step EPISODE of the hot path over one batch: 1024 envs x up to 6 turns, each turn =
GPT-2-small policy samples an action token by token (persistent KV cache, fused LM-head sampler), the batched
Wordle kernel steps the envs, the observation is injected as tokens.  `value` = env.step calls completed by all
ranks / wall time of the K timed steps (barrier + synchronize on both sides, max over ranks).

Synthetic workload (BASELINE.md M2'): random-init GPT-2-small (HF init, seed 0), bf16 weights/activations with
fp32 accumulation and fp32 residual stream, temperature 1, full-vocabulary Gumbel-max sampling.  A random-init
policy never spells a word, so each env's sampler is STEERED towards a scripted guess (uniform over the 431-word
vocabulary, 10 % non-words; +30 on that token's logit): every logit is still computed and sampled from, but the
episode then has the token mix of a trained policy (valid guesses, 6-token observations, early wins).

Multi-GPU: rollouts shard by env with no data-path collective -> weak scaling, one process per GPU.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Kernel whose HIP-event time is reported as the roofline line: the decode attention (attention_decode_kernel, the single-token attention
# that streams the KV cache) — the largest HBM-bound kernel of the step and the one the north star's ">= 60 % of HBM roofline" refers to.
ROOFLINE_TAG = "attention_decode"
ROOFLINE_KERNEL = "attention_decode_kernel"                 # its name in the rocprofv3 / PMC summaries
PROFILE_STATS = "profiles/r06_bench_kernel_stats.csv"       # committed `rocprofv3 --kernel-trace --stats` summary of this command
PROFILE_PMC = "profiles/r06_pmc_fetch_write.json"           # committed FETCH_SIZE / WRITE_SIZE passes (tools/pmc_fetch_write.sh)
SECONDARY_TAGS = ["gemm_bf16_64x64", "gemm_bf16_64x128", "gemm_bf16_128x128", "lm_head_sample", "attention_chunk"]
MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0   # /opt/skills/guides/MI355X_MICROARCH.md (dense, no sparsity)
HBM_PEAK_GBS = 8000.0                  # same guide: 8 TB/s spec (6.3 TB/s measured achievable)


def csrc_digest():
    """sha256 over every kernel source / header of the library (lmrl-gym_amd/csrc/*.hip, *.h): ties a committed PMC summary to the code it
    was collected on (tools/pmc_fetch_write.sh records it as __meta__.csrc_digest)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "lmrl-gym_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(base, "*.hip")) + glob.glob(os.path.join(base, "*.h"))):
        h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return h.hexdigest()


def scripted_guesses(vocab_words, n_eps, n_turns, batch, seed=12345):
    """[n_eps][n_turns][batch] packed guesses: uniform over the vocabulary from MT19937(seed), 10 % non-words."""
    from lmrl_gym_amd.envs import wordle as W
    rng = np.random.RandomState(seed)
    packed = np.array([W.pack_guess(w) for w in vocab_words], dtype=np.uint32)
    g = packed[rng.randint(0, len(packed), size=(n_eps, n_turns, batch))]
    bad = rng.rand(n_eps, n_turns, batch) < 0.10
    junk = rng.randint(0, 26, size=(n_eps, n_turns, batch, 5)).astype(np.uint32)
    junk_packed = sum(junk[..., i] << np.uint32(5 * i) for i in range(5)).astype(np.uint32)
    g[bad] = junk_packed[bad]
    return g


def _cpu_rollout(vocab_words, n_envs, max_turns, budget_s):
    """The reference's per-turn structure on the host cores: re-prefill the whole history, then decode with a KV cache (HF PyTorch
    GPT2LMHeadModel as the port of the JAX model), env = the C oracle.  Returns (env steps, seconds, turns run)."""
    import torch
    import transformers
    from oracle.wordle import OracleWordleEnv
    from lmrl_gym_amd.rollout import WordleTokenTable
    transformers.logging.set_verbosity_error()
    torch.manual_seed(0)
    model = transformers.GPT2LMHeadModel(transformers.GPT2Config()).eval()
    tab = WordleTokenTable.default_gpt2()
    rng = np.random.RandomState(1)
    envs = [OracleWordleEnv(vocab_words, True, -10.0) for _ in range(n_envs)]
    hist = [e.reset(i) for i, e in enumerate(envs)]
    n_steps = 0
    t0 = time.perf_counter()
    with torch.no_grad():
        for turn in range(max_turns):
            words = [vocab_words[k] for k in rng.randint(0, len(vocab_words), size=n_envs)]
            ids = torch.tensor([tab.encode_text("".join(t for t, _ in h)) for h in hist])
            out = model(ids, use_cache=True)
            past, logits = out.past_key_values, out.logits[:, -1]
            for k in range(6):
                steer = torch.tensor([(tab.encode_text(" ".join(w) + "\n"))[k] for w in words])
                logits[torch.arange(n_envs), steer] += 30.0
                tok = torch.multinomial(torch.softmax(logits.float(), -1), 1)
                if k < 5:
                    out = model(tok, past_key_values=past, use_cache=True)
                    past, logits = out.past_key_values, out.logits[:, -1]
            for i, e in enumerate(envs):
                hist[i], r, d = e.step(hist[i] + ((" ".join(words[i]) + "\n", True),))
                n_steps += 1
            if time.perf_counter() - t0 > budget_s:
                break
    return n_steps, time.perf_counter() - t0, turn + 1


def cpu_baseline(vocab_words, budget_s=14.0):
    """the CPU path timed beside the GPU number — (1) the rollout port on many host threads at 32 envs (the faster of all / 16
    threads: `value`, `cores`), (2) the same port at 256 and 1024 envs, 2 turns each (CPU GEMM efficiency rises with the batch: the 32-env point
    understates what the host does at the metric's batch — VERDICT r04 weak #12), (3) on ONE thread, (4) the env alone (C oracle, no LM) on one
    thread and on all cores (OpenMP over the independent envs).  Bounded samples, ~30 s in total."""
    import torch
    from oracle.wordle import run_scripted_timed
    n_all = torch.get_num_threads()
    # all host threads is not always the fastest configuration for 32 short sequences (oversubscription on 128-thread hosts): the multi-thread
    # point is the best of {all threads, 16 threads}; `cores` reports the count it was measured with
    runs = {}
    for n_thr in sorted({n_all, min(16, n_all)}, reverse=True):
        torch.set_num_threads(n_thr)
        runs[n_thr] = _cpu_rollout(vocab_words, 32, 6, budget_s * 0.3)
    n_best = max(runs, key=lambda k: runs[k][0] / runs[k][1])
    s_all, t_all, turns_all = runs[n_best]
    by_batch = {}
    torch.set_num_threads(n_all)
    # a turn cannot be interrupted: the larger batches run only when the time of the smaller one predicts they stay inside the leg's bound
    st, tt, tu = _cpu_rollout(vocab_words, 256, 2, 2.5) if t_all / max(turns_all, 1) * 8 <= 12.0 else (0, 0.0, 0)
    if tu:
        by_batch["256"] = dict(value=round(st / tt, 2), unit="env-steps/s", cores=n_all, sample=f"256 envs x {tu} turn(s), same path; {tt:.1f} s")
        per_turn_1024 = tt / tu * 4.0
        if per_turn_1024 <= 10.0:
            st, tt, tu = _cpu_rollout(vocab_words, 1024, 2, 4.0)
            by_batch["1024"] = dict(value=round(st / tt, 2), unit="env-steps/s", cores=n_all, sample=f"1024 envs x {tu} turn(s), same path; {tt:.1f} s")
        elif per_turn_1024 <= 30.0:      # the metric's own batch, ONE lock-step turn (a turn cannot be interrupted; ~10-30 s of CPU work is the leg's bound)
            st, tt, tu = _cpu_rollout(vocab_words, 1024, 1, 0.0)
            by_batch["1024"] = dict(value=round(st / tt, 2), unit="env-steps/s", cores=n_all, sample=f"1024 envs x {tu} turn, same path; {tt:.1f} s")
        else:
            by_batch["1024"] = dict(value=None, skipped=f"one 256-env turn took {tt / tu:.1f} s on this host: a 1024-env turn (~{per_turn_1024:.0f} s) would not fit the leg's bound")
    else:
        by_batch["256"] = dict(value=None, skipped=f"one 32-env turn took {t_all / max(turns_all, 1):.1f} s on this host: larger batches would not fit the leg's bound")
    torch.set_num_threads(1)
    try:
        s_one, t_one, turns_one = _cpu_rollout(vocab_words, 16, 2, budget_s * 0.25)
    finally:
        torch.set_num_threads(n_all)
    rng = np.random.RandomState(2)
    gi = rng.randint(0, len(vocab_words), size=(6, 2048))
    env_steps, te, _ = run_scripted_timed(vocab_words, 2048, gi, threads=1)
    n_mt = 65536
    gi_mt = rng.randint(0, len(vocab_words), size=(6, n_mt))
    mt_runs = {}
    for thr in sorted({0, min(os.cpu_count() or 1, 32), min(os.cpu_count() or 1, 8)}):
        st, tt, used = run_scripted_timed(vocab_words, n_mt, gi_mt, threads=thr)
        mt_runs[used] = st / tt
    used_best = max(mt_runs, key=mt_runs.get)
    # `value` = the host's BEST measured point (any batch, any thread count tried): the fairest single number to put beside the GPU's
    points = [(s_all / t_all, n_best, f"32 envs x {turns_all} turns; {t_all:.1f} s")] + \
             [(v["value"], v["cores"], v["sample"]) for v in by_batch.values() if v.get("value")]
    best_value, best_cores, best_sample = max(points, key=lambda x: x[0])
    return dict(value=best_value, unit="env-steps/s", cores=best_cores, kind="port",
                by_threads={str(k): round(v[0] / v[1], 2) for k, v in runs.items()},
                sample=f"{best_sample} (valid scripted guesses), GPT-2-small fp32 on torch-CPU re-prefilling the history every "
                       f"turn as the reference does + C oracle env; the best of the batch / thread points below",
                value_32_envs=round(s_all / t_all, 2), by_batch=by_batch, best_value_any_batch=round(best_value, 2),
                one_thread=dict(value=s_one / t_one, cores=1, sample=f"16 envs x {turns_one} turns, same path; {t_one:.1f} s"),
                env_only=dict(value=env_steps / te, unit="env-steps/s", cores=1,
                              sample=f"C oracle env alone (no LM), 2048 envs x 6 scripted steps, one thread, stepping loop only; {te:.2f} s",
                              all_cores=dict(value=round(mt_runs[used_best], 1), unit="env-steps/s", cores=used_best,
                                             by_threads={str(k): round(v, 1) for k, v in mt_runs.items()},
                                             sample=f"the same C oracle env, {n_mt} envs x 6 scripted steps, OpenMP over the envs (best of the thread counts tried)"),
                              reference_python_env=dict(value_v431=204.0, value_v2315=45.0, unit="env-steps/s", cores=1, kind="reference",
                                                        note="the reference's OWN Python env (llm_rl_scripts/wordle/env, no LM) as timed by the survey "
                                                             "session in the build container (BASELINE.md section 2); it cannot travel to the GPU box")))


def gpu_env_only(vocab, dev):
    """M1: the batched Wordle step kernel alone — scripted guesses (10 % non-words), reset + 6 steps per episode, no LM — at the
    bench's 1024 envs and at 262 144 envs (where the lane-per-env kernel fills the chip).  The kernel's bound is NOT the 96 B / env-step of
    state traffic BASELINE.md's table assumes (0.02 of HBM peak at best) but the O(V) vocabulary filter: ~100 VALU instructions per vocabulary
    word per valid step (two consistency sweeps + the membership sweep; counted in the gfx950 ISA of wordle_step_lanes_kernel), i.e.
    `valu_frac` = env-steps/s x V x 100 / 64 lanes / (256 CU x 4 SIMD x 1.2 G wave-instructions/s) of the VALU issue peak (lane = env:
    one wave-instruction serves 64 envs)."""
    import torch
    from lmrl_gym_amd.envs import wordle as W
    packed = np.array([W.pack_guess(w) for w in vocab.all_vocab], dtype=np.uint32)
    out = {}
    for n in (1024, 262144):
        env = W.VectorWordleEnv(vocab, True, -10.0)
        env._alloc(n)
        rng = np.random.RandomState(12345)
        g = packed[rng.randint(0, len(packed), size=(6, n))]
        g[rng.rand(6, n) < 0.1] = W.BAD_GUESS
        gd = torch.from_numpy(g.view(np.int32)).to(dev)
        seeds = np.arange(n, dtype=np.uint64)
        env.reset_device(seeds)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        t_reset = t_step = 0.0
        reps = 10
        for _ in range(reps):
            ev[0].record(); env.reset_device(seeds); ev[1].record()
            for t in range(6):
                env.step_device(gd[t], None)
            ev[2].record(); torch.cuda.synchronize()
            t_reset += ev[0].elapsed_time(ev[1]); t_step += ev[1].elapsed_time(ev[2])
        rate = 6 * n / (t_step / reps / 1e3)
        out[str(n)] = dict(env_steps_per_s_steps_only=round(rate, 0), env_steps_per_s_incl_reset=round(6 * n / ((t_reset + t_step) / reps / 1e3), 0),
                           valu_frac=round(rate * 0.9 * len(packed) * 100.0 / 64.0 / (256 * 4 * 1.2e9), 3))
        env.close()
    return dict(unit="env-steps/s", vocab_words=len(packed), envs=out, bound="valu",
                note="Wordle step kernel alone on this GPU (reset + 6 scripted steps, 10 % non-words skip the sweeps: x 0.9 in valu_frac); "
                     "bound = VALU issue (O(V) vocabulary filter), not HBM: DESIGN.md section 4")


def _resolve_backend(world, n_dev, explicit):
    """Which torch.distributed backend an N-rank bench run uses.  RCCL ("nccl") needs one GPU per rank.  A run with more ranks than GPUs can only
    work on gloo with ranks SHARING GPUs — a launch-path test, never a scaling measurement — so it is refused (SystemExit, non-zero) unless the
    caller asks for it by name with LMRL_BENCH_BACKEND=gloo: on a mis-provisioned 8-GPU lease the N = 8 line must fail, not quietly become a
    gloo number."""
    if explicit:
        if explicit not in ("nccl", "gloo"):
            raise SystemExit(f"[bench] LMRL_BENCH_BACKEND={explicit!r}: must be 'nccl' (RCCL) or 'gloo'")
        return explicit
    if world > 1 and n_dev < world:
        raise SystemExit(f"[bench] refusing to run {world} ranks on {n_dev} visible GPU(s): RCCL needs one GPU per rank, and a gloo run with ranks sharing "
                         "GPUs is not a multi-GPU measurement.  Set LMRL_BENCH_BACKEND=gloo explicitly for a launch-path test.")
    return "nccl"


def _dist_setup(torch):
    """(world, rank, dev, backend, use_dist) from the launcher's environment; initialises the process group when world > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    n_dev = max(torch.cuda.device_count(), 1)
    backend = _resolve_backend(world, n_dev, os.environ.get("LMRL_BENCH_BACKEND"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) % n_dev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("LMRL_BENCH_FORCE_DIST") == "1"   # FORCE_DIST: 1-rank RCCL group, launch-path test
    if use_dist:
        import torch.distributed as dist
        if not dist.is_initialized():
            dist.init_process_group("nccl", device_id=dev) if backend == "nccl" else dist.init_process_group(backend)
    return world, rank, dev, backend, use_dist


def _dist_report(torch, dev, backend, use_dist):
    """What the process group itself says about this run, for the JSON line of an N > 1 run (so the first 8-GPU record verifies itself):
    the backend torch.distributed reports, ITS world size, every rank's device (index, name, PCI bus id, uuid when the runtime exposes them)
    gathered over the group, and the RCCL version.  With RCCL the ranks' devices must be pairwise distinct — checked here, on every rank."""
    import torch.distributed as dist
    if not (use_dist and dist.is_initialized()):
        return None
    p = torch.cuda.get_device_properties(dev)
    mine = dict(rank=dist.get_rank(), local_rank=int(os.environ.get("LOCAL_RANK", "0")), device_index=dev.index, name=p.name,
                pci_bus_id=getattr(p, "pci_bus_id", None), uuid=str(getattr(p, "uuid", "")) or None, hbm_gb=round(p.total_memory / 2 ** 30, 1),
                visible=os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES"))
    ranks = [None] * dist.get_world_size()
    dist.all_gather_object(ranks, mine)
    be = dist.get_backend()
    if be == "nccl":
        ids = [(r["device_index"], r["pci_bus_id"], r["uuid"]) for r in ranks]
        if len(set(ids)) != len(ids):
            raise SystemExit(f"[bench] RCCL group with ranks sharing a device: {ids}")
    try:
        ver = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:
        ver = None
    return dict(backend=be, backend_requested=backend, world_size=dist.get_world_size(), rccl_version=ver if be == "nccl" else None,
                torch=torch.__version__, hip=getattr(torch.version, "hip", None), rank_devices=ranks,
                ranks_share_devices=len({r["device_index"] for r in ranks}) < len(ranks))


def run_train_step(mode, matmul, B, steps, warmup, dev, rank, world, use_dist, backend, measure_exposed=True, model="small"):
    """One train-step measurement (`mode` = "ilql-step" | "ppo-step") at the sizes SURVEY.md §8d names (M3: GPT-2-small, B = 32 x T = 512 per GPU;
    M4: B = 32 x T = 1024), synthetic ids ~ U[0, 50257) with the 6-on / 6-off action pattern after a 4-token header.  N > 1: pure data
    parallelism, every rank its own B sequences (weak scaling), ONE gradient all-reduce per step overlapped with the backward pass
    (lmrl_gym_amd.dist.GradReducer) — ilql/gpt2/interface.py:292-324.  Returns a dict (ms_per_step, executed flops, all-reduce bytes and, for
    N > 1, the EXPOSED all-reduce time = step with the collective - the same step without it)."""
    import torch
    import lmrl_gym_amd  # noqa: F401
    from lmrl_gym_amd import dist as D
    from lmrl_gym_amd.algorithms import ilql, ppo
    from lmrl_gym_amd.gpt2 import GPT2Config, init_hf_style_state_dict
    from lmrl_gym_amd.train.gpt2_f32 import GPT2F32, LinearHeadF32, MLPHeadF32
    # +1 added <|pad|> token (train_ilql_gpt2.py:115-116); "medium" = configs[3]'s policy (chess/ppo/train_ppo_gpt2_online.py:201-222)
    cfg = dict(small=GPT2Config.gpt2_small, medium=GPT2Config.gpt2_medium)[model](50258)
    pad = 50257
    T = 512 if mode == "ilql-step" else 1024
    rng = np.random.RandomState(1000 + rank)
    ids = rng.randint(0, 50257, size=(B, T)).astype(np.int32)
    t = np.arange(T - 1)
    sta = np.broadcast_to(((t >= 4) & (((t - 4) // 6) % 2 == 0))[None, :], (B, T - 1)).copy()
    sd = init_hf_style_state_dict(cfg, seed=0)
    d, V = cfg.d_model, cfg.vocab
    tok = B * T
    mmode = matmul
    if mode == "ilql-step":
        base, tbase = GPT2F32(sd, cfg.n_head, device=dev, matmul=mmode), GPT2F32(sd, cfg.n_head, device=dev, matmul=mmode)
        g = torch.Generator().manual_seed(1)
        mk = lambda out, b2: MLPHeadF32({"dense1.kernel": torch.randn(d, d, generator=g) * 0.02, "dense1.bias": torch.zeros(d),
                                         "dense2.kernel": torch.zeros(d, out), "dense2.bias": torch.full((out,), b2)}, dev, matmul=mmode)
        tr = ilql.GPT2ILQLTrain(base, mk(V, -4.4), mk(V, -4.4), mk(1, -4.4), pad, dict(gamma=0.99, tau=0.7, cql_weight=0.01), target_base=tbase, lr=3e-5)
        rewards = np.where(sta & ~np.roll(sta, -1, axis=1), -1.0, 0.0).astype(np.float32)
        dones = (rng.rand(B) < 0.5).astype(np.float32)
        step = lambda: tr.step(ids, sta, rewards, dones)
        # q1, q2 forward + backward (3x each) on the rows the loss reads (should_take_action x attention mask: GPT2ILQLTrain.compact_q_rows — the
        # flops actually executed, not the dense B*T count); the two target heads: dense1 forward + ONE column of dense2 per token
        q_tok = int(sta.sum())          # no padding in the synthetic batch: attention_mask = 1
        head_flops = 2 * (d * d + d * V) * q_tok * (3 * 2) + 2 * (d * d + d) * tok * 2
        # matmul parameters of the transformer blocks only: the embedding tables do no flops and ILQL never forms LM logits
        n_mm = cfg.n_layer * (4 * d * d + 2 * d * cfg.d_ff)
        flops = (6 + 2) * n_mm * tok + head_flops + cfg.n_layer * 6 * 2 * T * d * tok
        workload = f"configs[2] / M3: ILQL train step, GPT-2-{model} fp32, B={B} x T={T} per GPU (train_ilql_gpt2.py:55-110), target base + 2 Q heads + V head; should_take_action on {100.0 * q_tok / tok:.0f} % of the tokens"
    else:
        pol = GPT2F32(sd, cfg.n_head, device=dev, matmul=mmode)
        head = LinearHeadF32(dict(kernel=torch.randn(d, 1) * 0.01, bias=torch.tensor([-4.1])), dev)
        tr = ppo.GPT2PPOTrain(pol, head, pad, dict(cliprange_value=0.2, cliprange=0.2, value_loss_coef=1.0), lr=1e-5)
        f = lambda s_: (rng.randn(B, T - 1) * s_).astype(np.float32)
        olp, ov, oa, orr = f(0.1) - 10.8, f(1), f(1), f(1)
        step = lambda: tr.step(ids, sta, olp, ov, oa, orr)
        # the tied LM head (V x d of the 124.4 M parameters) runs on the rows the loss reads only (GPT2PPOTrain.compact_rows): executed flops
        q_tok = int(sta.sum())
        n_mm = cfg.n_layer * (4 * d * d + 2 * d * cfg.d_ff)          # transformer-block matmul parameters (embedding lookups do no flops)
        flops = 6 * n_mm * tok + 6 * V * d * q_tok + cfg.n_layer * 6 * 2 * T * d * tok
        workload = (f"M4: PPO train step, GPT-2-{model} fp32, B={B} x T={T} per GPU (train_ppo_gpt2.py:60-112), LinearHead value function; "
                    f"should_take_action on {100.0 * q_tok / tok:.0f} % of the tokens")

    def barrier():
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed(n):
        barrier()
        t0 = time.perf_counter()
        loss = None
        for _ in range(n):
            _, loss, _ = step()
        barrier()
        dt = time.perf_counter() - t0
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        if use_dist:
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        return float(tt.item()), float(loss)

    for _ in range(max(warmup, 1)):
        step()
    dt, loss = timed(steps)
    out = dict(mode=mode, matmul=matmul, per_gpu_batch=B, seq_len=T, steps=steps, warmup=max(warmup, 1), ms_per_step=round(dt * 1e3 / steps, 2),
               sequences_per_s=round(world * B * steps / dt, 2), executed_tflops_per_gpu=round(flops / (dt / steps) / 1e12, 1),
               last_loss=loss, workload=workload, flops_per_step_per_gpu=flops)
    bf = matmul == "bf16"
    peak = MFMA_BF16_DENSE_PEAK_TFLOPS if bf else 157.3
    out["mfma_peak_tflops"] = peak
    out["frac"] = round(out["executed_tflops_per_gpu"] / peak, 4)
    if world > 1:
        out["allreduce_bytes_per_step_per_rank"] = int(D.LAST_REDUCE_BYTES)
        out["grad_allreduce_dtype"] = D.grad_compression() or "f32"
        out["allreduce"] = (f"one in-place SUM all-reduce of the gradient arenas per step ({backend}; wire dtype {D.grad_compression() or 'f32'}), "
                            ">=64 MB slices overlapped with the backward pass")
        if measure_exposed:
            D.set_grad_reduce(False)                 # timing only: same kernels, no data-path collective (ranks diverge — nothing reads the result)
            try:
                step()
                dt_off, _ = timed(steps)
            finally:
                D.set_grad_reduce(True)
            out["ms_per_step_without_allreduce"] = round(dt_off * 1e3 / steps, 2)
            out["allreduce_exposed_ms"] = round((dt - dt_off) * 1e3 / steps, 2)
    del tr, step
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out


def run_rl_reduce(dev, chains=(4096, 65536), L=96, budget_bytes=640 << 20, iters=40):
    """M5 (north_star's "per-token reward-to-go / GAE scan ... as wavefront shuffle kernels ... rocprof HBM GB/s"): `lmrl_gae`,
    `lmrl_rtg`, `lmrl_whiten_moments` + `lmrl_whiten_apply` on token chains of the rollout's shape (L = 96 slots, 40..96 of them used, 6-on /
    6-off action runs after a 4-token header), timed with HIP events over `iters` launches that ROTATE through enough independent buffer sets to
    exceed the 256 MB memory-side cache (a loop over one 100 MB set would be served from it).  `achieved` = ALGORITHMIC bytes per launch / average
    launch time: inputs only below each chain's length (values 4 B + rewards 4 B + flag 1 B per used slot, + bootstrap value and length per
    chain), outputs for every slot (adv + ret 8 B; rtg 4 B); whitening: 5 B per element read by each pass + 4 B written by the second.
    Reference code: ppo/base_interface.py:245-293, mc_returns/data.py:10-14."""
    import torch
    import lmrl_gym_amd  # noqa: F401
    from lmrl_gym_amd import _lib
    Lb, sp = _lib.lib(), _lib.stream_ptr
    out = {}
    for B in chains:
        rng = np.random.RandomState(B)
        lens = rng.randint(40, L + 1, size=B).astype(np.int32)
        t = np.arange(L)[None, :]
        sta = ((t >= 4) & (((t - 4) // 6) % 2 == 0) & (t < lens[:, None])).astype(np.uint8)
        n = B * L
        used = int(lens.sum())
        set_bytes = n * (4 + 4 + 1 + 8 + 4 + 4) + B * 8
        n_sets = int(min(64, max(2, -(-budget_bytes // set_bytes))))
        sets = []
        for _ in range(n_sets):
            sets.append(dict(v=torch.randn(B, L + 1, device=dev), r=torch.randn(B, L, device=dev), s=torch.from_numpy(sta).to(dev),
                             ln=torch.from_numpy(lens).to(dev), adv=torch.empty(B, L, device=dev), ret=torch.empty(B, L, device=dev),
                             rtg=torch.empty(B, L, device=dev), y=torch.empty(B, L, device=dev), mom=torch.zeros(3, dtype=torch.float64, device=dev),
                             part=torch.zeros(max(Lb.lmrl_gae_moments_partials(B, L), 1), 3, dtype=torch.float64, device=dev)))
        p = lambda x: x.data_ptr()
        npart = Lb.lmrl_gae_moments_partials(B, L)

        def chain4(d):      # get_advantages_and_returns + whiten as four launches (the round-5 pipeline)
            _lib.check(Lb.lmrl_gae(p(d["v"]), p(d["r"]), p(d["s"]), p(d["ln"]), p(d["adv"]), p(d["ret"]), B, L, 0.99, 0.95, sp()))
            _lib.check(Lb.lmrl_whiten_moments(p(d["adv"]), p(d["s"]), p(d["mom"]), n, sp()))
            _lib.check(Lb.lmrl_whiten_apply(p(d["adv"]), p(d["s"]), p(d["mom"]), p(d["y"]), n, 1, sp()))

        def chain2(d):      # the same as two: the GAE launch leaves the partial moments, the apply launch adds them up itself
            _lib.check(Lb.lmrl_gae_moments(p(d["v"]), p(d["r"]), p(d["s"]), p(d["ln"]), p(d["adv"]), p(d["ret"]), B, L, 0.99, 0.95, p(d["part"]), sp()))
            _lib.check(Lb.lmrl_whiten_apply_partials(p(d["adv"]), p(d["s"]), p(d["part"]), npart, p(d["y"]), n, 1, sp()))
        fns = dict(
            gae=(lambda d: _lib.check(Lb.lmrl_gae(p(d["v"]), p(d["r"]), p(d["s"]), p(d["ln"]), p(d["adv"]), p(d["ret"]), B, L, 0.99, 0.95, sp())),
                 used * 9 + B * 8 + n * 8),
            rtg=(lambda d: _lib.check(Lb.lmrl_rtg(p(d["r"]), p(d["s"]), p(d["ln"]), p(d["rtg"]), B, L, 0.99, sp())), used * 5 + B * 4 + n * 4),
            whiten_moments=(lambda d: _lib.check(Lb.lmrl_whiten_moments(p(d["adv"]), p(d["s"]), p(d["mom"]), n, sp())), n * 5),
            whiten_apply=(lambda d: _lib.check(Lb.lmrl_whiten_apply(p(d["adv"]), p(d["s"]), p(d["mom"]), p(d["y"]), n, 1, sp())), n * 9),
            gae_whiten_4_launches=(chain4, used * 9 + B * 8 + n * 8 + n * 5 + n * 9),
            gae_whiten_2_launches=(chain2, used * 9 + B * 8 + n * 8 + n * 9))
        res = {}
        for name, (fn, nbytes) in fns.items():
            for d in sets[:4]:
                fn(d)
            torch.cuda.synchronize()
            # the `iters` launches as ONE hipGraph replay: a Python / ctypes launch costs ~7 us of host time, more than these kernels run
            st = torch.cuda.Stream(device=dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(st):
                with torch.cuda.graph(graph, stream=st, capture_error_mode="thread_local"):
                    for i in range(iters):
                        fn(sets[i % n_sets])
                graph.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                graph.replay()
                e1.record(st)
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / iters
            del graph
            gbs = nbytes / (us * 1e-6) / 1e9
            res[name] = dict(bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4), avg_launch_us=round(us, 2),
                             algorithmic_bytes_per_launch=int(nbytes), traffic=None)
        # HBM traffic per launch from the committed counter passes of this same leg (tools/prof_rl_reduce.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
        # separate runs; 2 x FETCH + WRITE per the gfx950 correction of the microarch guide) — not measurable from inside the process
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r06_rl_reduce_pmc.json" if os.path.exists(os.path.join(ROOT, "profiles", "r06_rl_reduce_pmc.json"))
                                              else "r05_rl_reduce_pmc.json")))
            pick = {4096: min, 65536: max}.get(B)          # the counter passes ran this leg's default sizes (tools/prof_rl_reduce.sh); other sizes: no counters
            for name, pat in (("gae", "chain_scan_row2_kernel<true, 3, false"), ("rtg", "chain_scan_row2_kernel<false"), ("whiten_moments", "whiten_moments_kernel"),
                              ("whiten_apply", "whiten_apply_kernel")):
                ks = [k for k in pmc if pat in k]
                if ks and pick is not None:
                    k = pick(ks, key=lambda q: int(q.split("grid=")[1]))
                    res[name]["traffic"] = round((2.0 * pmc[k].get("fetch_size_kb_avg", 0.0) + pmc[k].get("write_size_kb_avg", 0.0)) * 1024)
                    res[name]["traffic_source"] = "profiles/r06_rl_reduce_pmc.json (2*FETCH_SIZE + WRITE_SIZE)"
        except Exception:
            pass
        # SURVEY.md §8(d) / BASELINE.md count 20 B per ACTION token for GAE / RTG / whiten; the kernels work on the token-slot layout of PPOData (the
        # scatter of :635-645 and the index build of :230-243 are part of the launch), so `frac` above prices every slot they read / write —
        # `frac_per_action_token` is the same launch priced by §8(d)'s figure (the slots that are not action tokens count as overhead there)
        n_act = float(sta.sum())
        for name in ("gae", "rtg", "whiten_moments", "whiten_apply"):
            res[name]["frac_per_action_token"] = round(20.0 * n_act / (res[name]["avg_launch_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
            res[name]["action_slot_share_of_bytes"] = round(20.0 * n_act / res[name]["algorithmic_bytes_per_launch"], 3)
        res["action_tokens"] = int(n_act)
        res["action_tokens_per_s_gae"] = round(n_act / (res["gae"]["avg_launch_us"] * 1e-6), 0)
        res["buffer_sets"] = n_sets
        out[str(B)] = res
        del sets
        torch.cuda.empty_cache()
    return dict(chains=out, slots_per_chain=L, kernels="chain_scan_row2_kernel<GAE|RTG, 3> (four chains per wave, one per 16-lane DPP row, two slots per lane: register-resident reverse scans "
                "of affine maps by row_shl DPP steps, no LDS, streaming accesses), whiten_moments_kernel + whiten_finish_kernel / whiten_apply_kernel (16-byte accesses, fp64 partials)",
                note="HIP events around ONE hipGraph replay of 40 launches rotating through `buffer_sets` independent input / output sets (> 256 MB in total: not "
                     "served by the memory-side cache); avg_launch_us therefore includes the ~1 us dispatch gap between graph nodes; rocprofv3 summary of the same "
                     "command: profiles/r06_rl_reduce_kernel_stats.csv")


def run_warpers(vocab, guesses, seeds_all, B, dev, reps=3):
    """The headline workload with the task scripts' logits warpers (policy_top_k / policy_top_p, train_ppo_gpt2.py:98-99, 218-227) — informational leg,
    rank 0: top_k = 40 and top_k = 40 + top_p = 0.95 run on the FUSED top-k path (8 candidates per (row, tile) kept in the LM-head epilogue, no logits
    in HBM: csrc/sampler.hip lm_topc_epilogue / topc_reduce_sample_kernel); hipGraph replays, env-steps/s."""
    import torch
    from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine
    from lmrl_gym_amd.rollout import WordleRolloutEngine
    eng = GPT2Engine.random_init(GPT2Config.gpt2_small(), seed=0, device=dev)
    out = {"unit": "env-steps/s", "note": "same engine / envs / steer as `value`; top_k <= 64: fused candidate epilogue (tokens identical to the materialised "
                                          "path: tests/test_gpu_gpt2.py::test_fused_topk_candidate_path_equals_the_materialised_one)"}
    for name, kw in (("top_k=40", dict(top_k=40)), ("top_k=40,top_p=0.95", dict(top_k=40, top_p=0.95))):
        ro = WordleRolloutEngine(eng, vocab, B, max_new_tokens=6, bad_word_reward=-10.0)
        ro.capture_episode(temperature=1.0, sample_seed=7, steer_strength=30.0, scripted=True, **kw)
        ro.replay_episode(seeds_all[0, :B], guesses[0]); torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps = torch.zeros((), dtype=torch.int64, device=dev)
        for i in range(reps):
            ro.replay_episode(seeds_all[1 + i, :B], guesses[1 + i])
            steps += ro.traj["n_steps"].sum()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out[name] = {"value": round(int(steps.item()) / dt, 1), "ms_per_step": round(dt * 1e3 / reps, 2), "steps": reps}
        ro.close()
        del ro
    return out


def run_maze_rollout(dev, batches=(8, 1024), max_steps=20, max_new=12, reps=3):
    """configs[0]'s environment on the device loop (informational leg, rank 0): the fully observed Maze (`double_t_maze`,
    `describe_observation_give_position`, llm_rl_scripts/maze/bc/fully_observed_bc.py:230-283) with a random-init GPT-2-small policy and the byte-level
    stand-in tokenizer, `MazeRolloutEngine.run_episode` under per-turn hipGraph replay with the prompt-prefix cache: env-steps/s at configs[0]'s batch of 8
    and at the metric's 1024 envs, and one online-PPO data round (`ppo_rollouts`: episodes -> PPO data in HBM) at 1024."""
    import torch
    from lmrl_gym_amd import datasets as DS
    from lmrl_gym_amd.envs import maze as M
    from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine
    from lmrl_gym_amd.maze_rollout import MazeRolloutEngine
    tok = DS.ByteTokenizer()
    eng = GPT2Engine.random_init(GPT2Config.gpt2_small(), seed=0, device=dev)
    out = {"workload": f"double_t_maze, describe_observation_give_position, last_k=1, max_steps={max_steps}, max_new_tokens={max_new}, GPT-2-small random init, "
                       "byte tokenizer (prompt rows 129-141 tokens), temperature 1", "unit": "env-steps/s", "by_batch": {}}
    for B in batches:
        env = M.setup_maze_env("double_t_maze", "describe_observation_give_position", "standard_reward", last_k=1, max_steps=max_steps)
        r = MazeRolloutEngine(eng, tok, env, B, max_new_tokens=max_new, eos_token_id=tok.eos_token_id, max_input_length=160)
        r.run_episode(list(range(B)), sample_seed=1, use_graph=True, sync_every=0)           # capture + warm-up
        torch.cuda.synchronize(); t0 = time.perf_counter()
        steps = torch.zeros((), dtype=torch.int64, device=dev)
        for e in range(reps):
            r.run_episode(list(range(100 + e * B, 100 + (e + 1) * B)), sample_seed=1, episode=e + 1, use_graph=True, sync_every=0)
            steps += r.traj["n_turns"].sum()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out["by_batch"][str(B)] = {"value": round(int(steps.item()) / dt, 1), "ms_per_episode": round(dt * 1e3 / reps, 2), "episodes": reps,
                                   "ms_per_lockstep_turn": round(dt * 1e3 / reps / r.T, 3)}
        r.close()
        del r
    # the partially observed task's item window (last_k = 40, llm_rl_scripts/maze/bc/partially_observed_bc.py:241) on the persistent per-env KV cache
    # (round 6): 7-turn episodes at 1024 envs — every turn appends the action's tail + the new observation; the sampler is steered to random LEGAL
    # moves (a random-init policy never spells one, and an illegal string restarts the window, maze/env/env.py:179-180); tools/bench_maze_history.py
    # holds the longer regimes (re-prefill turns)
    B = batches[-1]
    env = M.setup_maze_env("double_t_maze", "describe_observation_only_walls", "standard_reward", last_k=40, max_steps=6)
    r = MazeRolloutEngine(eng, tok, env, B, max_new_tokens=max_new, eos_token_id=tok.eos_token_id, max_input_length=512)
    r.set_scripted_actions(np.random.RandomState(7).randint(0, 4, size=(r.T, B)), strength=30.0)
    r.run_episode(list(range(B)), sample_seed=3, use_graph=True, sync_every=0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    steps = torch.zeros((), dtype=torch.int64, device=dev)
    for e in range(reps):
        r.run_episode(list(range(100 + e * B, 100 + (e + 1) * B)), sample_seed=3, episode=e + 1, use_graph=True, sync_every=0)
        steps += r.traj["n_turns"].sum()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out["last_k40_window_growing"] = {"value": round(int(steps.item()) / dt, 1), "envs": B, "ms_per_lockstep_turn": round(dt * 1e3 / reps / r.T, 3), "turns": r.T,
                                      "schedule_flags": r.history_flags(),
                                      "note": "describe_observation_only_walls, last_k=40, max_input_length=512: append turns only (45-token observation + 11-token "
                                              "action forwarded per env and turn on the persistent cache + 12 decode steps)"}
    r.close()
    del r
    return out


def run_ppo_iteration(matmul, B, vocab, dev, rank, world, use_dist, backend, iters=2, train_steps=4, train_bsize=32, max_length=1024, host_path=True):
    """One ONLINE PPO iteration end to end (llm_rl_scripts/wordle/ppo/train_ppo_gpt2.py:301-353 + LLM_RL/algorithms/ppo/train.py):
    B-env lock-step rollouts on the device engine -> PPO data (policy + initial-policy log-probs, values, KL-shaped rewards, GAE, whitening) ->
    `train_steps` gradient steps of `train_bsize` x `max_length` (the script's blocking: max_input_length + max_output_length = 1024) -> the new
    weights pushed into the rollout engine.  Device-resident path (`WordleRolloutEngine.ppo_rollouts` -> `DevicePPODataset` ->
    `GPT2PPOTrain.step(**device batch)` -> `GPT2Engine.load_params`): no host text, no re-tokenisation, no [B, T, V] logits.  `host_path`: the same
    iteration through the reference-shaped host functions (`text_env_eval` -> text chains -> `get_ppo_data_from_text_trajectory_chain` ->
    `PPODataset` -> numpy batches -> a rebuilt engine), once, on rank 0's clock.  N > 1: rollouts and data are rank-local, the advantage moments and
    the gradients are all-reduced (RCCL)."""
    import torch
    import lmrl_gym_amd  # noqa: F401
    from lmrl_gym_amd.algorithms import ppo
    from lmrl_gym_amd.algorithms.common import BlockingStrategy, Padding, Truncation
    from lmrl_gym_amd.algorithms.ppo_inference import GPT2PPOInference, text_trajectory_chains_from_interactions
    from lmrl_gym_amd.datasets import WordleTokenizer
    from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine, init_hf_style_state_dict
    from lmrl_gym_amd.rollout import WordleRolloutEngine, WordleTokenTable
    from lmrl_gym_amd.train.gpt2_f32 import GPT2F32, LinearHeadF32
    cfg = GPT2Config.gpt2_small()
    pad = cfg.vocab                                         # the added <|pad|> token (train_ppo_gpt2.py:124-126): first id after the vocabulary, never sampled
    sd = init_hf_style_state_dict(cfg, seed=0)
    eng = GPT2Engine(cfg, sd, dev)
    table = WordleTokenTable.default_gpt2(pad=pad)
    ro = WordleRolloutEngine(eng, vocab, B, tokens=table, max_new_tokens=6, bad_word_reward=-10.0)
    pol, init = GPT2F32(sd, cfg.n_head, device=dev, matmul=matmul), GPT2F32(sd, cfg.n_head, device=dev, matmul=matmul)
    head = LinearHeadF32(dict(kernel=torch.zeros(cfg.d_model, 1), bias=torch.tensor([-4.1])), dev)          # train_ppo_gpt2.py:254-260
    lk = dict(cliprange_value=0.2, cliprange=0.2, value_loss_coef=1.0)
    inf = GPT2PPOInference(pol, head, pad, initial_policy=init, tokenizer=WordleTokenizer(table, pad_token_id=pad), loss_kwargs=lk)
    tr = ppo.GPT2PPOTrain(pol, head, pad, lk, lr=1e-5)
    n_it = iters + 1
    g = torch.from_numpy(scripted_guesses(vocab.all_vocab, n_it + 1, 6, B, seed=777 + rank).view(np.int32)).to(dev)
    kw = dict(gamma=1.0, lam=0.95, kl_weight=0.001, max_length=max_length)
    rng = np.random.default_rng(5 + rank)
    seeds = iter(range(10 ** 7 * (rank + 1), 10 ** 9))

    def barrier():
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def device_iteration(it, tm, trim=False):
        ds, kls, summary = ro.ppo_rollouts(inf, B, seed_generator=seeds, scripted_guesses_fn=lambda bid: g[it], steer_strength=30.0, temperature=1.0,
                                           sample_seed=4000 + rank, use_graph=True, timings=tm, **kw)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e2 = torch.cuda.Event(enable_timing=True)
        e0.record()
        perm = rng.permutation(len(ds))
        loss = None
        for k in range(train_steps):
            _, loss, _ = tr.step(**ds.batch(perm[k * train_bsize:(k + 1) * train_bsize], width=ds.trimmed_width() if trim else None))
        tm["batch_width"] = ds.trimmed_width() if trim else int(ds.input_ids.shape[1])
        e1.record()
        eng.load_params(pol.p)
        e2.record()
        torch.cuda.synchronize()
        tm["train_ms"] = tm.get("train_ms", 0.0) + e0.elapsed_time(e1)
        tm["push_weights_ms"] = tm.get("push_weights_ms", 0.0) + e1.elapsed_time(e2)
        return int(round(summary["length"]["mean"] * B)), float(loss), float(kls.mean())

    # the headline of this leg trains on batches cut to the round's longest episode (`DevicePPODataset.batch(width=trimmed_width())`, what
    # scripts/harness.py does by default): identical loss / logs / gradients to the script's fixed 1024-column blocking — asserted in
    # tests/test_gpu_ppo_device.py::test_train_step_on_trimmed_batches_equals_the_full_width_step — which exists for XLA's static shapes
    device_iteration(0, {}, trim=True)                       # warm: graph capture, workspaces, optimizer state
    tm = {}
    barrier()
    t0 = time.perf_counter()
    n_steps, loss, kl = 0, None, None
    for it in range(1, n_it):
        ns, loss, kl = device_iteration(it, tm, trim=True)
        n_steps += ns
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt, float(n_steps)], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if use_dist:
        mx = tt[:1].clone(); sm = tt[1:].clone()
        torch.distributed.all_reduce(mx, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(sm, op=torch.distributed.ReduceOp.SUM)
        dt, n_steps = float(mx.item()), int(sm.item())
    ph = {k: round(v / iters, 2) for k, v in tm.items() if k.endswith("_ms")}
    out = dict(value=round(n_steps / dt, 1), unit="env-steps/s", ms_per_iteration=round(dt * 1e3 / iters, 2), iterations=iters, matmul=matmul,
               envs_per_gpu=B, train_steps_per_iteration=train_steps, train_batch=f"{train_bsize} x {tm.get('batch_width')} (blocking width {max_length}, cut to ceil64(longest episode))",
               phases_ms=dict(rollout=ph.get("rollout_ms"), data_block=ph.get("block_ms"), data_forwards=ph.get("forward_ms"),
                              data_logprobs=ph.get("logprobs_ms"), data_shape_gae_whiten=ph.get("shape_gae_ms"), train=ph.get("train_ms"),
                              push_weights=ph.get("push_weights_ms")),
               data_rows=dict(sequences=tm.get("sequences"), forward_width=tm.get("forward_width"), lm_head_rows=tm.get("rows"),
                              action_tokens=tm.get("action_tokens"), pad_ids_inside=tm.get("pad_ids_inside")),
               last_loss=loss, mean_kl=kl,
               note=f"rank-local {B}-env rollouts (hipGraph replay) -> device PPO data -> {train_steps} steps of {train_bsize} x {tm.get('batch_width')} "
                    f"(an epoch over {B} rollouts would be {B // train_bsize} steps; the script's defaults are 128 rollouts = 4 steps) -> weights pushed into the engine; "
                    "phases: HIP events on the launch stream, value: wall clock over whole iterations")
    # the same iteration on the script's fixed blocking (every train batch max_length = 1024 columns, 93 % of them padding)
    device_iteration(0, {}, trim=False)
    tm2 = {}
    barrier()
    t0 = time.perf_counter()
    n2 = 0
    for it in range(1, n_it):
        n2 += device_iteration(it, tm2, trim=False)[0]
    barrier()
    dt2 = time.perf_counter() - t0
    tt2 = torch.tensor([dt2, float(n2)], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if use_dist:
        mx = tt2[:1].clone(); sm = tt2[1:].clone()
        torch.distributed.all_reduce(mx, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(sm, op=torch.distributed.ReduceOp.SUM)
        dt2, n2 = float(mx.item()), int(sm.item())
    out["full_width_batches"] = dict(value=round(n2 / dt2, 1), unit="env-steps/s", ms_per_iteration=round(dt2 * 1e3 / iters, 2),
                                     train_batch=f"{train_bsize} x {tm2.get('batch_width')}", train_ms=round(tm2.get("train_ms", 0.0) / iters, 2),
                                     note="train batches at the script's blocking width (max_input_length + max_output_length columns); everything else as above")
    if host_path and rank == 0 and world == 1:
        # the same iteration through the host-array functions with the reference's signatures (what scripts/harness.py ran before this round)
        tok = inf.tokenizer
        bs = BlockingStrategy(Padding.RIGHT, Truncation.RIGHT, max_length)
        th = {}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        raw, _ = ro.text_env_eval(B, seed_generator=seeds, scripted_guesses_fn=lambda bid: g[n_it], steer_strength=30.0, temperature=1.0, sample_seed=4000, use_graph=True)
        torch.cuda.synchronize(); th["rollout_and_host_lists"] = time.perf_counter() - t0
        t1 = time.perf_counter()
        chains = text_trajectory_chains_from_interactions(raw, tok, max_length, kw["gamma"])
        th["text_chains_and_length_rule"] = time.perf_counter() - t1
        t1 = time.perf_counter()
        datas, kls_h = inf.get_ppo_data_from_text_trajectory_chain(chains, bsize=32, max_length=max_length, gamma=kw["gamma"], lam=kw["lam"], kl_weight=kw["kl_weight"])
        dsh = ppo.PPODataset.from_ppo_data_list(datas, tok, bs)
        torch.cuda.synchronize(); th["ppo_data"] = time.perf_counter() - t1
        t1 = time.perf_counter()
        perm = rng.permutation(len(dsh))
        for k in range(train_steps):
            tr.step(**dsh[perm[k * train_bsize:(k + 1) * train_bsize]])
        torch.cuda.synchronize(); th["train"] = time.perf_counter() - t1
        t1 = time.perf_counter()
        eng2 = GPT2Engine(cfg, {k: v.detach().cpu() for k, v in pol.p.items()}, dev)
        torch.cuda.synchronize(); th["push_weights"] = time.perf_counter() - t1
        th_total = time.perf_counter() - t0
        del eng2
        out["host_path"] = dict(value=round(sum(len(ep) for ep in raw) / th_total, 1), unit="env-steps/s", ms_per_iteration=round(th_total * 1e3, 1), iterations=1,
                                phases_ms={k: round(v * 1e3, 1) for k, v in th.items()},
                                note="text_env_eval -> text chains -> get_ppo_data_from_text_trajectory_chain (bsize 32, [32, 1024, V] fp32 logits per forward) -> "
                                     "PPODataset -> numpy batches -> a new engine from host copies of the parameters")
    ro.close()
    del ro, eng, pol, init, tr, inf
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out


def main_train_step(args):
    """`--mode ilql-step` / `--mode ppo-step`: the train step of configs[2] as its own bench line (`value` = sequences / s over all ranks)."""
    import torch
    world, rank, dev, backend, use_dist = _dist_setup(torch)
    dist_info = _dist_report(torch, dev, backend, use_dist)
    r = run_train_step(args.mode, args.train_matmul, args.train_batch, args.steps, args.warmup, dev, rank, world, use_dist, backend, model=args.model)
    if rank == 0:
        bf = args.train_matmul == "bf16"
        B, T, ach = r["per_gpu_batch"], r["seq_len"], r["executed_tflops_per_gpu"]
        prec = "bf16 matmul operands, fp32 accumulation / parameters / optimizer" if bf else "fp32"
        roof = ({"bound": "mfma", "kernel": "gemm8_kernel / gemm_bf16_glds_kernel (Dense / Conv1D / head products: v_mfma_f32_16x16x32_bf16) + "
                                            "sgemm_f32_kernel (attention products, fp32)", "achieved": ach, "peak": 2500.0, "unit": "TFLOP/s",
                 "frac": round(ach / 2500.0, 4), "traffic": None,
                 "note": "model flops of one step / wall time of the whole step, priced against the dense bf16 MFMA peak although the attention "
                         "products and every elementwise / reduction kernel of the step stay fp32"} if bf else
                {"bound": "mfma", "kernel": "sgemm_f32_kernel (every matmul of the step: v_mfma_f32_32x32x2_f32)", "achieved": ach,
                 "peak": 157.3, "unit": "TFLOP/s", "frac": round(ach / 157.3, 4), "traffic": None,
                 "note": "model flops of one step / wall time of the whole step (lower bound for the GEMM kernel itself: 91 % of kernel time, "
                         "profiles/r01_train_kernel_stats_final.csv)"})
        line = {
            "metric": f"{args.mode} sequences/sec (GPT-2-{args.model} {'bf16-matmul' if bf else 'fp32'}, B={B}, T={T})", "value": r["sequences_per_s"],
            "unit": "sequences/s", "n_gpus": world, "steps": r["steps"], "warmup": r["warmup"], "ms_per_step": r["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if bf else "f32", "data": "synthetic",
            "config": {"workload": r["workload"].replace("fp32", prec), "per_gpu_batch": B, "seq_len": T,
                       "parallelism": f"dp{world}, one overlapped gradient all-reduce per step", "train_matmul": args.train_matmul, "last_loss": r["last_loss"]},
            "roofline": roof}
        for k in ("allreduce_bytes_per_step_per_rank", "allreduce_exposed_ms", "ms_per_step_without_allreduce", "grad_allreduce_dtype"):
            if k in r:
                line[k] = r[k]
        if dist_info is not None:
            line["dist"] = dist_info
            line["backend"], line["world_size"], line["rccl_version"] = dist_info["backend"], dist_info["world_size"], dist_info["rccl_version"]
        print(json.dumps(line), flush=True)
    if use_dist:
        torch.distributed.destroy_process_group()


def _spawn_ranks(argv, n):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: re-exec this command under torch.distributed.run, one rank per GPU
    (RCCL), exactly as the driver's explicit launch line does.  Rank 0's JSON line passes through on stdout."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    if "LMRL_BENCH_SKIP_DEVICE_CHECK" not in os.environ:        # (unit tests of the command line run without a GPU)
        import torch
        _resolve_backend(n, max(torch.cuda.device_count(), 1), os.environ.get("LMRL_BENCH_BACKEND"))   # refuse here, before N ranks start
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this host driver (RCCL / cross-process device memory)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--mode", default="rollout", choices=["rollout", "ilql-step", "ppo-step"],
                    help="rollout (default): the headline env-steps/s line; ilql-step / ppo-step: the train step of configs[2] (M3 / M4 sizes)")
    ap.add_argument("--model", default="small", choices=["small", "medium"], help="train-step modes: GPT-2 size (medium = configs[3]'s PPO policy)")
    ap.add_argument("--train-batch", type=int, default=32, help="sequences per GPU in the train-step modes")
    ap.add_argument("--train-matmul", default="f32", choices=["f32", "bf16"],
                    help="train-step modes: f32 = the reference's default arithmetic; bf16 = its optional bf16_activations mode (bf16 MFMA operands, "
                         "fp32 accumulation / parameters / gradients / optimizer)")
    ap.add_argument("--grad-allreduce", default="f32", choices=["f32", "bf16"],
                    help="N > 1, train step: wire format of the gradient all-reduce (f32 = exact, default; bf16 = lmrl_gym_amd.dist.set_grad_compression)")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024, help="envs per GPU")
    ap.add_argument("--vocab-file", default="wordle_official_400.txt")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", type=int, default=1, help="1: capture the episode into a hipGraph and replay it (default); 0: eager launches")
    ap.add_argument("--share-header", type=int, default=1, help="1 (default): the K/V rows of the header text every env starts from are computed "
                    "once per episode and broadcast to all envs (bit-identical to per-env prefill); 0: prefill the header per env")
    ap.add_argument("--streams", type=int, default=1, help="split the batch into this many sub-batches on separate HIP streams")
    ap.add_argument("--session-flags", type=int, default=0, help="lmrl_gpt2_forward variant bits of the headline engine's KV sessions (A/B runs: 64 / 128 = decode "
                                                                 "attention with 2 / 3 (env, head) items per wave)")
    ap.add_argument("--breakdown", action="store_true", help="after the timed region, print a per-kernel-class event breakdown to stderr")
    ap.add_argument("--no-fp32-mode", action="store_true", help="skip the `fp32_mode` leg of the default line (the same rollout on the fp32 engine)")
    ap.add_argument("--no-batch-sweep", action="store_true", help="skip the informational `larger_batches` leg of the default line (the same run at 4096 and 8192 envs per GPU, child processes)")
    ap.add_argument("--no-train-step", action="store_true", help="skip the `train_step` leg of the default line (ILQL M3 step, fp32 and bf16-matmul)")
    ap.add_argument("--no-rl-reduce", action="store_true", help="skip the `rl_reduce` leg of the default line (GAE / reward-to-go / whitening kernels at 4096 and 65536 chains)")
    ap.add_argument("--mode-rl-reduce-only", action="store_true", help="run ONLY the `rl_reduce` leg and print it as a JSON line (profiling: tools/prof_rl_reduce.sh)")
    ap.add_argument("--no-maze", action="store_true", help="skip the `maze_rollout` and `sampling_warpers` legs of the default line (configs[0]'s Maze env on the device loop at 8 and 1024 envs; the headline workload with top-k / top-p)")
    ap.add_argument("--no-ppo-iteration", action="store_true", help="skip the `ppo_iteration` leg of the default line (rollouts -> PPO data -> 4 train steps -> weights pushed back, device-resident and host path)")
    ap.add_argument("--ppo-iters", type=int, default=2, help="`ppo_iteration` leg: timed iterations per arithmetic mode (after one warm-up iteration)")
    ap.add_argument("--ppo-train-steps", type=int, default=4, help="`ppo_iteration` leg: gradient steps per iteration (32 sequences each)")
    ap.add_argument("--ppo-max-length", type=int, default=1024, help="`ppo_iteration` leg: blocking width of the PPO data / train batches (the script's 512 + 512)")
    ap.add_argument("--train-steps", type=int, default=5, help="timed steps per arithmetic mode in the `train_step` leg of the default line")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: spawn the N ranks ourselves (an external torch.distributed.run launch sets WORLD_SIZE and skips this)
        sys.exit(_spawn_ranks(sys.argv[1:], args.gpus))
    if args.grad_allreduce != "f32":
        import lmrl_gym_amd  # noqa: F401
        from lmrl_gym_amd import dist as _D
        _D.set_grad_compression(args.grad_allreduce)
    if args.mode != "rollout":
        return main_train_step(args)
    if args.mode_rl_reduce_only:
        import torch
        torch.cuda.set_device(0)
        print(json.dumps({"rl_reduce": run_rl_reduce(torch.device("cuda", 0))}), flush=True)
        return

    import torch
    import lmrl_gym_amd  # noqa: F401
    from lmrl_gym_amd import _lib
    from lmrl_gym_amd.envs import wordle as W
    from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine
    from lmrl_gym_amd.rollout import WordleRolloutEngine

    world, rank, dev, backend, use_dist = _dist_setup(torch)
    dist_info = _dist_report(torch, dev, backend, use_dist)

    L = _lib.lib()
    vocab = W.Vocabulary.builtin(args.vocab_file)
    cfg = GPT2Config.gpt2_small()
    eng = GPT2Engine.random_init(cfg, seed=0, device=dev)
    B, n_turns = args.batch, W.N_TRIES
    S = args.streams
    assert B % S == 0
    Bs = B // S
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else [torch.cuda.current_stream(dev)]
    ros = []
    for st in streams:
        with torch.cuda.stream(st):
            ros.append(WordleRolloutEngine(eng, vocab, Bs, max_new_tokens=6, bad_word_reward=-10.0, share_header=bool(args.share_header),
                                           session_flags=args.session_flags))
    ro = ros[0]
    n_eps = max(args.steps, 2) + args.warmup + (1 if args.breakdown else 0)      # (the roofline leg replays episodes warmup .. warmup + 1 eagerly)
    guesses = torch.from_numpy(scripted_guesses(vocab.all_vocab, n_eps, n_turns, B, seed=12345 + rank).view(np.int32)).to(dev)
    guesses_s = [guesses[:, :, k * Bs:(k + 1) * Bs].contiguous() for k in range(S)]
    # env seeds of every episode resident in HBM up front: no host->device copy (= host sync) between episodes
    seeds_all = torch.from_numpy((np.arange(n_eps * world * B, dtype=np.int64)).reshape(n_eps, world, B)[:, rank].copy()).to(dev)
    total_steps = torch.zeros((), dtype=torch.int64, device=dev)
    tag_ids = {L.lmrl_prof_tag_name(t).decode(): t for t in range(L.lmrl_prof_n_tags())}
    torch.cuda.synchronize()

    if args.graph:
        try:
            for r, st in zip(ros, streams):
                with torch.cuda.stream(st):
                    r.capture_episode(temperature=1.0, sample_seed=1000 + rank * 16 + ros.index(r), steer_strength=30.0, scripted=True)
            torch.cuda.synchronize()
        except Exception as e:   # never lose the measurement to a capture problem: same launches, issued eagerly
            print(f"[bench] rank {rank}: hipGraph capture failed ({type(e).__name__}: {e}); falling back to eager launches", file=sys.stderr)
            args.graph = 0
            torch.cuda.synchronize()
        if use_dist:             # every rank must time the same mode
            flag = torch.tensor([args.graph], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            args.graph = int(flag.item())

    def episode(i, count, eager=False):
        if args.graph and not eager:
            for k, (r, st) in enumerate(zip(ros, streams)):
                with torch.cuda.stream(st):
                    r.replay_episode(seeds_all[i, k * Bs:(k + 1) * Bs], guesses_s[k][i])
                    if count:
                        total_steps_parts.append(r.traj["n_steps"].sum())
            return
        gens = []
        for k, (r, st) in enumerate(zip(ros, streams)):
            with torch.cuda.stream(st):
                seeds = seeds_all[i, k * Bs:(k + 1) * Bs]
                gens.append(r.episode_phases(seeds, temperature=1.0, sample_seed=1000 + rank * 16 + k,
                                             scripted_guesses=guesses_s[k][i], steer_strength=30.0))
        live = list(range(S))
        while live:                      # round-robin: one phase per engine per pass
            for k in list(live):
                with torch.cuda.stream(streams[k]):
                    try:
                        next(gens[k])
                    except StopIteration:
                        live.remove(k)
        if count:
            for r, st in zip(ros, streams):
                with torch.cuda.stream(st):
                    total_steps_parts.append(r.traj["n_steps"].sum())

    total_steps_parts = []

    def barrier():
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        episode(i, False)
    L.lmrl_prof_reset()
    if not args.graph:
        L.lmrl_prof_enable(1 << tag_ids[ROOFLINE_TAG])   # only the roofline kernel is bracketed inside the timed region
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        episode(args.warmup + i, True)
    barrier()
    dt = time.perf_counter() - t0
    L.lmrl_prof_enable(0)
    # what a drop-in caller of `text_env_eval` pays on top of the device episode (LLM_RL/environment.py:211-267 returns HOST lists of
    # InteractionTransition): device -> host copy of the records + building the Python objects for this rank's B envs.  Outside `value`.
    th = time.perf_counter()
    n_trans = sum(len(ep) for r in ros for ep in r.interactions())
    host_materialise_ms = (time.perf_counter() - th) * 1e3
    # ... and the public call itself on the same steered workload: `WordleRolloutEngine.text_env_eval` over 4 episode batches (graph
    # replays; the host builds batch k's transition lists while the device runs batch k + 1), env steps returned / wall time
    tev = None
    if S == 1:
        n_b = min(4, n_eps)
        gen_seeds = iter(range(10 ** 6, 10 ** 9))
        kwf = dict(scripted_guesses_fn=lambda bid: guesses[bid % n_eps], steer_strength=30.0, temperature=1.0, sample_seed=9, use_graph=True)
        ro.text_env_eval(B, seed_generator=gen_seeds, **kwf)       # warm: pinned buffers, graph capture
        torch.cuda.synchronize(); th = time.perf_counter()
        inter_all, _summary = ro.text_env_eval(n_b * B, seed_generator=gen_seeds, **kwf)
        tev_s = time.perf_counter() - th
        tev = dict(value=round(sum(len(ep) for ep in inter_all) / tev_s, 1), unit="env-steps/s", episode_batches=n_b, ms_per_batch=round(tev_s * 1e3 / n_b, 2),
                   note="rank 0: WordleRolloutEngine.text_env_eval(n_rollouts = 4 x B) end to end — device episodes (hipGraph replays) + host lists of "
                        "InteractionTransition + summary dict, host materialisation of batch k overlapped with the device work of batch k + 1")
        del inter_all
        # three episode batches in flight (concurrent=3: twin engines — own KV cache, env state, graph — on their own HIP streams over the same
        # weights): each batch is still one lock-step batch of B envs; the dependent launch chains fill each other's gaps on the 256 CUs
        n_b2 = 12
        ro.text_env_eval(3 * B, seed_generator=gen_seeds, concurrent=3, **kwf)      # warm: twin engines, their graphs, pinned buffers
        torch.cuda.synchronize(); th = time.perf_counter()
        inter_all, _summary = ro.text_env_eval(n_b2 * B, seed_generator=gen_seeds, concurrent=3, **kwf)
        tev2_s = time.perf_counter() - th
        tev["three_batches_in_flight"] = dict(value=round(sum(len(ep) for ep in inter_all) / tev2_s, 1), unit="env-steps/s", episode_batches=n_b2,
                                            ms_per_batch=round(tev2_s * 1e3 / n_b2, 2),
                                            note="text_env_eval(n_rollouts = 12 x B, concurrent=3): batches k .. k + 2 run at once on three HIP streams "
                                                 "(3 x B envs in flight; the headline `value` stays one batch of B); tools/bench_text_env_eval_lanes.py: 1 / 2 / 3 lanes")
        del inter_all

    ms, work, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong()

    def read_tag(tag):
        _lib.check(L.lmrl_prof_read(tag_ids[tag], ctypes.byref(ms), ctypes.byref(work), ctypes.byref(n)))
        hbm = tag.startswith("attention")
        rate = (work.value / (ms.value * 1e-3)) / (1e9 if hbm else 1e12) if ms.value > 0 else 0.0
        peak = HBM_PEAK_GBS if hbm else MFMA_BF16_DENSE_PEAK_TFLOPS
        return dict(bound="hbm" if hbm else "mfma", kernel=tag, achieved=round(rate, 1), peak=peak,
                    unit="GB/s" if hbm else "TFLOP/s", frac=round(rate / peak, 4), traffic=None, launches=int(n.value),
                    avg_launch_us=round(ms.value * 1e3 / max(n.value, 1), 2), share_of_step_time=round(ms.value * 1e-3 / dt, 3))

    if args.graph:
        # hipGraph replays cannot carry per-kernel event brackets: the roofline kernel is timed in eager episodes of the
        # same workload run right after the timed region, in this process (same launches, same data).
        L.lmrl_prof_reset(); L.lmrl_prof_enable(1 << tag_ids[ROOFLINE_TAG])
        torch.cuda.synchronize(); tb = time.perf_counter()
        for i in range(2):
            episode(args.warmup + i, False, eager=True)
        torch.cuda.synchronize(); dt_keep, dt = dt, time.perf_counter() - tb
        L.lmrl_prof_enable(0)
        roofline = read_tag(ROOFLINE_TAG)
        roofline["timed_in"] = "2 eager episodes after the graph-replayed timed region"
        dt = dt_keep
    else:
        roofline = read_tag(ROOFLINE_TAG)
    # HBM traffic per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs of this same
    # command; gfx950 FETCH_SIZE counts 64 B per 128 B request for wide coalesced loads -> doubled, per the microarch
    # guide).  Not measurable from inside the process, so the committed summary is reported, with its source.
    try:
        pmc = json.load(open(os.path.join(ROOT, PROFILE_PMC)))
        key = next(k for k in pmc if ROOFLINE_KERNEL in k)
        live = csrc_digest()
        if pmc.get("__meta__", {}).get("csrc_digest") != live:
            # the counters were collected on other kernel sources than the ones running now: refuse to report them as this run's traffic
            roofline["traffic"] = None
            roofline["traffic_note"] = (f"{PROFILE_PMC} was collected on kernel sources {str(pmc.get('__meta__', {}).get('csrc_digest'))[:12]}, "
                                        f"live sources are {live[:12]}: stale, not reported (regenerate with tools/pmc_fetch_write.sh)")
        else:
            roofline["traffic"] = round((2.0 * pmc[key]["fetch_kb_avg"] + pmc[key]["write_kb_avg"]) * 1024)
            roofline["traffic_unit"] = f"bytes per launch (2*FETCH_SIZE + WRITE_SIZE, {PROFILE_PMC}, kernel sources {live[:12]})"
        roofline["algorithmic_bytes_per_launch"] = round(work.value / max(n.value, 1))
    except Exception:
        roofline["traffic"] = None
    # the committed rocprofv3 --kernel-trace --stats summary of this same command, for cross-checking the live event figure
    # (events bracket the launch on the stream and therefore include the dispatch gap, ~3 us, that rocprof's kernel
    # begin/end timestamps do not)
    try:
        import csv
        prof = os.path.join(ROOT, PROFILE_STATS)
        row = next(r for r in csv.DictReader(open(prof)) if ROOFLINE_KERNEL in r["Name"])
        roofline["rocprofv3_avg_launch_us"] = round(float(row["AverageNs"]) / 1e3, 2)
        roofline["rocprofv3_summary"] = PROFILE_STATS
    except Exception:
        pass
    # other kernel classes: one extra, untimed episode with their brackets on (brackets cost ~2 us per launch); the same episode yields
    # the whole-step algorithmic flop / byte totals of `roofline_step`
    L.lmrl_prof_reset()
    step_tags = SECONDARY_TAGS + [ROOFLINE_TAG]
    mask = 0
    for tg in step_tags:
        mask |= 1 << tag_ids[tg]
    L.lmrl_prof_enable(mask)
    torch.cuda.synchronize(); tb = time.perf_counter()
    episode(args.warmup, False, eager=True)
    torch.cuda.synchronize(); dt_keep, dt = dt, time.perf_counter() - tb
    L.lmrl_prof_enable(0)
    roofline_secondary = [read_tag(tg) for tg in SECONDARY_TAGS]
    dt = dt_keep
    # whole step: algorithmic flops = every GEMM / LM-head flop of the episode; algorithmic bytes = K/V rows the attention reads (device-
    # counted, shared prefix once) + the weights each forward has to stream once (bf16 layers per forward, the tied LM head per sampled token)
    flops = bytes_kv = 0.0
    for tg in step_tags:
        _lib.check(L.lmrl_prof_read(tag_ids[tg], ctypes.byref(ms), ctypes.byref(work), ctypes.byref(n)))
        if tg.startswith("attention"):
            bytes_kv += work.value
        else:
            flops += work.value
    n_fwd = n_turns * 6 + (0 if args.share_header else 1)                 # forwards over all envs: 5 decode + 1 chunk per turn (+ header chunk)
    layer_w = cfg.n_layer * (3 * cfg.d_model * cfg.d_model + cfg.d_model * cfg.d_model + 2 * cfg.d_model * cfg.d_ff) * 2
    bytes_w = n_fwd * layer_w + n_turns * 6 * cfg.vocab_padded * cfg.d_model * 2
    step_s = dt_keep / args.steps
    roofline_step = dict(flops=flops, bytes=bytes_kv + bytes_w, tflops=round(flops / step_s / 1e12, 1), tb_per_s=round((bytes_kv + bytes_w) / step_s / 1e12, 3),
                         frac_mfma_peak=round(flops / step_s / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS, 4),
                         frac_hbm_peak=round((bytes_kv + bytes_w) / step_s / 1e9 / HBM_PEAK_GBS, 4),
                         note="algorithmic work of one episode (counted in one eager episode of the same workload) / graph-timed ms_per_step")

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    steps_all = torch.stack(total_steps_parts).sum() if total_steps_parts else total_steps.clone()
    if use_dist:
        if backend != "nccl":
            t, steps_all = t.cpu(), steps_all.cpu()
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(steps_all, op=torch.distributed.ReduceOp.SUM)
    dt_max, n_env_steps = float(t.item()), int(steps_all.item())

    if args.breakdown and rank == 0:
        L.lmrl_prof_reset(); L.lmrl_prof_enable(0xFFFFFFFF)
        torch.cuda.synchronize(); tb = time.perf_counter()
        episode(n_eps - 1, False, eager=True)
        torch.cuda.synchronize(); te = time.perf_counter() - tb
        L.lmrl_prof_enable(0)
        print(f"[breakdown] one episode with every tag bracketed: {te * 1e3:.2f} ms", file=sys.stderr)
        for name, tid in tag_ids.items():
            _lib.check(L.lmrl_prof_read(tid, ctypes.byref(ms), ctypes.byref(work), ctypes.byref(n)))
            if n.value:
                rate = f"{work.value / (ms.value * 1e-3) / 1e12:8.1f} T(FLOP|B)/s" if work.value > 0 else ""
                print(f"[breakdown] {name:22s} {n.value:6d} launches {ms.value:9.3f} ms  avg {ms.value * 1e3 / n.value:8.2f} us {rate}", file=sys.stderr)

    if rank == 0:
        out = {
            "metric": "env-steps/sec (GPT-2-small Wordle, batch 1024)", "value": round(n_env_steps / dt_max, 1),
            "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_max * 1e3 / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "dtype_note": "the reference's OPTIONAL bf16 mode (bf16 weights / activations, fp32 accumulation, residual stream and logits; "
                          "BASELINE.md M2); the reference's default arithmetic is fp32",
            "config": {"workload": "configs[1]: Wordle env, GPT-2-small policy (random-init, steered sampling), "
                                   f"{B} lock-step envs per GPU, {n_turns} turns x <=6 generated tokens, vocab {args.vocab_file}",
                       "envs_per_gpu": B, "hip_streams": S, "hip_graph": bool(args.graph), "shared_header_prefill": bool(args.share_header), "max_new_tokens": 6, "parallelism": f"env-sharded x{world}, no data-path collective",
                       "env_steps_timed": n_env_steps},
            "roofline": roofline, "roofline_secondary": roofline_secondary, "roofline_step": roofline_step,
            "host_materialise_ms": round(host_materialise_ms, 2),
            "host_materialise_note": f"rank 0, one episode batch: device->host copy of the records + {n_trans} InteractionTransition objects built in "
                                     "Python (what text_env_eval returns); not inside `value`, which ends on the device",
            "value_incl_host_materialise": round(n_env_steps / (dt_max + args.steps * host_materialise_ms * 1e-3), 1),
            "text_env_eval": tev,
        }
        if dist_info is not None:     # N > 1 (or a forced 1-rank group): what the process group itself reports (the N = 1 line is unchanged)
            out["dist"] = dist_info
            out["backend"], out["world_size"], out["rccl_version"] = dist_info["backend"], dist_info["world_size"], dist_info["rccl_version"]
    for r in ros:
        r.close()
    del ros, ro, eng
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    if not args.no_fp32_mode:
        # the reference's DEFAULT rollout arithmetic (eval_bc_gpt2.py:34,69: float32): the same workload on GPT2EngineF32 — fp32 weights,
        # activations and K/V cache, exact-fp32 MFMA products, materialised fp32 logits, same sampler and env kernels; eager launches.
        # Its every sampled token is pinned to the float64 oracle in tests/test_gpu_f32_engine.py.  Reported beside `value`, never as it.
        from lmrl_gym_amd.gpt2_f32_engine import GPT2EngineF32
        for key_, mm_, desc_ in (("fp32_mode", "f32", "GPT2EngineF32: fp32 weights / activations / KV cache, v_mfma_f32_32x32x2_f32 GEMMs, materialised fp32 logits"),
                                 ("bf16x3_mode", "bf16x3", "GPT2EngineF32(matmul='bf16x3'): fp32 activations / KV cache / attention, every Dense + LM-head product as one "
                                                           "bf16 GEMM over three-term splits of the fp32 operands (~16 mantissa bits per product), fused LM-head sampler")):
            engf = GPT2EngineF32.random_init(cfg, seed=0, device=dev, matmul=mm_)
            rof = WordleRolloutEngine(engf, vocab, B, max_new_tokens=6, bad_word_reward=-10.0, share_header=bool(args.share_header))
            kwf = dict(temperature=1.0, sample_seed=1000 + rank * 16, steer_strength=30.0)
            graph_f = bool(args.graph)
            if graph_f:
                try:
                    rof.capture_episode(scripted=True, **kwf)
                except Exception as e:
                    print(f"[bench] rank {rank}: hipGraph capture of the {mm_} engine failed ({type(e).__name__}: {e}); eager launches", file=sys.stderr)
                    graph_f = False
            if use_dist:
                flagf = torch.tensor([int(graph_f)], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
                torch.distributed.all_reduce(flagf, op=torch.distributed.ReduceOp.MIN)
                graph_f = bool(int(flagf.item()))
            runf = (lambda i: rof.replay_episode(seeds_all[i], guesses[i])) if graph_f else (lambda i: rof.run_episode(seeds_all[i], scripted_guesses=guesses[i], **kwf))
            runf(0)
            barrier()
            tf0 = time.perf_counter()
            nf = []
            for i in range(2):
                runf(args.warmup + i)
                nf.append(rof.traj["n_steps"].sum())
            barrier()
            tf = torch.tensor([time.perf_counter() - tf0], dtype=torch.float64, device=dev)
            nfs = torch.stack(nf).sum()
            if use_dist:
                if backend != "nccl":
                    tf, nfs = tf.cpu(), nfs.cpu()
                torch.distributed.all_reduce(tf, op=torch.distributed.ReduceOp.MAX)
                torch.distributed.all_reduce(nfs, op=torch.distributed.ReduceOp.SUM)
            if rank == 0:
                out[key_] = {"value": round(int(nfs.item()) / float(tf.item()), 1), "unit": "env-steps/s", "ms_per_step": round(float(tf.item()) * 500.0, 2),
                             "steps": 2, "dtype": "f32" if mm_ == "f32" else "bf16x3 products, f32 everything else", "engine": desc_, "hip_graph": graph_f,
                             "parity": ("every sampled token == float64 oracle" if mm_ == "f32" else "every sampled token whose top-2 gap exceeds 3e-3 == float64 oracle")
                                       + " (tests/test_gpu_f32_engine.py)"}
            rof.close()
            del rof, engf
            gc.collect()
            torch.cuda.empty_cache()
    if not args.no_train_step:
        # the gradient step of the path (configs[2]): ILQL M3 in the reference's default fp32 arithmetic and in its optional bf16-matmul mode;
        # for N > 1 every rank takes part (data parallel, one gradient all-reduce per step over RCCL)
        ts = {}
        for mm in ("f32", "bf16"):
            ts["ilql_" + mm] = run_train_step("ilql-step", mm, args.train_batch, args.train_steps, 2, dev, rank, world, use_dist, backend)
        if rank == 0:
            out["train_step"] = ts
    if rank == 0 and not args.no_rl_reduce:
        out["rl_reduce"] = run_rl_reduce(dev)
    if rank == 0 and world == 1 and not args.no_maze and S == 1 and args.graph and n_eps >= 4:
        try:
            out["sampling_warpers"] = run_warpers(vocab, guesses, seeds_all, B, dev)
        except Exception as e:              # informational leg: never fails the line
            out["sampling_warpers"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        gc.collect(); torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_maze and S == 1 and args.graph:
        try:
            out["maze_rollout"] = run_maze_rollout(dev)
        except Exception as e:              # informational leg: never fails the line
            out["maze_rollout"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        gc.collect(); torch.cuda.empty_cache()
    if not args.no_ppo_iteration and S == 1 and args.graph:
        # the online loop end to end (rollouts -> PPO data -> train steps -> weights back into the engine), device-resident, in the headline's bf16
        # mode and in the reference's default fp32 arithmetic; the host-array path of the same iteration beside the bf16 one
        pi = {}
        for mm in ("bf16", "f32"):
            try:
                pi[mm] = run_ppo_iteration(mm, B, vocab, dev, rank, world, use_dist, backend, iters=args.ppo_iters, train_steps=args.ppo_train_steps,
                                           train_bsize=min(32, B), max_length=args.ppo_max_length, host_path=(mm == "bf16"))
            except Exception as e:          # an informational leg never loses the headline line (N > 1: a failing rank fails the collectives of all)
                if use_dist:
                    raise
                pi[mm] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if rank == 0:
            out["ppo_iteration"] = pi
    if rank == 0:
        if world == 1 and not args.no_batch_sweep and not args.no_cpu_baseline and B == 1024 and S == 1 and args.graph:       # the full default line only (the profiling tools pass --no-cpu-baseline)
            # the same engine, kernels and timed region at 4096 envs per GPU (a child process: this one's sessions and graphs stay as they are).  Not the
            # metric's configuration (1024 envs) — reported because the roofline kernel's fraction is a function of the launch size: north_star's
            # ">= 100 k env-steps/s at >= 60 % of the HBM roofline" holds together from 4096 envs per GPU up (profiles/r04_bench_batch_sweep.txt)
            import subprocess
            gc.collect(); torch.cuda.empty_cache()
            out["larger_batches"] = {"note": "python bench.py --batch B: the default line's engine, kernels and timed region at B envs per GPU; NOT the metric's "
                                             "configuration (1024 envs)"}
            for b_ in (4096, 8192):
                try:
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--batch", str(b_), "--steps", "4", "--warmup", "1", "--no-train-step", "--no-fp32-mode",
                                        "--no-cpu-baseline", "--no-batch-sweep", "--no-ppo-iteration", "--no-rl-reduce", "--no-maze"], capture_output=True, text=True, timeout=240)
                    d4 = json.loads(r.stdout.strip().splitlines()[-1])
                    out["larger_batches"][str(b_)] = {"value": d4["value"], "unit": d4["unit"], "ms_per_step": d4["ms_per_step"], "steps": d4["steps"],
                                                      "roofline_frac": d4["roofline"]["frac"], "roofline_kernel": d4["roofline"]["kernel"]}
                except Exception as e:       # informational leg: never fails the line
                    out["larger_batches"][str(b_)] = {"error": f"{type(e).__name__}: {e}"[:200]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(vocab.all_vocab)
            out["env_only"] = gpu_env_only(vocab, dev)
        print(json.dumps(out), flush=True)
    if use_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
