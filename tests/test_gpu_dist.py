"""GPU tier: the data-parallel plumbing of lmrl_gym_amd.dist on RCCL (backend "nccl") with device tensors.  One GPU is
available to this tier, so the group has a single rank: every collective is the identity, which makes the distributed
train step comparable bit-for-bit with the single-process one while still driving the bucket packing, the fp64 statistic
reductions and the RCCL calls themselves.  The world_size-2 semantics are covered on gloo in tests/test_dist_cpu.py."""
import numpy as np
import pytest

import lmrl_gym_amd  # noqa: F401

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture()
def rccl_group(monkeypatch):
    import torch.distributed as dist
    from lmrl_gym_amd import _lib, dist as D
    dev = _lib.require_gpu()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1, device_id=dev)
    monkeypatch.setattr(D, "is_distributed", lambda: True)
    yield dev
    dist.destroy_process_group()


def test_rccl_collectives_and_train_step_match_single_process(rccl_group):
    import torch.distributed as dist
    from lmrl_gym_amd import dist as D
    from lmrl_gym_amd.algorithms import ppo
    from lmrl_gym_amd.gpt2 import GPT2Config, init_hf_style_state_dict
    from lmrl_gym_amd.train.gpt2_f32 import GPT2F32, LinearHeadF32
    dev = rccl_group
    assert dist.get_backend() == "nccl"
    # bucketed gradient all-reduce: several buckets, mixed sizes, values preserved
    g = torch.Generator().manual_seed(0)
    grads = [{f"w{i}": torch.randn(1000 + 37 * i, generator=g).to(dev) for i in range(9)}, {"b": torch.randn(5, generator=g).to(dev)}]
    ref = [{k: v.clone() for k, v in d.items()} for d in grads]
    nb = D.allreduce_grads(grads, bucket_bytes=16 << 10)
    assert nb >= 2
    for d, r in zip(grads, ref):
        for k in d:
            assert torch.equal(d[k], r[k])
    x = torch.randn(4096, generator=g).to(dev); mask = (torch.rand(4096, generator=g) < 0.4).to(torch.uint8).to(dev)
    y = D.whiten_distributed(x, mask, shift_mean=True).cpu().numpy()
    xm = x.cpu().numpy()[mask.cpu().numpy() > 0]
    np.testing.assert_allclose(y[mask.cpu().numpy() > 0], (xm - xm.mean()) / np.sqrt(xm.var() + 1e-8), rtol=2e-5, atol=2e-5)
    # a PPO train step under the distributed code path == the single-process step (1 rank: same numbers)
    cfg = GPT2Config(2, 2, 64, 128, 211, 32)
    sd = init_hf_style_state_dict(cfg, seed=4)
    rng = np.random.RandomState(1)
    B, T, pad = 4, 14, cfg.vocab - 1
    ids = rng.randint(1, pad, size=(B, T)).astype(np.int32); ids[1, 9:] = pad
    sta = np.zeros((B, T - 1), bool); sta[:, 3:8] = True
    f = lambda s: (rng.randn(B, T - 1) * s).astype(np.float32)
    olp, ov, oa, orr = f(0.1) - 5.0, f(1), f(1), f(1)
    kw = dict(cliprange_value=0.2, cliprange=0.2, value_loss_coef=1.0)

    def run(distributed):
        import lmrl_gym_amd.dist as DD
        if not distributed:
            saved = DD.is_distributed
            DD.is_distributed = lambda: False
        try:
            pol = GPT2F32({k: v.clone() for k, v in sd.items()}, cfg.n_head, device=dev)
            head = LinearHeadF32(dict(kernel=torch.zeros(cfg.d_model, 1), bias=torch.tensor([-1.0])), dev)
            tr = ppo.GPT2PPOTrain(pol, head, pad, kw, lr=1e-3)
            _, loss, logs = tr.step(ids, sta, olp, ov, oa, orr)
            return loss, pol.p["h.1.mlp.c_fc.weight"].clone(), tr.last_grads[0]["wte.weight"].clone()
        finally:
            if not distributed:
                DD.is_distributed = saved

    l1, w1, g1 = run(True)
    l0, w0, g0 = run(False)
    assert abs(l1 - l0) <= 1e-6 * max(1.0, abs(l0)) and torch.equal(g1, g0) and torch.equal(w1, w0)


# ------------------------------------------------------------------ 2 ranks x half batch == 1 rank x full batch (VERDICT r01 #9a)
def _toy(seed, vocab=211):
    from lmrl_gym_amd.gpt2 import GPT2Config, init_hf_style_state_dict
    cfg = GPT2Config(2, 2, 64, 128, vocab, 32)
    sd = init_hf_style_state_dict(cfg, seed=seed)
    g = torch.Generator().manual_seed(seed + 100)
    for k in sd:
        sd[k] = sd[k] * 4 + (0.1 * torch.randn(sd[k].shape, generator=g) if sd[k].dim() == 1 else 0)
    return cfg, sd


def _batches(seed, B, T, vocab):
    rng = np.random.RandomState(seed)
    pad = vocab - 1
    ids = rng.randint(1, vocab - 1, size=(B, T)).astype(np.int32)
    lens = rng.randint(T // 2, T + 1, size=B); lens[0] = T; lens[B // 2] = T       # every half holds a full-length row: same T after blocking
    for b in range(B):
        ids[b, lens[b]:] = pad
    sta = np.zeros((B, T - 1), dtype=bool)
    for b in range(B):
        for t in range(3, lens[b] - 1):
            sta[b, t] = (t // 3) % 2 == (b % 2)
    f = lambda s: (rng.randn(B, T - 1) * s).astype(np.float32)
    return dict(ids=ids, sta=sta, olp=f(0.2) - 5.0, ov=f(1), oa=f(1), orr=f(1), rewards=f(1) * sta, dones=(rng.rand(B) < 0.5).astype(np.float32),
                returns=f(1) * sta, pad=pad)


def _run_algo(algo, dev, rows):
    """One optimizer step of `algo` on the batch rows `rows`; returns (loss, flat logs, {name: grad}, {name: param after AdamW})."""
    from lmrl_gym_amd.algorithms import ilql, mc_returns as mc, ppo
    from lmrl_gym_amd.train.gpt2_f32 import GPT2F32, LinearHeadF32, MLPHeadF32
    cfg, sd = _toy(7)
    bt = _batches(11, 8, 15, cfg.vocab)
    sl = lambda k: bt[k][rows]
    g = torch.Generator().manual_seed(3)
    d, V = cfg.d_model, cfg.vocab
    mk = lambda out: {"dense1.kernel": torch.randn(d, d, generator=g) * 0.2, "dense1.bias": torch.randn(d, generator=g) * 0.1,
                      "dense2.kernel": torch.randn(d, out, generator=g) * 0.2, "dense2.bias": torch.full((out,), -0.4)}
    base = GPT2F32({k: v.clone() for k, v in sd.items()}, cfg.n_head, device=dev)
    if algo == "ppo":
        head = LinearHeadF32(dict(kernel=torch.randn(d, 1, generator=g) * 0.1, bias=torch.tensor([-1.0])), dev)
        tr = ppo.GPT2PPOTrain(base, head, bt["pad"], dict(cliprange_value=0.2, cliprange=0.2, value_loss_coef=1.0), lr=1e-3, weight_decay=0.01)
        _, loss, logs = tr.step(sl("ids"), sl("sta"), sl("olp"), sl("ov"), sl("oa"), sl("orr"))
        grads, heads = tr.last_grads[0], {"head." + k: v for k, v in tr.last_grads[1].items()}
    elif algo == "ilql":
        tr = ilql.GPT2ILQLTrain(base, MLPHeadF32(mk(V), dev), MLPHeadF32(mk(V), dev), MLPHeadF32(mk(1), dev), bt["pad"],
                                dict(gamma=0.99, tau=0.7, cql_weight=0.01), lr=1e-3)
        _, loss, logs = tr.step(sl("ids"), sl("sta"), sl("rewards"), sl("dones"))
        grads = tr.last_grads[0]
        heads = {f"h{i}." + k: v for i, hg in enumerate(tr.last_grads[1:]) for k, v in hg.items()}
    else:
        tr = mc.GPT2MCTrain(base, MLPHeadF32(mk(V), dev), bt["pad"], dict(cql_weight=0.05), lr=1e-3)
        _, loss, logs = tr.step(sl("ids"), sl("sta"), sl("returns"))
        grads, heads = tr.last_grads[0], {"q." + k: v for k, v in tr.last_grads[1].items()}
    flat = {}

    def fl(dct, pre=""):
        for k, v in dct.items():
            if isinstance(v, dict):
                fl(v, pre + k + ".")
            else:
                flat[pre + k] = float(v)
    fl(logs)
    gd = {k: v.detach().cpu().clone() for k, v in list(grads.items()) + list(heads.items())}
    pd = {k: v.detach().cpu().clone() for k, v in base.p.items()}
    return float(loss), flat, gd, pd


def _dp_worker(rank, world, port, path, ret):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)      # both ranks share the one GPU; device tensors are staged by dist.py
    try:
        from lmrl_gym_amd import _lib, dist as D
        dev = _lib.require_gpu()
        assert D.is_distributed()
        ref = torch.load(path)
        for algo in ("ppo", "ilql", "mc"):
            lo, hi = D.shard_range(8, rank, world)
            loss, logs, gd, pd = _run_algo(algo, dev, slice(lo, hi))
            e_loss, e_logs, e_gd, e_pd = ref[algo]
            assert abs(loss - e_loss) <= 2e-6 * max(1.0, abs(e_loss)), (algo, loss, e_loss)
            assert set(logs) == set(e_logs)
            for k in e_logs:
                if not (np.isnan(e_logs[k]) and np.isnan(logs[k])):
                    assert abs(logs[k] - e_logs[k]) <= 3e-6 * max(1.0, abs(e_logs[k])), (algo, k, logs[k], e_logs[k])
            for k in e_gd:                                    # summed local gradients == the full-batch gradient
                torch.testing.assert_close(gd[k], e_gd[k], rtol=2e-5, atol=2e-5 * float(e_gd[k].abs().max()) + 1e-9, msg=f"{algo} grad {k}")
            for k in ("h.0.attn.c_attn.weight", "ln_f.weight", "wte.weight"):     # AdamW step 1 = lr * sign-like update: compare where the gradient is not ~0
                big = e_gd[k].abs() > 1e-4 * e_gd[k].abs().max()
                assert float((pd[k] - e_pd[k])[big].abs().max()) < 1e-5, (algo, k)
        # optional bf16 wire format of the gradient all-reduce (dist.set_grad_compression("bf16")): loss / logs do not pass
        # through it (exact as above); every reduced gradient tensor within 2^-7 relative L2 of the full-batch fp32 gradient and each element
        # within 3 * 2^-8 of the tensor's largest entry (each addend and the sum rounded to 8 mantissa bits); gradients stay fp32 tensors
        D.set_grad_compression("bf16")
        try:
            lo, hi = D.shard_range(8, rank, world)
            loss, logs, gd, pd = _run_algo("ilql", dev, slice(lo, hi))
            e_loss, e_logs, e_gd, e_pd = ref["ilql"]
            assert abs(loss - e_loss) <= 2e-6 * max(1.0, abs(e_loss))
            assert D.LAST_REDUCE_BYTES == 2 * sum(v.numel() for v in e_gd.values())
            for k in e_gd:
                assert gd[k].dtype == torch.float32
                nrm = float(e_gd[k].norm())
                assert float((gd[k] - e_gd[k]).norm()) <= 2.0 ** -7 * nrm + 1e-12, (k, float((gd[k] - e_gd[k]).norm()), nrm)
                assert float((gd[k] - e_gd[k]).abs().max()) <= 3 * 2.0 ** -8 * float(e_gd[k].abs().max()) + 1e-12, k
        finally:
            D.set_grad_compression(None)
        ret[rank] = "ok"
    except Exception as e:
        import traceback
        ret[rank] = traceback.format_exc()
    finally:
        dist.destroy_process_group()


def test_two_ranks_half_batch_equal_one_rank_full_batch(tmp_path):
    """Real PPO / ILQL / MC train steps under data parallelism: 2 processes (gloo; they share the single GPU of this tier) x half batch
    must reproduce the single-process full-batch step — loss, every log entry, the all-reduced gradients and the post-AdamW parameters.
    Exercises the n-before-loss convention, the stat reductions, the in-place arena all-reduce and the backward-overlapped reducer."""
    import socket
    import torch.multiprocessing as mp
    from lmrl_gym_amd import _lib
    dev = _lib.require_gpu()
    ref = {algo: _run_algo(algo, dev, slice(0, 8)) for algo in ("ppo", "ilql", "mc")}
    path = str(tmp_path / "ref.pt")
    torch.save(ref, path)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_worker, args=(2, port, path, ret), nprocs=2, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)


def test_bench_gpus_2_spawns_two_ranks_and_reports_the_train_step():
    """the plain driver command `python bench.py --gpus 2 ...` must itself launch 2 ranks (torch.distributed.run, one
    process per GPU; on this 1-GPU tier the ranks share the GPU, which the bench refuses unless LMRL_BENCH_BACKEND=gloo is explicit) and rank 0 must print ONE JSON line with
    n_gpus = 2 whose `train_step` object carries the ILQL step with its gradient all-reduce (bytes, exposed time)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LMRL_BENCH_BACKEND")}
    if torch.cuda.device_count() < 2:
        # two ranks on ONE GPU: without the explicit override the bench must refuse (non-zero) instead of quietly measuring gloo
        r0 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "64", "--no-fp32-mode",
                             "--no-train-step", "--no-cpu-baseline", "--no-ppo-iteration"], capture_output=True, text=True, timeout=600, env=env)
        assert r0.returncode != 0 and "LMRL_BENCH_BACKEND=gloo" in (r0.stderr + r0.stdout), (r0.returncode, r0.stderr[-1500:])
        assert not [ln for ln in r0.stdout.splitlines() if ln.startswith("{")]
        env["LMRL_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "64",
                        "--train-batch", "2", "--train-steps", "1", "--no-cpu-baseline", "--ppo-iters", "1", "--ppo-train-steps", "1", "--ppo-max-length", "192"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["envs_per_gpu"] == 64
    # the line verifies itself: what the process group reports, per-rank devices
    assert out["world_size"] == 2 and out["backend"] == ("gloo" if torch.cuda.device_count() < 2 else "nccl")
    assert [rd["rank"] for rd in out["dist"]["rank_devices"]] == [0, 1] and out["dist"]["ranks_share_devices"] == (torch.cuda.device_count() < 2)
    assert (out["rccl_version"] is not None) == (out["backend"] == "nccl")
    assert out["config"]["env_steps_timed"] >= 2 * 64            # both ranks' env steps are summed
    assert out["fp32_mode"]["value"] > 0 and out["bf16x3_mode"]["value"] > 0 and out["host_materialise_ms"] > 0
    # the online PPO iteration as a 2-rank data-parallel job: rank-local rollouts and PPO data, advantage moments and gradients all-reduced
    for mm in ("bf16", "f32"):
        pi = out["ppo_iteration"][mm]
        assert pi["value"] > 0 and np.isfinite(pi["last_loss"]) and pi["envs_per_gpu"] == 64 and pi["phases_ms"]["train"] > 0, pi
    ts = out["train_step"]
    for k in ("ilql_f32", "ilql_bf16"):
        assert ts[k]["ms_per_step"] > 0 and np.isfinite(ts[k]["last_loss"])
        # base transformer + two Q heads + V head, fp32: the ONE data-path collective of the step
        assert ts[k]["allreduce_bytes_per_step_per_rank"] > 4 * 124e6
        assert "allreduce_exposed_ms" in ts[k]
