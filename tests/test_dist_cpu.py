"""CPU tier, world_size 2 over gloo: the N > 1 plumbing (env sharding, bucketed gradient all-reduce, stat / moment
reductions, the bench's max-time / sum-steps reduction)."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist
import torch.multiprocessing as mp

import lmrl_gym_amd  # noqa: F401


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lmrl_gym_amd import dist as D
    try:
        assert D.is_distributed() and D.world() == (rank, world)
        # 1. shards tile the env range exactly
        lo, hi = D.shard_range(1027, rank, world)
        spans = [None] * world
        dist.all_gather_object(spans, (lo, hi))
        assert spans[0][0] == 0 and spans[-1][1] == 1027 and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        # 2. bucketed gradient all-reduce == sum over ranks, several buckets, ragged shapes
        g = torch.Generator().manual_seed(123)
        shapes = dict(a=(7, 5), b=(3,), c=(11, 2, 2), d=(1,))
        full = [{k: torch.randn(*s, generator=g) for k, s in shapes.items()} for _ in range(world * 2)]
        mine = [{k: v.clone() for k, v in full[rank * 2 + i].items()} for i in range(2)]
        n_coll = D.allreduce_grads(mine, bucket_bytes=100)
        assert n_coll >= 3
        for i in range(2):
            for k in shapes:
                exp = sum(full[r * 2 + i][k] for r in range(world))
                torch.testing.assert_close(mine[i][k], exp)
        avg = [{k: v.clone() for k, v in full[rank * 2].items()}]
        D.allreduce_grads(avg, average=True)
        torch.testing.assert_close(avg[0]["a"], sum(full[r * 2]["a"] for r in range(world)) / world)
        # 3. n-before-loss scheme: local grads of sum(x)/n_global summed over ranks == global-mean gradient
        x = torch.arange(10, dtype=torch.float64)[rank::world]
        n_glob = D.allreduce_sum_(torch.tensor([float(x.numel())], dtype=torch.float64))
        local_grad = {"w": (x / n_glob).sum().reshape(1)}
        D.allreduce_grads([local_grad])
        assert abs(float(local_grad["w"]) - 4.5) < 1e-12
        # 4. stats and whitening moments
        data = np.random.RandomState(0).randn(1000)
        sh = data[lo * 1000 // 1027: hi * 1000 // 1027] if False else data[rank::world]
        s, mn, mx = D.reduce_stat_partials([sh.sum(), (sh ** 2).sum(), len(sh)], [sh.min()], [sh.max()])
        assert abs(s[0] - data.sum()) < 1e-9 and s[2] == 1000 and mn[0] == data.min() and mx[0] == data.max()
        var = s[1] / s[2] - (s[0] / s[2]) ** 2
        assert abs(var - data.var()) < 1e-9
        # 4b. gradient arena: in-place bucketed all-reduce on slices of the flat buffer + the overlapped reducer fed in finalisation order
        from lmrl_gym_amd.train.gpt2_f32 import GradArena
        gp = torch.Generator().manual_seed(7)
        params = {"ln_f.weight": torch.zeros(5), "h.1.w": torch.zeros(4, 3), "h.1.b": torch.zeros(3), "h.0.w": torch.zeros(4, 3), "h.0.b": torch.zeros(3), "wte": torch.zeros(6, 2)}
        order = ["ln_f.weight", "h.1.w", "h.1.b", "h.0.w", "h.0.b", "wte"]
        per_rank = [{k: torch.randn(v.shape, generator=gp) for k, v in params.items()} for _ in range(world)]
        expect = {k: sum(per_rank[r][k] for r in range(world)) for k in params}
        ar = GradArena(params, order)
        assert ar.flat.numel() == sum(v.numel() for v in params.values()) and ar["h.0.w"].data_ptr() == ar.flat[ar.order["h.0.w"][0]:].data_ptr()
        for k in params:
            ar[k].copy_(per_rank[rank][k])
        assert D.allreduce_grads([ar], bucket_bytes=40) == -(-ar.flat.numel() * 4 // 40)          # 10 floats per collective, in place
        for k in params:
            torch.testing.assert_close(ar[k], expect[k])
        for k in params:
            ar[k].copy_(per_rank[rank][k])
        red = D.GradReducer(bucket_bytes=60)
        cb = red.ready(ar)
        cb(["ln_f.weight"]); cb(["h.1.w", "h.1.b"]); cb(["h.0.w", "h.0.b"]); cb(["wte"])
        extra = {"kernel": per_rank[rank]["wte"].clone()}
        assert red.finish([extra]) >= 3
        for k in params:
            torch.testing.assert_close(ar[k], expect[k])
        torch.testing.assert_close(extra["kernel"], expect["wte"])
        try:                                                     # out-of-order hand-over is a bug in the caller: refused
            red2 = D.GradReducer(); cb2 = red2.ready(ar); cb2(["h.0.w"])
            raise RuntimeError("expected an assertion")
        except AssertionError:
            pass
        # 4c. optional bf16 wire format of the gradient all-reduce (dist.set_grad_compression): fp32 arena in, fp32 arena out, every rank the SAME
        # bits, and within the written tolerance of the fp32 reduction: each addend and the sum are rounded to 8 mantissa bits ->
        # |err| <= 2^-8 * (sum of |addends| + |sum|) <= 3 * 2^-8 * sum |addends| (loose bound; RNE halves it)
        D.set_grad_compression("bf16")
        try:
            for k in params:
                ar[k].copy_(per_rank[rank][k])
            red = D.GradReducer(bucket_bytes=60)
            cb = red.ready(ar)
            cb(["ln_f.weight"]); cb(["h.1.w", "h.1.b"]); cb(["h.0.w", "h.0.b"]); cb(["wte"])
            red.finish()
            assert D.LAST_REDUCE_BYTES == ar.flat.numel() * 2                     # half the bytes on the wire
            assert ar.flat.dtype == torch.float32
            for k in params:
                bound = 3 * 2.0 ** -8 * sum(per_rank[r][k].abs() for r in range(world)) + 1e-30
                assert bool(((ar[k] - expect[k]).abs() <= bound).all()), k
                assert float((ar[k] - expect[k]).norm() / expect[k].norm()) < 2.0 ** -7, k
            mine_bits = ar.flat.clone()
            both = [torch.empty_like(mine_bits) for _ in range(world)]
            dist.all_gather(both, mine_bits)
            assert all(torch.equal(both[0], b) for b in both)                       # ranks cannot drift apart: identical reduced values
        finally:
            D.set_grad_compression(None)
        # 5. bench reduction: max time, summed steps
        t = torch.tensor([0.5 + rank], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
        n = torch.tensor([100 + rank]); dist.all_reduce(n, op=dist.ReduceOp.SUM)
        assert float(t) == 0.5 + world - 1 and int(n) == sum(100 + r for r in range(world))
        ret[rank] = "ok"
    except Exception as e:   # surface the failure in the parent
        ret[rank] = repr(e)
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)


def test_single_process_is_a_no_op():
    from lmrl_gym_amd import dist as D
    assert not D.is_distributed() and D.world() == (0, 1)
    g = [{"w": torch.ones(3)}]
    assert D.allreduce_grads(g) == 0 and torch.equal(g[0]["w"], torch.ones(3))
    assert D.shard_range(10, 0, 1) == (0, 10)


def test_bench_refuses_a_silent_gloo_fallback():
    """more ranks than GPUs -> non-zero exit unless gloo is asked for BY NAME; one GPU per rank -> RCCL."""
    import importlib
    import pytest
    bench = importlib.import_module("bench")
    assert bench._resolve_backend(1, 1, None) == "nccl" and bench._resolve_backend(8, 8, None) == "nccl"
    with pytest.raises(SystemExit) as e:
        bench._resolve_backend(8, 1, None)
    assert e.value.code not in (0, None) and "LMRL_BENCH_BACKEND=gloo" in str(e.value.code)
    with pytest.raises(SystemExit):
        bench._resolve_backend(2, 1, "")
    assert bench._resolve_backend(2, 1, "gloo") == "gloo" and bench._resolve_backend(2, 1, "nccl") == "nccl"
    with pytest.raises(SystemExit):
        bench._resolve_backend(2, 2, "mpi")


def test_bench_spawn_command(monkeypatch):
    """`python bench.py --gpus N` without a launcher environment re-execs itself under torch.distributed.run with N ranks on 127.0.0.1
    (the driver's own N > 1 line); with WORLD_SIZE set (external launcher) it must not spawn again."""
    import importlib
    import subprocess
    bench = importlib.import_module("bench")
    seen = {}
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.setenv("LMRL_BENCH_SKIP_DEVICE_CHECK", "1")     # no GPU in this tier: the device-count refusal is tested on its own above
    assert bench._spawn_ranks(["--gpus", "4", "--steps", "3"], 4) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # launcher environment present -> main() goes straight to the rank code (which needs a GPU): _spawn_ranks must not be called
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setattr(bench, "_spawn_ranks", lambda *a: (_ for _ in ()).throw(AssertionError("spawned twice")))
    monkeypatch.setattr(bench, "main_train_step", lambda args: "rank code")
    monkeypatch.setattr("sys.argv", ["bench.py", "--gpus", "4", "--mode", "ilql-step"])
    assert bench.main() == "rank code"
