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
