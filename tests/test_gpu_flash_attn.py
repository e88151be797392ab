"""GPU tier: the train step's tiled attention (csrc/flash_attn_train.hip) against a float64 torch restatement of the HF-Flax GPT-2
attention the reference differentiates (softmax(Q K^T / sqrt(64) + causal & key-padding mask) V): forward output, and dq / dk / dv for a
random upstream gradient.  Cases: T not a multiple of 64, right padding, LEFT padding (queries with no valid key -> zero rows, as
lmrl_softmax_causal_fwd), several heads / batches.  Tolerances: fp32 operands 2e-5 of the tensor's largest entry; bf16 operands 2.5e-2."""
import math

import numpy as np
import pytest

import lmrl_gym_amd  # noqa: F401

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _ref(qkv, km, datt, B, T, H):
    d = H * 64
    x = qkv.double().cpu().view(B, T, 3, H, 64).requires_grad_(True)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)      # [B, H, T, 64]
    s = q @ k.transpose(-1, -2) / math.sqrt(64)
    causal = torch.tril(torch.ones(T, T, dtype=torch.bool))
    ok = causal[None, None] & (km.cpu().bool()[:, None, None, :])
    s = s.masked_fill(~ok, float("-inf"))
    p = torch.softmax(s, dim=-1)
    p = torch.where(ok.any(-1, keepdim=True), p, torch.zeros_like(p))          # rows without a valid key: all zeros (no NaN)
    p = torch.nan_to_num(p, nan=0.0)
    o = (p @ v).transpose(1, 2).reshape(B * T, d)
    (o * datt.double().cpu()).sum().backward()
    return o.detach(), x.grad.view(B * T, 3 * d)


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("B,T,H,pad", [(2, 96, 3, "right"), (1, 200, 2, "left"), (3, 64, 1, "none"), (2, 130, 12, "right"),
                                       (1, 1, 1, "none"), (2, 63, 2, "right"), (1, 129, 3, "left"), (2, 257, 2, "random"), (1, 1000, 1, "right"),
                                       (9, 128, 1, "random")])       # 9 heads: the XCD map's partial last group of 8
def test_flash_attention_fwd_bwd(B, T, H, pad, bf16):
    from lmrl_gym_amd import _lib
    from lmrl_gym_amd.train import ops
    dev = _lib.require_gpu()
    g = torch.Generator().manual_seed(B * 1000 + T)
    d = H * 64
    qkv = (torch.randn(B * T, 3 * d, generator=g) * 1.5).to(dev)
    datt = torch.randn(B * T, d, generator=g).to(dev)
    km = torch.ones(B, T, dtype=torch.uint8)
    if pad == "right":
        km[0, T - min(17, T // 2):] = 0
    elif pad == "left":
        km[0, :23] = 0
    elif pad == "random":                   # holes anywhere (key 0 of batch 0 masked too: its first queries have no valid key)
        km = (torch.rand(B, T, generator=g) > 0.3).to(torch.uint8)
        km[0, 0] = 0
    km = km.to(dev)
    ws, lse_n = ops.flash_attn_ws(B, H, T, bf16, dev)
    att = torch.full((B * T, d), 7.0, device=dev)
    lse = torch.empty(lse_n, device=dev)
    ops.flash_attn_fwd(qkv, km, att, lse, ws, B, H, T, bf16)
    dqkv = torch.full((B * T, 3 * d), 7.0, device=dev)
    ops.flash_attn_bwd(qkv, km, att, datt, lse, dqkv, ws, B, H, T, bf16, qkv_staged=True)     # the forward's staged q / k / v are still in ws
    dqkv2 = torch.full((B * T, 3 * d), 7.0, device=dev)
    ops.flash_attn_bwd(qkv, km, att, datt, lse, dqkv2, ws, B, H, T, bf16)                       # staged again: same bits
    assert torch.equal(dqkv, dqkv2)
    o_ref, g_ref = _ref(qkv, km, datt, B, T, H)
    tol = 2.5e-2 if bf16 else 2e-5
    err_o = float((att.double().cpu() - o_ref).abs().max()) / float(o_ref.abs().max())
    assert err_o <= tol, ("att", err_o)
    gd = dqkv.double().cpu()
    for name, sl in (("dq", slice(0, d)), ("dk", slice(d, 2 * d)), ("dv", slice(2 * d, 3 * d))):
        scale = float(g_ref[:, sl].abs().max())
        if scale < 1e-9:        # T = 1: dq and dk are exactly 0 (dS = P (dP - D) with P = 1, dP = D); the kernels' dP and D round differently
            scale = float(g_ref.abs().max())
        err = float((gd[:, sl] - g_ref[:, sl]).abs().max()) / scale
        assert err <= tol, (name, err)
    if pad == "left":                                  # queries before the first valid key: exact zero rows
        assert float(att.view(B, T, d)[0, :23].abs().max()) == 0.0
    if bf16:     # the staged form: d(qkv) written as the bf16 operand of the c_attn backward products == the rounded fp32 output
        mm = ops.MatmulBF16(dev)
        dst = ops.flash_attn_bwd_staged(mm, qkv, km, att, datt, lse, ws, B, H, T, qkv_staged=True)
        ldb = ops._pitch(3 * d)
        assert torch.equal(dst[: B * T * ldb].view(B * T, ldb)[:, : 3 * d], dqkv.to(torch.bfloat16))
        # D = rowsum(dO o O) from the bf16 copy of the attention output (ops.D_FROM_BF16_O: no fp32 O in the bf16-matmul train mode): within the bf16 tolerance
        lda = ops._pitch(d)
        attb = torch.zeros(B * T, lda, dtype=torch.bfloat16, device=dev)
        attb[:, :d] = att.to(torch.bfloat16)
        dst2 = ops.flash_attn_bwd_staged(mm, qkv, km, None, datt, lse, ws, B, H, T, qkv_staged=True, attb=attb, ld_attb=lda)
        g2 = dst2[: B * T * ldb].view(B * T, ldb)[:, : 3 * d].double().cpu()
        for name, sl in (("dq", slice(0, d)), ("dk", slice(d, 2 * d)), ("dv", slice(2 * d, 3 * d))):
            scale = float(g_ref[:, sl].abs().max()) or float(g_ref.abs().max())
            assert float((g2[:, sl] - g_ref[:, sl]).abs().max()) / scale <= tol, name


def test_flash_and_materialized_train_paths_agree():
    """The same GPT2F32 forward + backward with attention="flash" and "materialized" (fp32): hidden states and every gradient."""
    from lmrl_gym_amd import _lib
    from lmrl_gym_amd.gpt2 import GPT2Config, init_hf_style_state_dict
    from lmrl_gym_amd.train.gpt2_f32 import GPT2F32
    dev = _lib.require_gpu()
    cfg = GPT2Config(2, 4, 256, 512, 300, 160)
    sd = init_hf_style_state_dict(cfg, seed=3)
    for k in sd:
        sd[k] = sd[k] * 3
    B, T = 3, 150
    rng = np.random.RandomState(0)
    ids = torch.from_numpy(rng.randint(0, 299, size=(B, T)).astype(np.int32)).to(dev)
    am = torch.ones(B, T, dtype=torch.uint8); am[1, 120:] = 0
    am = am.to(dev)
    pos = torch.arange(T, dtype=torch.int32).repeat(B, 1).to(dev)
    dh = torch.from_numpy(rng.randn(B * T, cfg.d_model).astype(np.float32)).to(dev)
    out = {}
    for mode in ("flash", "materialized"):
        m = GPT2F32({k: v.clone() for k, v in sd.items()}, cfg.n_head, device=dev, attention=mode)
        hid, cache = m.forward(ids, am, pos)
        assert cache["flash"] == (mode == "flash")
        grads = m.zero_grads()
        m.backward(cache, dh.clone(), grads)
        out[mode] = (hid.clone(), {k: v.clone() for k, v in grads.items()})
    a, b = out["flash"][0], out["materialized"][0]       # both fp32, different summation orders: 1e-4 of the tensor's largest entry
    assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max())
    for k in out["flash"][1]:
        a, b = out["flash"][1][k], out["materialized"][1][k]
        assert float((a - b).abs().max()) <= 1e-4 * max(float(b.abs().max()), 1e-6), k
