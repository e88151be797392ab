"""The train step's flash-attention kernels alone (forward, dQ, dK/dV; bf16 and fp32 operands) at the ILQL (T = 512) and PPO (T = 1024) bench
shapes, B = 32, H = 12 — run under `rocprofv3 --kernel-trace` to read per-kernel durations.  `--variants 0 7`: also the round-3 kernels
(lmrl_flash_set_variant bits) and the max |difference| of their outputs to the default kernels' on the same inputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lmrl_gym_amd import _lib
from lmrl_gym_amd.train import ops
dev = _lib.require_gpu()
L = _lib.lib()
variants = [int(v) for v in sys.argv[sys.argv.index("--variants") + 1:]] if "--variants" in sys.argv else [0]
for T in (512, 1024):
    for bf16 in (True, False):
        B, H = 32, 12
        d = H * 64
        g = torch.Generator().manual_seed(0)
        qkv = torch.randn(B * T, 3 * d, generator=g).to(dev)
        datt = torch.randn(B * T, d, generator=g).to(dev)
        ws, lse_n = ops.flash_attn_ws(B, H, T, bf16, dev)
        ref = None
        for v in variants:
            L.lmrl_flash_set_variant(v)
            att = torch.empty(B * T, d, device=dev); lse = torch.empty(lse_n, device=dev); dqkv = torch.empty(B * T, 3 * d, device=dev)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            for it in range(4):
                ev[0].record()
                ops.flash_attn_fwd(qkv, None, att, lse, ws, B, H, T, bf16)
                ev[1].record()
                ops.flash_attn_bwd(qkv, None, att, datt, lse, dqkv, ws, B, H, T, bf16, qkv_staged=True)
                ev[2].record()
            torch.cuda.synchronize()
            diff = ""
            if ref is None:
                ref = (att.clone(), lse.clone(), dqkv.clone())
            else:
                fin = torch.isfinite(ref[1])
                diff = "   vs variant %d: max|d att| %.2e  max|d lse| %.2e  max|d dqkv| %.2e (max|dqkv| %.2e)" % (
                    variants[0], (att - ref[0]).abs().max().item(), (lse[fin] - ref[1][fin]).abs().max().item(), (dqkv - ref[2]).abs().max().item(), ref[2].abs().max().item())
            print("T=%4d %s variant %d  forward (3 stagings + sweep) %7.1f us   backward (dO staging + dQ + dK/dV) %7.1f us%s"
                  % (T, "bf16" if bf16 else "fp32", v, ev[0].elapsed_time(ev[1]) * 1e3, ev[1].elapsed_time(ev[2]) * 1e3, diff))
L.lmrl_flash_set_variant(0)
