"""Timing-only ablations of the round-4 bf16 flash forward sweep (`flash_fwd2_bf16_kernel<ABL>`, tools build only: `python lmrl-gym_amd/build.py --tools`):
what a launch costs without its QK^T MFMAs (1), exponentials (2), PV MFMAs (4), all three (7), waits + barriers after the first block (8), waits +
barriers + tile traffic (24).  Results of the ablated launches are garbage.  B = 32, H = 12, T = 512 / 1024, operands pre-staged."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lmrl_gym_amd import _lib
_lib.SO_PATH = os.path.join(os.path.dirname(_lib.SO_PATH), "liblmrl_amd_tools.so")
assert os.path.exists(_lib.SO_PATH), "build it first: python lmrl-gym_amd/build.py --tools"
from lmrl_gym_amd.train import ops
dev = _lib.require_gpu()
L = _lib.lib()
for T in (512, 1024):
    B, H = 32, 12
    d = H * 64
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(B * T, 3 * d, generator=g).to(dev)
    ws, lse_n = ops.flash_attn_ws(B, H, T, True, dev)
    att = torch.empty(B * T, d, device=dev); lse = torch.empty(lse_n, device=dev); attb = torch.empty(B * T, d, dtype=torch.bfloat16, device=dev)
    ops.flash_attn_fwd(qkv, None, att, lse, ws, B, H, T, True)        # stages q / k / v
    for abl, name in ((0, "full"), (1 << 8, "no QK^T MFMAs"), (2 << 8, "no exp"), (4 << 8, "no PV MFMAs"), (7 << 8, "none of the three"),
                      (8 << 8, "no waits / barriers"), (24 << 8, "no waits / barriers / tile DMA"), (31 << 8, "all of the above"),
                      (0x30 << 16, "full, +48 KB LDS (1 WG per CU)"), (8, "full, 4-wave workgroups (64 queries)"), (1, "round-3 kernel")):
        L.lmrl_flash_set_variant(abl)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        best = 1e9
        for it in range(5):
            ev[0].record()
            for _ in range(10):
                _lib.check(L.lmrl_flash_attn_fwd_staged(None, None, _lib.ptr(att), _lib.ptr(lse), _lib.ptr(ws), _lib.ptr(attb), d, B, H, T, 1, _lib.stream_ptr()))
            ev[1].record(); torch.cuda.synchronize()
            best = min(best, ev[0].elapsed_time(ev[1]) * 100)
        print("T=%4d %-46s %7.1f us per launch (10 back to back)" % (T, name, best), flush=True)
L.lmrl_flash_set_variant(0)
