"""GEMM micro-benchmark over the rollout shapes: A/B of kernel variants inside one process (interleaved rounds)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lmrl_gym_amd  # noqa
from lmrl_gym_amd import _lib
L = _lib.lib()
dev = torch.device("cuda")
shapes = [(1024, 2304, 768), (1024, 768, 768), (1024, 3072, 768), (1024, 768, 3072),
          (8192, 2304, 768), (8192, 768, 768), (8192, 3072, 768), (8192, 768, 3072), (1024, 50304, 768)]
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "0"])]
for (M, N, K) in shapes:
    NW = 12 if N < 10000 else 2
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    Ws = [W] + [W.clone() for _ in range(NW - 1)]   # cycle weight copies: cold L2 like consecutive layers
    b = torch.randn(N, device=dev); C = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    ref = None
    res = {}
    for rnd in range(3):
        for v in variants:
            L.lmrl_gemm_set_variant(v)
            for _ in range(3):
                L.lmrl_gemm_bf16(A.data_ptr(), W.data_ptr(), b.data_ptr(), C.data_ptr(), M, N, K, K, N, N, 0, None)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.default_stream())
            n = 24
            for it in range(n):
                L.lmrl_gemm_bf16(A.data_ptr(), Ws[it % NW].data_ptr(), b.data_ptr(), C.data_ptr(), M, N, K, K, N, N, 0, torch.cuda.current_stream().cuda_stream)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            res.setdefault(v, []).append(us)
            if ref is None:
                ref = C.float().clone()
            else:
                err = (C.float() - ref).abs().max().item()
                assert err < 0.1, (v, err)
    print(f"M={M:5d} N={N:5d} K={K:5d} " + "  ".join(f"v{v}: {min(t):7.1f} us {2*M*N*K/min(t)/1e6:6.0f} TF" for v, t in res.items()), flush=True)
L.lmrl_gemm_set_variant(0)
