"""Our bf16 GEMM kernels (lmrl_gemm_bf16, bf16 output) next to torch's bf16 matmul (= hipBLASLt on ROCm) on the shapes of the rollout and of
the train step.  A yardstick, not a dependency: nothing in the product path calls a BLAS library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lmrl_gym_amd
from lmrl_gym_amd import _lib
dev = _lib.require_gpu()
L = _lib.lib()
shapes = [("decode qkv", 1024, 2304, 768), ("decode proj", 1024, 768, 768), ("decode fc", 1024, 3072, 768), ("decode fc2", 1024, 768, 3072),
          ("prefill qkv", 7168, 2304, 768), ("prefill fc", 7168, 3072, 768), ("prefill fc2", 7168, 768, 3072), ("lm head", 1024, 50432, 768),
          ("train qkv", 16384, 2304, 768), ("train fc", 16384, 3072, 768), ("train fc2", 16384, 768, 3072), ("train head", 16384, 50432, 768),
          ("train dx head", 16384, 768, 50432), ("train dw fc", 768, 3072, 16384), ("train dwT head", 50258, 768, 16384), ("4096^3", 4096, 4096, 4096)]


def timed(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for name, M, N, K in shapes:
    a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.5).to(torch.bfloat16)
    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ours = timed(lambda: _lib.check(L.lmrl_gemm_bf16(a.data_ptr(), w.data_ptr(), None, c.data_ptr(), M, N, K, K, N, N, 0, _lib.stream_ptr())))
    ref = (a.float() @ w.float().t()) if M * N <= (1 << 26) else None
    if ref is not None:
        assert float((c.float() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())
    wt = w.t()
    lib = timed(lambda: torch.matmul(a, wt, out=c))
    fl = 2.0 * M * N * K / 1e6
    print("%-15s M=%6d N=%6d K=%6d   ours %8.1f us %7.1f TF   hipBLASLt (torch.matmul) %8.1f us %7.1f TF   ours/lib time %.2f"
          % (name, M, N, K, ours, fl / ours, lib, fl / lib, ours / lib), flush=True)
