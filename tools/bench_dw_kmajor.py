"""dW = x^T dy (bf16 operands, fp32 split-K accumulation) two ways on the train step's shapes: from transposed copies (lmrl_gemm_bf16_splitk; the
transposes are NOT timed) vs from the operands as staged (lmrl_gemm_bf16_splitk_kmajor, ds_read_b64_tr_b16 gathers).  Event-timed, 20 launches each,
interleaved rounds.  usage: python tools/bench_dw_kmajor.py"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from lmrl_gym_amd import _lib
from lmrl_gym_amd.train import ops

dev = _lib.require_gpu()
L = _lib.lib()
sp = _lib.stream_ptr


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for k, n, rows in ((768, 2304, 16384), (768, 768, 16384), (768, 3072, 16384), (3072, 768, 16384), (1024, 1024, 32768)):
    mm = ops.MatmulBF16(dev)
    x = torch.randn(rows, k, device=dev)
    dy = torch.randn(rows, n, device=dev) * 0.3
    xb, dyb = mm.cast("xs", x, rows, k, k).clone(), mm.cast("dys", dy, rows, n, n).clone()
    xt = mm.transpose_staged("xT", xb, ops._pitch(k), rows, k).clone()
    dyt = mm.transpose_staged("dyT", dyb, ops._pitch(n), rows, n).clone()
    dw0, dw1 = torch.zeros(k, n, device=dev), torch.zeros(k, n, device=dev)
    ws = torch.empty(L.lmrl_gemm_bf16_splitk_ws_bytes(k, n, rows) // 4, device=dev)
    ldr = ops._pitch(rows)
    f0 = lambda: _lib.check(L.lmrl_gemm_bf16_splitk(xt.data_ptr(), dyt.data_ptr(), dw0.data_ptr(), k, n, rows, ldr, ldr, n, n, 0, ws.data_ptr(), sp()), "splitk")
    f1 = lambda: _lib.check(L.lmrl_gemm_bf16_splitk_kmajor(xb.data_ptr(), dyb.data_ptr(), dw1.data_ptr(), k, n, rows, ops._pitch(k), ops._pitch(n), n, n, 0,
                                                           ws.data_ptr(), sp()), "kmajor")
    t = [[], []]
    for _ in range(3):
        t[0].append(timed(f0)); t[1].append(timed(f1))
    torch.cuda.synchronize()
    fl = 2.0 * k * n * rows
    print(f"dW [{k}][{n}] over {rows} rows: transposed operands {min(t[0]):7.1f} us ({fl / min(t[0]) / 1e6:5.0f} TF)   as staged {min(t[1]):7.1f} us "
          f"({fl / min(t[1]) / 1e6:5.0f} TF)   equal {bool(torch.equal(dw0, dw1))}", flush=True)
