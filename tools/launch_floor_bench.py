"""Per-kernel floor inside a hipGraph: N dependent tiny kernels back to back (what a launch boundary costs on this stack)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lmrl_gym_amd  # noqa
from lmrl_gym_amd import _lib
from lmrl_gym_amd.train import ops
dev = _lib.require_gpu()
for n_el, label in ((64, "1 workgroup"), (1024 * 768, "768 K elements (3 MB in, 3 MB out)"), (1024 * 3072, "3 M elements")):
    x = torch.zeros(n_el, device=dev); y = torch.zeros(n_el, device=dev)
    N = 400
    def body():
        for i in range(N // 2):
            ops.axpby(1.0, x, 0.0, None, y)      # y = x
            ops.axpby(1.0, y, 0.0, None, x)      # x = y  (dependent chain)
    body(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("%-36s %6.2f us per dependent kernel inside a hipGraph" % (label, dt / N * 1e6))
