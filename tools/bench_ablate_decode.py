"""What could fusing / hiding one launch of the decode layer buy AT MOST?  The bench episode (1024 envs, GPT-2-small, hipGraph replay) timed
with one launch class of the single-token decode layers left out (`LMRL_ABLATE_*` of csrc/ablate_tools.h: timing only, results are garbage; they exist only in the
-DLMRL_TOOLS build of the library, `python lmrl-gym_amd/build.py --tools`, which this script loads INSTEAD of the product library).  The difference
to the full episode is the in-situ cost of that launch including its share of ramp / tail / boundary — the upper bound for any scheme
that overlaps it with its neighbours.

    python tools/bench_ablate_decode.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import lmrl_gym_amd  # noqa: E402,F401
from lmrl_gym_amd import _lib  # noqa: E402
_lib.SO_PATH = os.path.join(os.path.dirname(_lib.SO_PATH), "liblmrl_amd_tools.so")   # the tools build; the product library rejects the ablation bits
assert os.path.exists(_lib.SO_PATH), "build it first: python lmrl-gym_amd/build.py --tools"
from lmrl_gym_amd.envs import wordle as W  # noqa: E402
from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine  # noqa: E402
from lmrl_gym_amd.rollout import WordleRolloutEngine  # noqa: E402
from bench import scripted_guesses  # noqa: E402

dev = _lib.require_gpu()
eng = GPT2Engine.random_init(GPT2Config.gpt2_small(), seed=0, device=dev)
vocab = W.Vocabulary.builtin("wordle_official_400.txt")
B, N = 1024, 6
g = torch.from_numpy(scripted_guesses(vocab.all_vocab, N + 1, 6, B, seed=1).view(np.int32)).to(dev)
seeds = torch.arange((N + 1) * B, dtype=torch.int64, device=dev).view(N + 1, B)
names = [("full episode", 0), ("proj via the aux stream, SERIAL (cost of one fork / join per layer)", 256), ("without decode qkv GEMM", 1), ("without decode attention", 2), ("without decode proj GEMM", 4),
         ("without decode fc GEMM", 8), ("without decode fc2 GEMM", 16), ("without proj + fc2", 20), ("without all five", 31),
         ("proj CONCURRENT with attention (aux stream, stale data)", 32), ("fc2 over half of K only", 128), ("decode attention over HALF of the cached positions", 512),
         ("fc2 as two CONCURRENT half-K launches (split-K 2, no seam)", 64),
         ("fc2 as ONE split-K 3 launch (64x64) + reduce launch (seam)", 1024), ("fc2 as ONE split-K 2 launch (64x64) + reduce launch (seam)", 2048),
         ("fc2 as ONE split-K 6 launch (128x128) + reduce launch (seam)", 4096)]
if "--seam" in sys.argv:
    names = [names[0]] + names[-3:]
base = None
n_dec = 30 * 12                                                   # decode layers per episode: 30 forwards x 12 layers
for name, bits in names * 2:                                     # two passes: the second one is reported (clocks settled)
    ro = WordleRolloutEngine(eng, vocab, B, max_new_tokens=6)
    ro.ses.flags |= bits << 16
    ro.capture_episode(temperature=1.0, sample_seed=5, steer_strength=30.0, scripted=True)
    ro.replay_episode(seeds[0], g[0]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(1, N + 1):
        ro.replay_episode(seeds[i], g[i])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / N * 1e3
    if bits == 0:
        base = ms
    print(f"{name:68s} {ms:7.2f} ms per episode   delta {base - ms:6.2f} ms = {(base - ms) * 1e3 / n_dec:6.2f} us per decode layer", flush=True)
    ro.close()
    del ro
