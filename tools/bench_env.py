"""Env-only micro-benchmark (workload M1 of BASELINE.md): scripted guesses, 6 steps / episode + reset."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lmrl_gym_amd  # noqa
from lmrl_gym_amd import _lib
from lmrl_gym_amd.envs import wordle as W

dev = torch.device("cuda")
for fname in ["wordle_official_400.txt", "wordle_official.txt"]:
    vocab = W.Vocabulary.builtin(fname)
    packed = np.array([W.pack_guess(w) for w in vocab.all_vocab], dtype=np.uint32)
    for n, variant in [(1024, 0), (8192, 0), (65536, 1), (65536, 2), (262144, 1), (262144, 2)]:      # variant 1: wave per env, 2: lane per env, 0: by size
        env = W.VectorWordleEnv(vocab, True, -10.0)
        env._alloc(n)
        _lib.check(_lib.lib().lmrl_wordle_set_variant(env._ctx, variant))
        rng = np.random.RandomState(12345)
        g = packed[rng.randint(0, len(packed), size=(6, n))]
        g[rng.rand(6, n) < 0.1] = W.BAD_GUESS
        gd = torch.from_numpy(g.view(np.int32)).to(dev)
        seeds = np.arange(n, dtype=np.uint64)
        env.reset_device(seeds)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        reps = 20
        t_reset = t_step = 0.0
        for _ in range(reps):
            ev[0].record(); env.reset_device(seeds); ev[1].record()
            for t in range(6):
                env.step_device(gd[t], None)
            ev[2].record(); torch.cuda.synchronize()
            t_reset += ev[0].elapsed_time(ev[1]); t_step += ev[1].elapsed_time(ev[2])
        print(f"V={len(packed)} N={n} kernel={['auto', 'wave/env', 'lane/env'][variant]}: reset {t_reset/reps*1e3:.1f} us, 6 steps {t_step/reps*1e3:.1f} us "
              f"-> {6*n/((t_reset+t_step)/reps/1e3)/1e6:.2f} M env-steps/s incl. reset, {6*n/(t_step/reps/1e3)/1e6:.2f} M/s steps only")
        env.close()
