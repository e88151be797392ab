"""CPU tier: oracle/warpers.py (what the GPU tier's sampler tests check the HIP warpers against) pinned to the installed HF `transformers`
logits processors — the PyTorch twins of the Flax warpers the reference's `generate` runs (train_ppo_gpt2.py:98-99, 218-227)."""
import numpy as np
import pytest

from oracle import warpers as W


def test_warp_equals_the_hf_logits_processors():
    torch = pytest.importorskip("torch")
    transformers = pytest.importorskip("transformers")
    from transformers import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
    rng = np.random.default_rng(5)
    B, V = 48, 503
    base = rng.normal(size=(B, V)) * rng.uniform(0.5, 6.0, size=(B, 1))
    base[:, 7::11] = base[:, 6::11][:, : base[:, 7::11].shape[1]]                       # exact ties
    n_cases = 0
    for temp, top_k, top_p in [(1.0, 0, 0.9), (0.7, 0, 0.5), (1.3, 40, 0.0), (1.0, 5, 0.6), (0.8, 50, 0.95), (1.0, 1, 0.0), (2.0, 0, 0.999), (1.0, V, 0.3)]:
        z, keep = W.warp(base, temp, top_k, top_p)
        s = torch.from_numpy(base.copy())
        ids = torch.zeros(B, 1, dtype=torch.long)
        s = TemperatureLogitsWarper(temp)(ids, s)
        if 0 < top_k < V:
            s = TopKLogitsWarper(top_k=top_k)(ids, s)
        if 0.0 < top_p < 1.0:
            s = TopPLogitsWarper(top_p=top_p)(ids, s)
        hf_keep = torch.isfinite(s).numpy()
        # rows whose crossing falls within rounding of a token boundary may differ by that token (float cumulative sums in another order)
        zk = np.where(keep, z, -np.inf)
        p = np.exp(zk - zk.max(1, keepdims=True)); p /= p.sum(1, keepdims=True)
        srt = -np.sort(-p, axis=1)
        before = np.cumsum(srt, axis=1) - srt
        near = (np.abs(before - top_p) < 1e-9).any(1) if 0.0 < top_p < 1.0 else np.zeros(B, bool)
        assert near.mean() < 0.1
        # the kept SCORES are the same multiset; which of two exactly tied tokens at the crossing is kept is the sort's tie order (unspecified)
        mine, theirs = np.sort(np.where(keep, z, -np.inf), axis=1), np.sort(np.where(hf_keep, s.numpy(), -np.inf), axis=1)
        np.testing.assert_allclose(mine[~near], theirs[~near], rtol=1e-12, atol=0)
        differ = (keep != hf_keep).any(1) & ~near
        assert differ.sum() <= B // 4
        for r in np.where(differ)[0]:                                                   # only exact ties may swap
            a, b = np.where(keep[r] & ~hf_keep[r])[0], np.where(hf_keep[r] & ~keep[r])[0]
            assert len(a) == len(b) == 1 and z[r, a[0]] == z[r, b[0]]
        same = ~differ & ~near
        assert np.array_equal(keep[same], hf_keep[same]), (temp, top_k, top_p)
        lp = W.log_probs(z, keep)
        ref = torch.log_softmax(s, -1).numpy()
        both = keep & hf_keep & same[:, None]
        np.testing.assert_allclose(lp[both], ref[both], rtol=1e-9, atol=1e-9)
        n_cases += 1
    assert n_cases == 8


def test_warp_shapes_and_switches():
    z, keep = W.warp(np.array([1.0, 3.0, 2.0, 3.0]), 1.0, 2, 0.0)
    assert keep.tolist() == [False, True, False, True]                                  # ties with the k-th largest are kept
    z, keep = W.warp(np.array([0.0, 0.0, np.log(8.0)]), 1.0, 0, 0.5)
    assert keep.tolist() == [False, False, True]                                        # 0.8 of the mass on one token: it alone crosses 0.5
    z, keep = W.warp(np.array([0.0, 0.0, np.log(8.0)]), 1.0, 0, 0.85)
    assert int(keep.sum()) == 2                                                         # the token that crosses top_p is kept
    assert W.warp(np.zeros(5), 2.0)[1].all()
