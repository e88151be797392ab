"""GPU tier: GPT-2 rollout engine (bf16 MFMA GEMM, KV-cache attention, fused LM-head sampler) through the C ABI
against the torch-CPU oracle (oracle/gpt2.py, itself pinned to HF PyTorch GPT-2)."""
import ctypes

import numpy as np
import pytest

import lmrl_gym_amd  # noqa: F401

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    from lmrl_gym_amd import _lib
    return _lib.require_gpu()


def _bf(x):
    return x.to(torch.bfloat16)


# ------------------------------------------------------------------ GEMM + epilogues
@pytest.mark.parametrize("M,N,K", [(1024, 768, 768), (1024, 2304, 768), (8192, 3072, 768), (1000, 768, 3072), (77, 128, 64),
                                   (1, 64, 64), (300, 1024, 1024)])
def test_gemm_bf16_epilogues(dev, M, N, K):
    from lmrl_gym_amd import _lib
    from oracle.gpt2 import gelu_new
    L = _lib.lib()
    g = torch.Generator().manual_seed(M + N + K)
    A = _bf(torch.randn(M, K, generator=g)); W = _bf(torch.randn(N, K, generator=g) * 0.05); b = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g)
    ref = A.double() @ W.double().t() + b.double()
    Ad, Wd, bd = A.to(dev), W.to(dev), b.to(dev)
    tol = dict(rtol=2e-2, atol=2e-2 * float(ref.abs().max()) / 8)   # bf16 outputs: 8 mantissa bits
    # asymmetric inputs -> a transposed store would fail (guide §5.4 rule 16)
    for epi, expect, out_dtype in [(0, ref, torch.bfloat16), (1, gelu_new(ref), torch.bfloat16), (3, ref, torch.float32),
                                   (4, torch.relu(ref), torch.bfloat16), (2, ref + R.double(), torch.float32)]:
        C = R.to(dev).clone() if epi == 2 else torch.zeros(M, N, dtype=out_dtype, device=dev)
        _lib.check(L.lmrl_gemm_bf16(_lib.ptr(Ad), _lib.ptr(Wd), _lib.ptr(bd), _lib.ptr(C), M, N, K, K, N, N, epi, _lib.stream_ptr()))
        got = C.cpu().double()
        if out_dtype == torch.float32:
            torch.testing.assert_close(got, expect, rtol=1e-3, atol=1e-3 * float(ref.abs().max()))
        else:
            torch.testing.assert_close(got, expect, **tol)


# ------------------------------------------------------------------ forward: chunked prefill + decode vs full-sequence oracle
@pytest.mark.parametrize("ln_fusion", [1, 0])    # LayerNorm folded into the neighbouring GEMMs (default) / stand-alone LN launches (per-session flag)
@pytest.mark.parametrize("cfgname", ["tiny", "small2", "medium2", "large2"])
def test_gpt2_forward_kv_cache(dev, cfgname, ln_fusion):
    from lmrl_gym_amd.gpt2 import FWD_LN_STANDALONE, GPT2Config, GPT2Engine, init_hf_style_state_dict
    from oracle import gpt2 as O
    cfg = dict(tiny=GPT2Config(2, 2, 128, 512, 1000, 64), small2=GPT2Config(2, 12, 768, 3072, 50257, 128),
               medium2=GPT2Config(2, 16, 1024, 4096, 5000, 64),                # GPT-2-medium width: the NQ = 4 LayerNorm-fold configuration
               large2=GPT2Config(2, 20, 1280, 5120, 3000, 64))[cfgname]        # GPT-2-large width (20 heads): NQ = 5
    sd = init_hf_style_state_dict(cfg, seed=1)
    g = torch.Generator().manual_seed(2)
    for k in sd:   # non-trivial LN / bias values
        if sd[k].dim() == 1:
            sd[k] = sd[k] + 0.3 * torch.randn(sd[k].shape, generator=g)
    sd["wpe.weight"] = sd["wpe.weight"] + 0.05          # a non-zero row mean in the residual stream (exercises mu * colsum)
    sd = O.round_weights_to_bf16(sd)
    eng = GPT2Engine(cfg, sd, dev)
    B, T = 5, 40
    ids = torch.randint(0, cfg.vocab, (B, T), generator=g)
    ref_logits, ref_hid = O.forward(sd, ids, cfg.n_head, dtype=torch.float64, return_hidden=True)
    ses = eng.session(B, 48, flags=0 if ln_fusion else FWD_LN_STANDALONE)
    # schedule: chunk of 8 with ragged counts, then single-token decode steps with some envs idle, then another chunk
    consumed = np.zeros(B, dtype=int)
    plan = [(8, [8, 5, 1, 0, 7]), (1, [1, 1, 1, 1, 0]), (1, [1, 0, 1, 1, 1]), (16, [16, 9, 3, 8, 12]), (1, [1, 1, 1, 1, 1]), (8, [4, 0, 8, 6, 8])]
    for C, cnts in plan:
        toks = torch.zeros(B, C, dtype=torch.int32)
        for b, c in enumerate(cnts):
            toks[b, :c] = ids[b, consumed[b]:consumed[b] + c]
        allh = torch.zeros(B * C, cfg.d_model, dtype=torch.bfloat16, device=dev)
        toks_d, cnt_d = toks.reshape(-1).to(dev), torch.tensor(cnts, dtype=torch.int32, device=dev)
        last = ses.forward(toks_d, cnt_d, C, all_hidden=allh).float().cpu()
        allh = allh.float().cpu().reshape(B, C, -1)
        for b, c in enumerate(cnts):
            for j in range(c):
                torch.testing.assert_close(allh[b, j].double(), ref_hid[b, consumed[b] + j], rtol=5e-2, atol=5e-2)
            if c:
                torch.testing.assert_close(last[b].double(), ref_hid[b, consumed[b] + c - 1], rtol=5e-2, atol=5e-2)
            consumed[b] += c
        assert ses.len.cpu().tolist() == consumed.tolist()
    # tighter aggregate check: relative Frobenius error of the final hidden states
    err = (last.double() - torch.stack([ref_hid[b, consumed[b] - 1] for b in range(B)])).norm() / torch.stack([ref_hid[b, consumed[b] - 1] for b in range(B)]).norm()
    assert err < 1.5e-2, float(err)


@pytest.mark.parametrize("size", ["medium", "large"])
def test_gpt2_medium_and_large_full_depth_vs_oracle(dev, size):
    """configs[3] / configs[4] models at FULL depth: GPT-2-medium (24 layers, d = 1024, 16 heads — the chess PPO policy
    and the Twenty-Questions guesser) and GPT-2-large (36 layers, d = 1280, 20 heads — the Twenty-Questions oracle model here): final
    hidden states of the bf16 engine through chunked prefill + decode steps against the float64 oracle on the same bf16-rounded weights."""
    from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine, init_hf_style_state_dict
    from oracle import gpt2 as O
    cfg = dict(medium=GPT2Config.gpt2_medium, large=GPT2Config.gpt2_large)[size](8192)      # (a short vocabulary keeps the oracle's LM head cheap)
    sd = O.round_weights_to_bf16(init_hf_style_state_dict(cfg, seed=2))
    eng = GPT2Engine(cfg, sd, dev)
    B, T = 3, 29
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, cfg.vocab, (B, T), generator=g)
    _, ref_hid = O.forward(sd, ids, cfg.n_head, dtype=torch.float64, return_hidden=True)
    ses = eng.session(B, 32)
    pos = 0
    for C, n in [(16, 16), (8, 8), (1, 1), (1, 1), (1, 1), (1, 1), (1, 1)]:
        toks = torch.zeros(B, C, dtype=torch.int32)
        toks[:, :n] = ids[:, pos:pos + n]
        last = ses.forward(toks.reshape(-1).to(dev), torch.full((B,), n, dtype=torch.int32, device=dev), C).float().cpu().double()
        pos += n
        ref = ref_hid[:, pos - 1]
        err = float((last - ref).norm() / ref.norm())
        assert err < 2.5e-2, (size, pos, err)                     # bf16 engine, 24 / 36 blocks: relative Frobenius error of the hidden state
        torch.testing.assert_close(last, ref, rtol=8e-2, atol=8e-2)
    assert ses.len.cpu().tolist() == [T] * B


def test_gpt2_small_full_depth_vs_oracle(dev):
    """The BASELINE configuration's model (GPT-2-small: 12 layers, d = 768, V = 50257): final hidden states and greedy tokens of
    the bf16 engine against the float64 oracle on the same bf16-rounded weights, through chunked prefill + decode steps."""
    from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine, SampleParams, init_hf_style_state_dict
    from oracle import gpt2 as O
    cfg = GPT2Config.gpt2_small()
    sd = O.round_weights_to_bf16(init_hf_style_state_dict(cfg, seed=0))
    sd["wte.weight"] = (sd["wte.weight"] * 4).to(torch.bfloat16).float()        # wider logit margins for the argmax comparison
    eng = GPT2Engine(cfg, sd, dev)
    B, T = 3, 21
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, cfg.vocab, (B, T), generator=g)
    ref_logits, ref_hid = O.forward(sd, ids, cfg.n_head, dtype=torch.float64, return_hidden=True)
    ses = eng.session(B, 32)
    pos = 0
    for C, n in [(8, 8), (8, 8), (1, 1), (1, 1), (1, 1), (1, 1), (1, 1)]:
        toks = torch.zeros(B, C, dtype=torch.int32)
        toks[:, :n] = ids[:, pos:pos + n]
        last = ses.forward(toks.reshape(-1).to(dev), torch.full((B,), n, dtype=torch.int32, device=dev), C).float().cpu()
        pos += n
        ref = ref_hid[:, pos - 1]
        err = (last.double() - ref).norm() / ref.norm()
        assert err < 2e-2, (pos, float(err))
        tok, _ = ses.sample(SampleParams(0.0, 0, 0, 0, 0.0, 0.0, 0))
        top2 = ref_logits[:, pos - 1, : cfg.vocab].topk(2, dim=-1)
        for b in range(B):
            if float(top2.values[b, 0] - top2.values[b, 1]) > 0.25:
                assert int(tok[b]) == int(top2.indices[b, 0]), (pos, b)
    assert pos == T


# ------------------------------------------------------------------ fused LM head + sampler
def _engine_and_hidden(dev, B, vocab=5003, d=128):
    from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine, init_hf_style_state_dict
    from oracle import gpt2 as O
    cfg = GPT2Config(1, d // 64, d, 256, vocab, 32)
    sd = O.round_weights_to_bf16(init_hf_style_state_dict(cfg, seed=3))
    sd["wte.weight"] = (sd["wte.weight"] * 20).to(torch.bfloat16).float()   # spread the logits
    eng = GPT2Engine(cfg, sd, dev)
    hid = _bf(torch.randn(B, d, generator=torch.Generator().manual_seed(4)))
    ses = eng.session(B, 8)
    logits = hid.double() @ sd["wte.weight"].double().t()
    return cfg, eng, ses, hid.to(dev), logits


def test_sampler_greedy_logprob_and_logits(dev):
    from lmrl_gym_amd.gpt2 import SampleParams
    B = 300
    cfg, eng, ses, hid, logits = _engine_and_hidden(dev, B)
    lo = torch.zeros(B, cfg.vocab_padded, device=dev)
    active = torch.ones(B, dtype=torch.uint8, device=dev); active[7] = 0
    tok, lp = ses.sample(SampleParams(0.0, 0, 1, 0, 0.0, 0.0, 50256), hidden=hid, logits_out=lo, active=active)
    tok, lp, lo = tok.cpu(), lp.cpu(), lo.cpu()[:, : cfg.vocab]
    torch.testing.assert_close(lo.double(), logits, rtol=1e-3, atol=2e-3)
    exp_lp = torch.log_softmax(logits, -1)
    for b in range(B):
        if b == 7:
            assert tok[b] == 50256
            continue
        top2 = logits[b].topk(2).values
        if top2[0] - top2[1] > 1e-2:
            assert tok[b] == logits[b].argmax()
        assert abs(float(lp[b]) - float(exp_lp[b, tok[b]])) < 5e-3


def test_sampler_gumbel_stream_matches_documented_scheme(dev):
    """Sampling == argmax(logits/T + Gumbel(philox(row, col//4, step))) on the materialised logits, for the fused
    epilogue and the top-k kernel alike; top-k keeps exactly the k largest (ties kept)."""
    from lmrl_gym_amd import _lib
    from lmrl_gym_amd.gpt2 import SampleParams
    from oracle.gpt2 import gumbel_noise
    B = 200
    cfg, eng, ses, hid, _ = _engine_and_hidden(dev, B)
    lo = torch.zeros(B, cfg.vocab_padded, device=dev)
    seed, step, T = 0x1234567811223344, 5, 0.7
    tok, lp = ses.sample(SampleParams(T, 0, seed, step, 0.0, 0.0, 0), hidden=hid, logits_out=lo)
    z = lo.cpu().numpy()[:, : cfg.vocab].astype(np.float32)
    g = gumbel_noise(B, cfg.vocab, seed, step)
    score = z / np.float32(T) + g
    exp_tok = score.argmax(1)
    tok = tok.cpu().numpy()
    srt = np.sort(score, 1)
    safe = (srt[:, -1] - srt[:, -2]) > 1e-3        # device log is a fast approximation
    assert safe.mean() > 0.9 and np.array_equal(tok[safe], exp_tok[safe])
    lse = np.log(np.exp((z / T) - (z / T).max(1, keepdims=True)).sum(1)) + (z / T).max(1)
    np.testing.assert_allclose(lp.cpu().numpy()[safe], (z / T)[np.arange(B), tok][safe] - lse[safe], atol=5e-3)
    # top-k path on the same logits / stream
    for k in (1, 5, 50, cfg.vocab):
        tk = torch.zeros(B, dtype=torch.int32, device=dev); lpk = torch.zeros(B, device=dev)
        p = SampleParams(T, k, seed, step, 0.0, 0.0, 0)
        _lib.check(_lib.lib().lmrl_sample_logits(_lib.ptr(lo), cfg.vocab_padded, B, cfg.vocab, ctypes.byref(p), None,
                                                 _lib.ptr(tk), _lib.ptr(lpk), _lib.stream_ptr()))
        tk = tk.cpu().numpy()
        kth = np.sort(z, 1)[:, -k][:, None]
        masked = np.where(z >= kth, score, -np.inf)
        e = masked.argmax(1)
        m2 = np.sort(masked, 1)
        ok = (m2[:, -1] - m2[:, -2]) > 1e-3 if k > 1 else np.ones(B, bool)
        assert np.array_equal(tk[ok], e[ok]), k
        assert np.all(z[np.arange(B), tk] >= kth[:, 0])
        if k == cfg.vocab:
            assert np.array_equal(tk[safe], tok[safe])


def test_sampler_jax_stream_equals_jax_random_categorical_restatement(dev):
    """LMRL_RNG_JAX: the token of every row == `jax.random.categorical(key, logits / T)` as restated in
    oracle/jax_random.py (threefry2x32-20 pinned by the Random123 vectors in tests/test_jax_prng.py) on the logits the device itself
    materialised — fused LM-head epilogue and the top-k kernel alike; ONE key for the whole [B, V] noise array, word index = row * V +
    column, so the draw of a row depends on the batch it is in, as in JAX."""
    from lmrl_gym_amd import _lib, jax_prng as JP
    from lmrl_gym_amd.gpt2 import RNG_JAX, SampleParams
    from oracle import jax_random as JR
    for B in (200, 37):                               # even and odd word counts (odd: the iota is padded with one 0)
        cfg, eng, ses, hid, _ = _engine_and_hidden(dev, B)
        lo = torch.zeros(B, cfg.vocab_padded, device=dev)
        key = JP.split(JP.prng_key(1234 + B))[1]
        T = 0.7
        p = SampleParams(T, 0, JP.key_to_seed(key), 0, 0.0, 0.0, 0, None, 0.0, RNG_JAX)
        tok, lp = ses.sample(p, hidden=hid, logits_out=lo)
        z = lo.cpu().numpy()[:, : cfg.vocab].astype(np.float32)
        g = JR.gumbel(np.array(key, np.uint32), (B, cfg.vocab))
        score = z / np.float32(T) + g
        srt = np.sort(score, 1)
        safe = (srt[:, -1] - srt[:, -2]) > 1e-5        # float log: a few ulp between implementations
        tok = tok.cpu().numpy()
        assert safe.mean() > 0.98 and np.array_equal(tok[safe], score.argmax(1)[safe])
        assert np.array_equal(tok[safe], JR.categorical(np.array(key, np.uint32), z / np.float32(T))[safe])
        zt = z / np.float32(T)
        lse = np.log(np.exp(zt - zt.max(1, keepdims=True)).sum(1)) + zt.max(1)
        np.testing.assert_allclose(lp.cpu().numpy(), zt[np.arange(B), tok] - lse, atol=5e-3)
        # a different key gives different draws; the Philox mode is untouched by the new field
        tok2, _ = ses.sample(SampleParams(T, 0, JP.key_to_seed(JP.split(key)[0]), 0, 0.0, 0.0, 0, None, 0.0, RNG_JAX), hidden=hid)
        assert (tok2.cpu().numpy() != tok).mean() > 0.2          # (peaked rows re-draw their mode)
        for k in (5, cfg.vocab):
            tk = torch.zeros(B, dtype=torch.int32, device=dev); lpk = torch.zeros(B, device=dev)
            pk = SampleParams(T, k, JP.key_to_seed(key), 0, 0.0, 0.0, 0, None, 0.0, RNG_JAX)
            _lib.check(_lib.lib().lmrl_sample_logits(_lib.ptr(lo), cfg.vocab_padded, B, cfg.vocab, ctypes.byref(pk), None,
                                                     _lib.ptr(tk), _lib.ptr(lpk), _lib.stream_ptr()))
            kth = np.sort(z, 1)[:, -k][:, None]
            masked = np.where(z >= kth, score, -np.inf)      # TopKLogitsWarper then categorical on the same [B, V] noise array
            m2 = np.sort(masked, 1)
            ok = (m2[:, -1] - m2[:, -2]) > 1e-5
            assert np.array_equal(tk.cpu().numpy()[ok], masked.argmax(1)[ok]), k


def test_sampler_distribution_chi_square(dev):
    """Identical rows -> empirical token frequencies follow softmax(logits/T)."""
    from lmrl_gym_amd.gpt2 import SampleParams
    B = 8192
    cfg, eng, ses, hid, logits = _engine_and_hidden(dev, B, vocab=300, d=128)
    hid = hid[:1].repeat(B, 1).contiguous()
    p = torch.softmax(logits[0] / 1.3, -1).numpy()
    counts = np.zeros(cfg.vocab)
    for step in range(4):
        tok, _ = ses.sample(SampleParams(1.3, 0, 99, step, 0.0, 0.0, 0), hidden=hid)
        counts += np.bincount(tok.cpu().numpy(), minlength=cfg.vocab)
    n = counts.sum()
    keep = p * n >= 5
    chi2 = (((counts - p * n) ** 2) / (p * n))[keep].sum() + ((counts[~keep].sum() - p[~keep].sum() * n) ** 2) / max(p[~keep].sum() * n, 1e-9)
    dof = keep.sum()
    assert chi2 < dof + 5 * np.sqrt(2 * dof), (chi2, dof)


def test_sampler_top_p_and_top_k_warpers(dev):
    """Nucleus sampling with the HF warper semantics (temperature -> top_k -> top_p; the descending-sorted prefix whose
    cumulative probability reaches top_p is kept): support, renormalised log-probs and empirical frequencies."""
    import ctypes
    from lmrl_gym_amd import _lib
    from lmrl_gym_amd.gpt2 import SampleParams
    V, B = 257, 4096
    g = torch.Generator().manual_seed(7)
    base = torch.randn(V, generator=g) * 2.0
    logits = base[None].repeat(B, 1).contiguous().to(dev)
    L = _lib.lib()
    for temp, top_k, top_p in [(1.0, 0, 0.8), (0.7, 0, 0.5), (1.2, 20, 0.9), (1.0, 5, 0.999)]:
        # reference kept set: oracle/warpers.py (pinned to the HF logits processors in tests/test_oracle_warpers.py)
        from oracle import warpers as OW
        zo, keep_o = OW.warp(base.double().numpy(), temp, top_k, top_p)
        keep2 = torch.from_numpy(keep_o)
        final = np.exp(OW.log_probs(zo, keep_o))
        counts = np.zeros(V)
        for step in range(3):
            sp = SampleParams(temp, top_k, 11, step, 0.0, 0.0, 0, None, top_p)
            tok = torch.zeros(B, dtype=torch.int32, device=dev); lp = torch.zeros(B, dtype=torch.float32, device=dev)
            _lib.check(L.lmrl_sample_logits(_lib.ptr(logits), V, B, V, ctypes.byref(sp), None, _lib.ptr(tok), _lib.ptr(lp), _lib.stream_ptr()))
            t = tok.cpu().numpy()
            assert keep2.numpy()[t].all(), (temp, top_k, top_p)                       # never outside the nucleus
            np.testing.assert_allclose(lp.cpu().numpy(), np.log(final[t]), rtol=1e-4, atol=1e-4)
            counts += np.bincount(t, minlength=V)
        n = counts.sum()
        big = final * n >= 5
        chi2 = (((counts - final * n) ** 2) / np.maximum(final * n, 1e-12))[big].sum()
        dof = max(int(big.sum()) - 1, 1)
        assert chi2 < dof + 6 * np.sqrt(2 * dof) + 10, (chi2, dof, temp, top_k, top_p)


@pytest.mark.parametrize("V,ld", [(50257, 50304), (20000, 20000), (65000, 65536)])
def test_register_row_warper_kernel_equals_the_strided_one(dev, V, ld):
    """Round 5: top-k / top-p selection with the whole row in the registers of a 1024-thread workgroup (`topk_sample_reg_kernel`: one read of the
    row instead of ten) against the strided-row kernel pinned to the HF warper semantics above (`lmrl_sampler_set_variant(1)`): the same
    thresholds and the same draw — every sampled token identical, log-probabilities to fp32 rounding — over top-k only, top-p only, both, greedy,
    inactive rows, both random streams, a GPT-2-sized and two other row lengths."""
    import ctypes
    from lmrl_gym_amd import _lib
    from lmrl_gym_amd.gpt2 import RNG_JAX, SampleParams
    L = _lib.lib()
    B = 96
    g = torch.Generator().manual_seed(V)
    logits = torch.zeros(B, ld)
    logits[:, :V] = torch.randn(B, V, generator=g) * 3.0
    n7 = (V - 1) // 7
    logits[:, 0:7 * n7:7] = logits[:, 1:7 * n7 + 1:7]                                           # exact ties across columns
    logits = logits.to(dev)
    active = torch.ones(B, dtype=torch.uint8); active[5::11] = 0
    active = active.to(dev)
    cases = [(1.0, 40, 0.0, 0), (0.7, 0, 0.9, 0), (1.3, 50, 0.95, 0), (1.0, 1, 0.0, 0), (0.0, 40, 0.0, 0), (1.0, 5, 0.5, RNG_JAX), (1.0, 1024, 0.0, 0),
             (1.0, 3000, 0.8, 0)]
    for temp, top_k, top_p, rng in cases:
        outs = []
        for variant in (1, 0):
            L.lmrl_sampler_set_variant(variant)
            try:
                sp = SampleParams(temp, top_k, 0xC0FFEE, 3, 0.0, 0.0, 7, None, top_p, rng)
                tok = torch.zeros(B, dtype=torch.int32, device=dev); lp = torch.zeros(B, dtype=torch.float32, device=dev)
                _lib.check(L.lmrl_sample_logits(_lib.ptr(logits), ld, B, V, ctypes.byref(sp), _lib.ptr(active), _lib.ptr(tok), _lib.ptr(lp), _lib.stream_ptr()))
                torch.cuda.synchronize()
                outs.append((tok.cpu().numpy(), lp.cpu().numpy()))
            finally:
                L.lmrl_sampler_set_variant(0)
        assert np.array_equal(outs[0][0], outs[1][0]), (V, temp, top_k, top_p, rng)
        np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=1e-5, atol=2e-5)
        assert (outs[0][0][active.cpu().numpy() == 0] == 7).all() and (outs[0][0] < V).all()


@pytest.mark.parametrize("B,vocab,d", [(1024, 50257, 128), (200, 9000, 128), (130, 300, 128)])
def test_fused_topk_candidate_path_equals_the_materialised_one(dev, B, vocab, d):
    """Round 5 (north_star: "fused token-sampling (top-k ...)"): with 0 < top_k <= 64 the LM-head epilogue keeps 8 candidates per (row, 128-column tile)
    and the reduce kernel selects, checks exactness, applies top-p and draws — no logits in HBM (`lm_topc_epilogue`, `topc_reduce_sample_kernel`).
    Against the materialised path (`lmrl_sampler_set_variant(2)`: logits written, register-row kernel; pinned to the HF warper semantics above): every
    sampled token identical, log-probabilities to fp32 rounding — persistent kernel (1024 x GPT-2 vocabulary) and one-tile kernel, top-k alone and with
    top-p, k = 1 .. 64, steered rows, inactive rows; and with a vocabulary whose large logits CLUSTER in a few tiles, so that the exactness check hands
    rows back to the materialised path (the flagged-row list is read back: it was used)."""
    from lmrl_gym_amd import _lib
    from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine, SampleParams, init_hf_style_state_dict
    L = _lib.lib()
    cfg = GPT2Config(1, d // 64, d, 256, vocab, 32)
    g = torch.Generator().manual_seed(vocab)
    hid = _bf(torch.randn(B, d, generator=g)).to(dev)
    steer = torch.randint(0, vocab, (B,), generator=g).to(torch.int32); steer[::3] = -1
    active = torch.ones(B, dtype=torch.uint8); active[5::17] = 0
    steer, active = steer.to(dev), active.to(dev)
    tiles_n = -(-vocab // 128)
    for clustered in (False, True, "ties"):
        sd = init_hf_style_state_dict(cfg, seed=3)
        w = sd["wte.weight"] * 20
        if clustered == "ties":      # few distinct embedding rows and a coarse grid of hidden values: most logits tie EXACTLY with many others (in and across tiles)
            base = torch.round(torch.randn(37, d, generator=g) * 2) / 2
            w = base[torch.randint(0, 37, (vocab,), generator=g)]
            hid_c = _bf(torch.round(hid.cpu().float() * 2) / 2).to(dev)
        elif clustered:              # twelve columns inside one tile carry most rows' largest logits (a shared direction, different gains)
            u = torch.randn(d, generator=g)
            c0 = 128 * (tiles_n // 2) + 7 if tiles_n > 2 else 3
            for i in range(12):
                w[c0 + 5 * i] = u * (0.6 + 0.03 * i) + w[c0 + 5 * i] * 0.2
            hid_c = _bf(hid.cpu().float() + u[None] * 0.8).to(dev)
        sd["wte.weight"] = w.to(torch.bfloat16).float()
        eng = GPT2Engine(cfg, sd, dev)
        ses = eng.session(B, 8)
        h = hid_c if clustered else hid
        lo = torch.zeros(B, cfg.vocab_padded, device=dev)
        flagged = 0
        for temp, top_k, top_p, strength in [(1.0, 40, 0.0, 0.0), (0.8, 50, 0.9, 0.0), (1.0, 1, 0.0, 0.0), (1.3, 64, 0.0, 6.0), (1.0, 5, 0.6, 0.0), (0.7, 20, 0.95, 3.0)]:
            if top_k >= vocab:
                continue
            outs = []
            for variant in (2, 0):
                L.lmrl_sampler_set_variant(variant)
                try:
                    lo.fill_(float("nan"))
                    sp = SampleParams(temp, top_k, 0xBEEF, 5, strength, 0.0, 9, None, top_p, 0)
                    tok, lp = ses.sample(sp, hidden=h, steer_tok=steer, active=active, logits_out=lo)
                    torch.cuda.synchronize()
                    outs.append((tok.cpu().numpy().copy(), lp.cpu().numpy().copy()))
                    if variant == 0:
                        fb = ses.sample_ws[_lib.lib().lmrl_sample_fb_offset(B, cfg.vocab_padded):].view(torch.int32)
                        n_fb = int(fb[0].item())
                        flagged += n_fb
                        rows = fb[16 + 64: 16 + 64 + n_fb].cpu().numpy()
                        assert len(set(rows.tolist())) == n_fb and all(fb[16 + r // 128].item() == 1 for r in rows)
                        touched = ~torch.isnan(lo[:, 0]).cpu().numpy()               # logits exist only for the flagged rows' 128-row blocks
                        assert touched.sum() <= 128 * max(n_fb, 0) and all(touched[r] for r in rows)
                finally:
                    L.lmrl_sampler_set_variant(0)
            assert np.array_equal(outs[0][0], outs[1][0]), (clustered, temp, top_k, top_p)
            np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=1e-5, atol=2e-5)
            assert (outs[0][0][active.cpu().numpy() == 0] == 9).all()
        if clustered is True and tiles_n > 2:
            assert flagged > 0, "the clustered vocabulary was meant to exercise the hand-back to the materialised path"
        elif clustered is False and vocab > 5000:
            assert flagged <= 2


@pytest.mark.parametrize("B,vocab,d,nq", [(1024, 50257, 128, 2), (200, 9000, 128, 2), (130, 300, 128, 1), (300, 20000, 256, 2)])
def test_fused_topk_for_the_ilql_value_policy_head(dev, B, vocab, d, nq):
    """Round 6 (north_star: "fused token-sampling (top-k / logit-perturb from the ILQL Q-head)", value_rl_base/gpt2/generation.py:97-119): the candidate
    epilogue on logits = pi_beta + beta * min(q1, q2) — candidates are taken after the perturbed logits are formed in registers — against the
    materialised path of the same three-operand head (`lmrl_sampler_set_variant(2)`): every sampled token identical, log-probs to fp32 rounding;
    one and two Q heads, top-k with / without top-p, steered and inactive rows, and a clustered vocabulary that forces the hand-back (whose
    materialised logits are the three-operand ones)."""
    from lmrl_gym_amd import _lib
    from lmrl_gym_amd.gpt2 import SAMPLE_WANT_LOGITS, GPT2Config, GPT2Engine, SampleParams, init_hf_style_state_dict
    L = _lib.lib()
    cfg = GPT2Config(1, d // 64, d, 256, vocab, 32)
    g = torch.Generator().manual_seed(vocab + nq)
    hid = _bf(torch.randn(B, d, generator=g)).to(dev)
    steer = torch.randint(0, vocab, (B,), generator=g).to(torch.int32); steer[::3] = -1
    active = torch.ones(B, dtype=torch.uint8); active[5::17] = 0
    steer, active = steer.to(dev), active.to(dev)
    tiles_n = -(-vocab // 128)
    qh = [_bf(torch.relu(torch.randn(B, d, generator=g))).to(dev) for _ in range(nq)]
    for clustered in (False, True):
        sd = init_hf_style_state_dict(cfg, seed=3)
        w = sd["wte.weight"] * 20
        qw = [torch.randn(cfg.vocab_padded, d, generator=g) * 0.3 for _ in range(nq)]
        qb = [torch.randn(cfg.vocab_padded, generator=g).to(dev) for _ in range(nq)]
        if clustered:                # the Q heads carry the cluster: twelve columns of one tile get most rows' largest PERTURBED logits
            c0 = 128 * (tiles_n // 2) + 7 if tiles_n > 2 else 3
            for i in range(12):
                for k in range(nq):
                    qw[k][c0 + 5 * i] = 0.35 + 0.01 * i
        qw = [_bf(x).to(dev) for x in qw]
        sd["wte.weight"] = w.to(torch.bfloat16).float()
        eng = GPT2Engine(cfg, sd, dev)
        ses = eng.session(B, 8)
        q1 = (qh[0], qw[0], qb[0])
        q2 = (qh[1], qw[1], qb[1]) if nq == 2 else None
        lo = torch.zeros(B, cfg.vocab_padded, device=dev)
        flagged = 0
        for temp, top_k, top_p, strength, beta in [(1.0, 40, 0.0, 0.0, 1.0), (0.8, 50, 0.9, 0.0, 4.0), (1.0, 1, 0.0, 0.0, 0.5), (1.3, 64, 0.0, 6.0, 2.0), (0.7, 20, 0.95, 3.0, 16.0)]:
            if top_k >= vocab:
                continue
            outs = []
            for variant in (2, 0):
                L.lmrl_sampler_set_variant(variant)
                try:
                    lo.fill_(float("nan"))
                    sp = SampleParams(temp, top_k, 0xFEED, 9, strength, beta, 9, None, top_p, 0)
                    tok, lp = ses.sample(sp, hidden=hid, steer_tok=steer, active=active, logits_out=lo, q1=q1, q2=q2)
                    torch.cuda.synchronize()
                    outs.append((tok.cpu().numpy().copy(), lp.cpu().numpy().copy()))
                    if variant == 0:
                        fb = ses.sample_ws[L.lmrl_sample_fb_offset(B, cfg.vocab_padded):].view(torch.int32)
                        flagged += int(fb[0].item())
                        assert torch.isnan(lo[:, 0]).sum().item() >= B - 128 * int(fb[0].item())       # no logits in HBM but the handed-back blocks'
                    else:
                        mat = lo.clone()
                finally:
                    L.lmrl_sampler_set_variant(0)
            assert np.array_equal(outs[0][0], outs[1][0]), (clustered, temp, top_k, top_p, beta)
            np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=1e-5, atol=2e-5)
            assert (outs[0][0][active.cpu().numpy() == 0] == 9).all()
        # LMRL_SAMPLE_WANT_LOGITS: the caller reads logits_out — materialised although top_k <= 64 (same tokens, the same logits as variant 2 wrote)
        sp = SampleParams(0.7, 20, 0xFEED, 9, 3.0, 16.0, 9, None, 0.95, 0, SAMPLE_WANT_LOGITS)
        lo.fill_(float("nan"))
        tok, _ = ses.sample(sp, hidden=hid, steer_tok=steer, active=active, logits_out=lo, q1=q1, q2=q2)
        torch.cuda.synchronize()
        assert np.array_equal(tok.cpu().numpy(), outs[0][0]) and torch.equal(lo[:, :vocab], mat[:, :vocab])
        if clustered and tiles_n > 2:
            assert flagged > 0, "the clustered Q heads were meant to exercise the hand-back to the materialised three-operand path"


@pytest.mark.parametrize("B,vocab,d,nq", [(1024, 50257, 128, 0), (200, 9000, 128, 2), (130, 300, 128, 0), (300, 20000, 256, 1)])
def test_fused_remaining_warper_forms_equal_the_materialised_path(dev, B, vocab, d, nq):
    """Round 6 (`train_ppo_gpt2.py:98-99, 218-227`: policy_top_k / policy_top_p, either alone): the candidate path for (a) top-p
    WITHOUT top-k — the epilogue also writes every tile's probability mass, the reduce kernel forms the row total from them, selects the nucleus among
    the candidates above every tile's hidden bound and hands a row back when its nucleus reaches further; (b) 64 < top_k <= 256 (pre-filter by the
    per-tile maxima); (c) the LMRL_RNG_JAX stream (the draw happens in the reduce kernel: threefry words for the kept columns only).  Against the
    materialised path (`lmrl_sampler_set_variant(2)`) on peaked logits (a trained policy's shape) and on flat ones (every top-p row goes back): every
    sampled token identical, log-probs to fp32 rounding; policy-only head (persistent and one-tile kernels) and the ILQL value policy's heads."""
    from lmrl_gym_amd import _lib
    from lmrl_gym_amd.gpt2 import RNG_JAX, GPT2Config, GPT2Engine, SampleParams, init_hf_style_state_dict
    L = _lib.lib()
    cfg = GPT2Config(1, d // 64, d, 256, vocab, 32)
    g = torch.Generator().manual_seed(vocab + nq)
    hid = _bf(torch.randn(B, d, generator=g)).to(dev)
    steer = torch.randint(0, vocab, (B,), generator=g).to(torch.int32); steer[::3] = -1
    active = torch.ones(B, dtype=torch.uint8); active[5::17] = 0
    steer, active = steer.to(dev), active.to(dev)
    qh = [_bf(torch.relu(torch.randn(B, d, generator=g))).to(dev) for _ in range(nq)]
    qw = [_bf(torch.randn(cfg.vocab_padded, d, generator=g) * 0.3).to(dev) for _ in range(nq)]
    qb = [torch.randn(cfg.vocab_padded, generator=g).to(dev) for _ in range(nq)]
    q1 = (qh[0], qw[0], qb[0]) if nq >= 1 else None
    q2 = (qh[1], qw[1], qb[1]) if nq == 2 else None
    fb_off = L.lmrl_sample_fb_offset(B, cfg.vocab_padded)
    for scale, peaked in ((60.0, True), (8.0, False)):
        sd = init_hf_style_state_dict(cfg, seed=3)
        sd["wte.weight"] = (sd["wte.weight"] * scale).to(torch.bfloat16).float()
        eng = GPT2Engine(cfg, sd, dev)
        ses = eng.session(B, 8)
        lo = torch.zeros(B, cfg.vocab_padded, device=dev)
        cases = [(1.0, 0, 0.9, 0.0, 0), (0.7, 0, 0.5, 4.0, 0), (1.0, 0, 0.97, 0.0, RNG_JAX), (1.0, 128, 0.0, 0.0, 0), (0.9, 256, 0.95, 3.0, 0),
                 (1.0, 40, 0.0, 0.0, RNG_JAX), (0.8, 100, 0.9, 0.0, RNG_JAX)]
        handed = {}
        for temp, top_k, top_p, strength, rng in cases:
            if top_k >= vocab:
                continue
            outs = []
            for variant in (2, 0):
                L.lmrl_sampler_set_variant(variant)
                try:
                    lo.fill_(float("nan"))
                    sp = SampleParams(temp, top_k, 0xD1CE, 4, strength, 2.0 if nq else 0.0, 9, None, top_p, rng)
                    tok, lp = ses.sample(sp, hidden=hid, steer_tok=steer, active=active, logits_out=lo, q1=q1, q2=q2)
                    torch.cuda.synchronize()
                    outs.append((tok.cpu().numpy().copy(), lp.cpu().numpy().copy()))
                    if variant == 0:
                        touched = int((~torch.isnan(lo[:, 0])).sum().item())
                        if top_k == 0 or top_k <= 64 or top_k <= cfg.vocab_padded // 128:
                            fb = ses.sample_ws[fb_off:fb_off + 4 * (16 + 64 + B)].view(torch.int32)
                            n_fb = int(fb[0].item())
                            handed[(temp, top_k, top_p, rng)] = n_fb
                            assert touched <= 128 * n_fb                   # logits exist only for handed-back rows' 128-row blocks
                        else:                                              # more than 64 and more than the tile count: materialised by choice
                            assert touched == B
                finally:
                    L.lmrl_sampler_set_variant(0)
            assert np.array_equal(outs[0][0], outs[1][0]), (peaked, temp, top_k, top_p, rng)
            np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=1e-5, atol=2e-5)
            assert (outs[0][0][active.cpu().numpy() == 0] == 9).all()
        n_act = int(active.sum().item())
        if peaked and vocab > 5000:       # the fused forms did the work: few rows handed back
            assert all(v <= n_act // 8 for v in handed.values()), handed
        if not peaked and vocab > 5000 and nq == 0:   # flat logits (the Q heads' perturbation would peak them): a 0.9 nucleus is thousands of tokens — every active row went back, same tokens
            assert handed[(1.0, 0, 0.9, 0)] == n_act, handed


@pytest.mark.parametrize("ilql", [False, True])
def test_fused_topk_at_bench_size_against_the_float64_oracle(dev, ilql):
    """The fused top-k path at the size the bench runs it (1024 rows x the GPT-2 vocabulary, d = 768), not against another HIP path but against
    the warper SEMANTICS on float64 logits (HF TopK then softmax, train_ppo_gpt2.py:98-99, 218-227) and the documented noise stream
    (oracle/gpt2.py::gumbel_noise): every sampled token lies in the oracle's top-k set (rows whose k-th / (k+1)-th logits are closer than the
    bf16-product rounding are skipped), its log-probability is the renormalised one, and on the first 128 rows the token IS the oracle's
    arg-max of logit / T + Gumbel over the kept set (near-ties of the perturbed scores skipped).  ilql: logits = pi + beta min(q1, q2)."""
    from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine, SampleParams, init_hf_style_state_dict
    from oracle import gpt2 as O
    B, d, top_k, temp, beta = 1024, 768, 40, 0.9, 2.0
    cfg = GPT2Config(1, 12, d, 3072, 50257, 32)
    V = cfg.vocab
    g = torch.Generator().manual_seed(77)
    sd = init_hf_style_state_dict(cfg, seed=9)
    sd["wte.weight"] = (sd["wte.weight"] * 6).to(torch.bfloat16).float()
    eng = GPT2Engine(cfg, sd, dev)
    ses = eng.session(B, 8)
    hid = _bf(torch.randn(B, d, generator=g))
    z = hid.double() @ sd["wte.weight"].double().t()                       # [B, V] float64 (bf16 operands: exact products)
    q1 = q2 = None
    if ilql:
        qh = [_bf(torch.relu(torch.randn(B, d, generator=g))) for _ in range(2)]
        qw = [_bf(torch.randn(cfg.vocab_padded, d, generator=g) * 0.05) for _ in range(2)]
        qb = [torch.randn(cfg.vocab_padded, generator=g) * 0.5 for _ in range(2)]
        qs = [qh[i].double() @ qw[i].double().t()[:, :V] + qb[i].double()[None, :V] for i in range(2)]
        z = z + beta * torch.minimum(qs[0], qs[1])
        q1, q2 = tuple(x.to(dev) for x in (qh[0], qw[0], qb[0])), tuple(x.to(dev) for x in (qh[1], qw[1], qb[1]))
    lo = torch.zeros(B, cfg.vocab_padded, device=dev)
    seed, step = 0xA5A5, 11
    sp = SampleParams(temp, top_k, seed, step, 0.0, beta if ilql else 0.0, 3, None, 0.0, 0)
    tok, lp = ses.sample(sp, hidden=hid.to(dev), logits_out=lo, q1=q1, q2=q2)
    torch.cuda.synchronize()
    tok, lp = tok.cpu().numpy(), lp.cpu().numpy()
    from oracle import warpers as OW
    zt, kept = OW.warp(z.numpy(), temp, top_k, 0.0)                          # (pinned to the HF processors: tests/test_oracle_warpers.py)
    srt = -np.sort(-zt, axis=1)[:, :top_k + 1]
    clear = (srt[:, top_k - 1] - srt[:, top_k]) > 1e-3                       # the top-k boundary is decided beyond the fp32-accumulation noise (~3e-5 here)
    assert clear.mean() > 0.9
    assert kept[np.arange(B), tok][clear].all()
    logp = OW.log_probs(zt, kept)
    np.testing.assert_allclose(lp[clear], logp[np.arange(B), tok][clear], rtol=0, atol=3e-3)
    n = 128
    noise = O.gumbel_noise(n, V, seed, step).astype(np.float64)
    score = np.where(kept[:n], zt[:n] + noise, -np.inf)
    best = score.argmax(1)
    top2 = -np.sort(-score, axis=1)[:, :2]
    sure = clear[:n] & ((top2[:, 0] - top2[:, 1]) > 1e-3)
    assert sure.mean() > 0.85 and (tok[:n][sure] == best[sure]).all()
    assert len(set(tok.tolist())) > 200                                     # a real spread of draws, not one dominant column


def test_fused_top_p_at_bench_size_against_the_float64_oracle(dev):
    """Top-p WITHOUT top-k on the candidate path (tile masses + nucleus among the candidates) at 1024 rows x the GPT-2 vocabulary, d = 768, against
    the warper SEMANTICS on float64 logits (HF `TopPLogitsWarper` / `FlaxTopPLogitsWarper`, train_ppo_gpt2.py:98-99, 218-227: the descending-sorted
    prefix whose cumulative probability reaches top_p) and the documented noise stream (oracle/gpt2.py::gumbel_noise) — not against another HIP path:
    every sampled token lies in the oracle's nucleus (rows whose cumulative mass passes top_p within 1e-3 of a token boundary are skipped: bf16-product
    rounding), its log-probability is the renormalised one, on the first 128 rows the token IS the oracle's arg-max of logit / T + Gumbel over the
    nucleus; and the candidate path did the work (no row handed back to materialised logits on this peaked, trained-policy-like distribution)."""
    from lmrl_gym_amd import _lib
    from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine, SampleParams, init_hf_style_state_dict
    from oracle import gpt2 as O
    B, d, top_p, temp = 1024, 768, 0.9, 0.8
    cfg = GPT2Config(1, 12, d, 3072, 50257, 32)
    V = cfg.vocab
    g = torch.Generator().manual_seed(78)
    sd = init_hf_style_state_dict(cfg, seed=9)
    sd["wte.weight"] = (sd["wte.weight"] * 14).to(torch.bfloat16).float()
    eng = GPT2Engine(cfg, sd, dev)
    ses = eng.session(B, 8)
    hid = _bf(torch.randn(B, d, generator=g))
    z = hid.double() @ sd["wte.weight"].double().t()
    lo = torch.full((B, cfg.vocab_padded), float("nan"), device=dev)
    seed, step = 0xA5A6, 13
    tok, lp = ses.sample(SampleParams(temp, 0, seed, step, 0.0, 0.0, 3, None, top_p, 0), hidden=hid.to(dev), logits_out=lo)
    torch.cuda.synchronize()
    tok, lp = tok.cpu().numpy(), lp.cpu().numpy()
    fb = ses.sample_ws[_lib.lib().lmrl_sample_fb_offset(B, cfg.vocab_padded):][:64].view(torch.int32)
    assert int(fb[0].item()) <= B // 50 and int(torch.isnan(lo[:, 0]).sum().item()) >= B - 128 * int(fb[0].item())
    from oracle import warpers as OW
    ztn, kept = OW.warp(z.numpy(), temp, 0, top_p)                           # (pinned to the HF processors: tests/test_oracle_warpers.py)
    n_keep = torch.from_numpy(kept.sum(1))
    assert 2.0 < n_keep.double().mean().item() < 200.0, n_keep.double().mean().item()
    ps = torch.softmax(torch.from_numpy(ztn), 1).sort(1, descending=True).values
    cum = ps.cumsum(1)
    before = torch.where(n_keep > 1, cum.gather(1, (n_keep - 2).clamp(min=0)[:, None])[:, 0], torch.zeros(B, dtype=torch.float64))
    at = cum.gather(1, (n_keep - 1)[:, None])[:, 0]
    clear = (((top_p - before) > 1e-3) & ((at - top_p) > 1e-3)).numpy()       # the crossing is decided beyond the bf16-product / fp32-accumulation noise
    assert clear.mean() > 0.9
    assert kept[np.arange(B), tok][clear].all()
    logp = OW.log_probs(ztn, kept)
    np.testing.assert_allclose(lp[clear], logp[np.arange(B), tok][clear], rtol=0, atol=3e-3)
    n = 128
    noise = O.gumbel_noise(n, V, seed, step).astype(np.float64)
    score = np.where(kept[:n], ztn[:n] + noise, -np.inf)
    best = score.argmax(1)
    top2 = -np.sort(-score, axis=1)[:, :2]
    sure = clear[:n] & ((top2[:, 0] - top2[:, 1]) > 1e-3)
    assert sure.mean() > 0.8 and (tok[:n][sure] == best[sure]).all()
    assert len(set(tok.tolist())) > 100


def test_sampler_steer_and_ilql_perturbation(dev):
    """logits = pi + beta * min(q1, q2)  (value_rl_base/gpt2/generation.py:112-117), greedy."""
    from lmrl_gym_amd.gpt2 import SampleParams
    B, d = 130, 128
    cfg, eng, ses, hid, logits = _engine_and_hidden(dev, B, vocab=2000, d=d)
    g = torch.Generator().manual_seed(8)
    qh = [_bf(torch.relu(torch.randn(B, d, generator=g))) for _ in range(2)]
    qw = [_bf(torch.randn(cfg.vocab_padded, d, generator=g) * 0.3) for _ in range(2)]
    qb = [torch.randn(cfg.vocab_padded, generator=g) for _ in range(2)]
    q = [qh[i].double() @ qw[i].double().t() + qb[i].double() for i in range(2)]
    beta = 4.0
    full = logits + beta * torch.minimum(q[0], q[1])[:, : cfg.vocab]
    lo = torch.zeros(B, cfg.vocab_padded, device=dev)
    ops = [tuple(x.to(dev) for x in (qh[i], qw[i], qb[i])) for i in range(2)]
    tok, lp = ses.sample(SampleParams(0.0, 0, 1, 0, 0.0, beta, 0), hidden=hid, logits_out=lo, q1=ops[0], q2=ops[1])
    torch.testing.assert_close(lo.cpu()[:, : cfg.vocab].double(), full, rtol=2e-3, atol=2e-2)
    top2 = full.topk(2).values
    safe = (top2[:, 0] - top2[:, 1]) > 5e-2
    assert safe.float().mean() > 0.8 and torch.equal(tok.cpu()[safe].long(), full.argmax(1)[safe])
    # steering adds `strength` to one logit per row
    steer = torch.randint(0, cfg.vocab, (B,), generator=g).to(torch.int32)
    tok, _ = ses.sample(SampleParams(1.0, 0, 5, 1, 1000.0, 0.0, 0), hidden=hid, steer_tok=steer.to(dev))
    assert torch.equal(tok.cpu(), steer)


def test_ragged_prefill_is_bit_identical(dev):
    """Chunk forwards on the compacted rows (sum of cnt) vs on all B*C slots: same last hidden states, same KV cache, same
    cache lengths, bit for bit, for ragged counts including empty envs; decode steps afterwards agree too."""
    from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine, init_hf_style_state_dict
    cfg = GPT2Config(2, 12, 768, 3072, 1000, 64)
    eng = GPT2Engine(cfg, init_hf_style_state_dict(cfg, seed=2), dev)
    B = 37
    g = torch.Generator().manual_seed(1)
    plan = [(8, torch.randint(0, 9, (B,), generator=g)), (16, torch.randint(0, 17, (B,), generator=g)), (1, torch.randint(0, 2, (B,), generator=g)),
            (1, torch.ones(B, dtype=torch.int64)), (8, torch.randint(3, 9, (B,), generator=g))]
    plan[0][1][0] = 0; plan[0][1][B - 1] = 8
    outs = []
    from lmrl_gym_amd.gpt2 import FWD_ATTN_ITEMS2, FWD_ATTN_ITEMS3, FWD_FULL_LAST_LAYER, FWD_KV_FROM_GEMM, FWD_RAGGED_ALWAYS, FWD_RAGGED_NEVER
    # the sessions are INTERLEAVED forward by forward: the variant is a property of the call, not of the process.  FWD_KV_FROM_GEMM: the
    # decode qkv GEMM's epilogue appends the new K / V rows instead of the attention kernel — same cache bytes, same outputs
    sess = [eng.session(B, 48, flags=FWD_RAGGED_ALWAYS), eng.session(B, 48, flags=FWD_RAGGED_NEVER),
            eng.session(B, 48, flags=FWD_RAGGED_ALWAYS | FWD_KV_FROM_GEMM), eng.session(B, 48, flags=FWD_RAGGED_NEVER | FWD_KV_FROM_GEMM),
            # FWD_FULL_LAST_LAYER: the last layer's projection + MLP on every chunk row instead of on each env's last new token only (the default)
            eng.session(B, 48, flags=FWD_RAGGED_ALWAYS | FWD_FULL_LAST_LAYER), eng.session(B, 48, flags=FWD_RAGGED_NEVER | FWD_FULL_LAST_LAYER),
            # FWD_ATTN_ITEMS2 / 3 (round 6, A/B): decode attention on a grid of half / a third as many waves, 2 / 3 (env, head) items per wave
            eng.session(B, 48, flags=FWD_ATTN_ITEMS2), eng.session(B, 48, flags=FWD_RAGGED_ALWAYS | FWD_ATTN_ITEMS3)]
    hs = [[] for _ in sess]
    for C, cnt in plan:
        toks = torch.randint(0, cfg.vocab, (B * C,), generator=torch.Generator().manual_seed(C)).to(torch.int32).to(dev)
        for i, ses in enumerate(sess):
            hs[i].append(ses.forward(toks, cnt.to(torch.int32).to(dev), C).clone())
    outs = [(hs[i], sess[i].kv.clone(), sess[i].len.clone()) for i in range(len(sess))]
    for o in outs[1:]:
        for a, b in zip(outs[0][0], o[0]):
            assert torch.equal(a, b)
        assert torch.equal(outs[0][1], o[1]) and torch.equal(outs[0][2], o[2])


@pytest.mark.parametrize("B,model", [(8, "small"), (16, "small"), (3, "small"), (8, "medium")])
def test_skinny_decode_products_equal_the_tile_kernels(dev, B, model):
    """LMRL_FWD_SKINNY (round 6; configs[0]'s 8-env batch): decode forwards of <= 16 sequences run their four Dense products per layer on
    csrc/skinny_gemm.h (one MFMA row block, K split over the 16 waves of a workgroup) instead of the 64 x 64-tile kernels.  Same formulas, K summed
    in 8 slices: hidden states and K/V rows agree with the default session to bf16 rounding over a prefill + 12 decode steps (ragged counts,
    finished rows, both row-compaction modes), at GPT-2-small and GPT-2-medium widths."""
    from lmrl_gym_amd.gpt2 import FWD_RAGGED_ALWAYS, FWD_SKINNY, GPT2Config, GPT2Engine, init_hf_style_state_dict
    cfg = GPT2Config(3, 12, 768, 3072, 1000, 64) if model == "small" else GPT2Config(2, 16, 1024, 4096, 1000, 64)
    eng = GPT2Engine(cfg, init_hf_style_state_dict(cfg, seed=6), dev)
    g = torch.Generator().manual_seed(B)
    sess = [eng.session(B, 48), eng.session(B, 48, flags=FWD_SKINNY), eng.session(B, 48, flags=FWD_SKINNY | FWD_RAGGED_ALWAYS)]
    toks = torch.randint(0, cfg.vocab, (B * 16,), generator=g).to(torch.int32).to(dev)
    cnt0 = torch.randint(5, 17, (B,), generator=g).to(torch.int32).to(dev)
    for ses in sess:
        ses.reset()
        ses.forward(toks, cnt0, 16)
    assert torch.equal(sess[0].last_hidden, sess[1].last_hidden)                    # (chunk forwards are not affected by the flag)
    worst = 0.0
    for step in range(12):
        t1 = torch.randint(0, cfg.vocab, (B,), generator=g).to(torch.int32).to(dev)
        cnt = (torch.rand(B, generator=g) < 0.85).to(torch.int32)
        if step == 0:
            cnt[:] = 1
        cnt = cnt.to(dev)
        hs = [ses.forward(t1, cnt, 1).float().clone() for ses in sess]
        live = cnt.bool()
        for h in hs[1:]:
            d = (h[live] - hs[0][live]).abs().max().item()
            worst = max(worst, d)
            assert d <= 0.06 * max(1.0, hs[0][live].abs().max().item()), (step, d)
        assert torch.equal(hs[1], hs[2])                                              # compacted rows: the same per-row arithmetic
        assert all(torch.equal(ses.len, sess[0].len) for ses in sess)
    assert worst > 0.0                                                                # (a different association order: not the same kernels)
    kv = [ses.kv.view(torch.bfloat16).float() for ses in sess]
    assert (kv[1] - kv[0]).abs().max().item() <= 0.06 * kv[0].abs().max().item()


@pytest.mark.parametrize("B,group", [(37, True), (37, False), (256, True)])
def test_indexed_prefix_attention_equals_copied_prefix(dev, B, group):
    """`attach_prefix_from` (prompt rows READ from the prefix session by the decode attention, lmrl_gpt2_forward_prefixed) vs
    `gather_prefix_from` (rows copied per env): same hidden states after every decode step and the same generated-token rows in the
    env caches, bit for bit — ragged prompt lengths, envs without a prompt (idx < 0), finished envs (cnt = 0), both launch orders
    (B = 256 with 12 heads: a grid the XCD-grouped order applies to)."""
    from lmrl_gym_amd.gpt2 import FWD_KV_FROM_GEMM, GPT2Config, GPT2Engine, init_hf_style_state_dict
    cfg = GPT2Config(2, 12, 768, 3072, 1000, 64)
    eng = GPT2Engine(cfg, init_hf_style_state_dict(cfg, seed=4), dev)
    R, cap, tmax = 9, 32, 40
    g = torch.Generator().manual_seed(7)
    src = eng.session(R, cap)
    src.reset()
    plens = torch.randint(1, cap + 1, (R,), generator=g)
    plens[0], plens[1] = cap, 1
    for c0 in range(0, cap, 16):
        toks = torch.randint(0, cfg.vocab, (R * 16,), generator=g).to(torch.int32).to(dev)
        src.forward(toks, torch.clamp(plens - c0, 0, 16).to(torch.int32).to(dev), 16)
    idx = torch.randint(0, R, (B,), generator=g).to(torch.int32)
    idx[3] = -1
    idx = idx.to(dev)
    for flags in (0, FWD_KV_FROM_GEMM):
        a, b = eng.session(B, tmax, flags=flags), eng.session(B, tmax, flags=flags)
        a.reset(); b.reset()
        a.gather_prefix_from(src, idx, cap)
        b.attach_prefix_from(src, idx, cap, group=group)
        assert torch.equal(a.len, b.len) and torch.equal(a.last_hidden, b.last_hidden)
        if group:
            order = b._pfx_order.cpu().numpy()
            assert sorted(order.tolist()) == list(range(B))
            keys = np.where(idx.cpu().numpy() >= 0, idx.cpu().numpy(), R)[order]
            assert (np.diff(keys) >= 0).all()                     # grouped by prefix row, envs without one last
        for step in range(tmax - cap):
            toks = torch.randint(0, cfg.vocab, (B,), generator=g).to(torch.int32).to(dev)
            cnt = (torch.rand(B, generator=g) < 0.9).to(torch.int32).to(dev)
            ha = a.forward(toks, cnt, 1).clone()
            hb = b.forward(toks, cnt, 1).clone()
            assert torch.equal(ha, hb), step
            assert torch.equal(a.len, b.len)
        # generated rows: position t >= prompt length of env i, layer l, K|V — equal in both caches (the prompt rows exist only in `a`)
        kva = a.kv.view(torch.int16).view(2 * cfg.n_layer, B, tmax, cfg.d_model)
        kvb = b.kv.view(torch.int16).view(2 * cfg.n_layer, B, tmax, cfg.d_model)
        n0 = torch.where(idx >= 0, src.len[idx.clamp(min=0).long()], torch.zeros_like(idx))
        t_idx = torch.arange(tmax, device=dev)[None, :]
        own = (t_idx >= n0[:, None]) & (t_idx < a.len[:, None])
        assert own.any()
        assert torch.equal(kva[:, own], kvb[:, own])
        with pytest.raises(Exception):
            b.forward(torch.zeros(B * 8, dtype=torch.int32, device=dev), torch.ones(B, dtype=torch.int32, device=dev), 8)
