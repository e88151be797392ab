"""Pins oracle/gpt2.py against the installed HF PyTorch GPT-2 (CPU)."""
import pytest
import torch


def test_oracle_gpt2_matches_hf_pytorch():
    transformers = pytest.importorskip("transformers")
    from oracle import gpt2 as O
    cfg = transformers.GPT2Config(vocab_size=211, n_positions=48, n_embd=64, n_layer=3, n_head=4,
                                  resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0)
    torch.manual_seed(0)
    model = transformers.GPT2LMHeadModel(cfg).eval()
    # give LN params / biases non-trivial values
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
    sd = {k[len("transformer."):]: v for k, v in model.state_dict().items() if k.startswith("transformer.") and not k.endswith(".attn.bias") and "masked_bias" not in k}
    ids = torch.randint(0, 211, (3, 17))
    with torch.no_grad():
        ref = model(ids).logits
    got = O.forward(sd, ids, n_head=4, dtype=torch.float32)
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4)
    # padded batch with attention mask + explicit positions (how initialize_attn_mask_pos_ids feeds the model)
    am = torch.ones(3, 17, dtype=torch.long); am[1, 12:] = 0; am[2, 5:] = 0
    pos = (am.cumsum(-1) - 1).clamp(min=0)
    with torch.no_grad():
        ref = model(ids, attention_mask=am, position_ids=pos).logits
    got = O.forward(sd, ids, n_head=4, attention_mask=am, position_ids=pos, dtype=torch.float32)
    m = am.bool()
    torch.testing.assert_close(got[m], ref[m], rtol=1e-4, atol=1e-4)


def test_philox_known_answer():
    # Random123 known-answer test vector for philox4x32-10: counter = key = 0 and all-ones
    import numpy as np
    from oracle.gpt2 import philox4x32_10
    o = philox4x32_10([0], [0], [0], [0], 0, 0)
    assert [int(x[0]) for x in o] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    o = philox4x32_10([0xffffffff], [0xffffffff], [0xffffffff], [0xffffffff], 0xffffffff, 0xffffffff)
    assert [int(x[0]) for x in o] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]


def test_gumbel_noise_is_finite_for_extreme_bits():
    # u must stay strictly inside (0, 1) for every 32-bit draw (fp32 rounding of the +0.5 offset)
    import numpy as np
    for bits in (0, 1, 0xFFFFFFFF, 0xFFFFFE00, 0x80000000):
        u = (np.float32(bits >> 9) + np.float32(0.5)) * np.float32(1.1920928955078125e-07)
        assert 0.0 < float(u) < 1.0
        assert np.isfinite(-np.log(-np.log(u)))
