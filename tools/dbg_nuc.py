import sys, numpy as np, torch
sys.path.insert(0, '.')
import lmrl_gym_amd
from lmrl_gym_amd import _lib
from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine, SampleParams, init_hf_style_state_dict
dev = torch.device('cuda:0')
L = _lib.lib()
B, vocab, d = 1024, 50257, 128
cfg = GPT2Config(1, d // 64, d, 256, vocab, 32)
g = torch.Generator().manual_seed(vocab)
hid = torch.randn(B, d, generator=g).to(torch.bfloat16).float().to(dev)
sd = init_hf_style_state_dict(cfg, seed=3)
sd["wte.weight"] = (sd["wte.weight"] * 60).to(torch.bfloat16).float()
eng = GPT2Engine(cfg, sd, dev)
ses = eng.session(B, 8)
lo = torch.zeros(B, cfg.vocab_padded, device=dev)
L.lmrl_sampler_set_variant(2)
sp = SampleParams(1.0, 0, 0xD1CE, 4, 0.0, 0.0, 9, None, 0.9, 0)
tok2, _ = ses.sample(sp, hidden=hid, logits_out=lo)
torch.cuda.synchronize()
z = lo[:, :vocab].double().cpu()
L.lmrl_sampler_set_variant(0)
lo2 = torch.full_like(lo, float('nan'))
tok0, _ = ses.sample(sp, hidden=hid, logits_out=lo2)
torch.cuda.synchronize()
fb_off = L.lmrl_sample_fb_offset(B, cfg.vocab_padded)
fb = ses.sample_ws[fb_off:fb_off + 4 * (16 + 64 + B)].view(torch.int32).cpu().numpy()
n_fb = fb[0]; rows = np.sort(fb[80:80 + n_fb])
print('handed', n_fb, rows[:40])
tiles = cfg.vocab_padded // 128
zp = torch.full((B, cfg.vocab_padded), -1e30, dtype=torch.float64); zp[:, :vocab] = z
zt = zp.view(B, tiles, 128)
srt = zt.sort(dim=2, descending=True).values
hidden = srt[:, :, 7].max(1).values
p = torch.softmax(z, 1)
ps, order = p.sort(1, descending=True)
cum = ps.cumsum(1)
nuc = (cum < 0.9).sum(1) + 1
kstar = torch.gather(z, 1, order.gather(1, (nuc - 1)[:, None]))[:, 0]
flag = np.zeros(B, bool); flag[rows] = True
print('nucleus size  flagged:', nuc[flag].float().mean().item(), nuc[flag].max().item(), ' others:', nuc[~flag].float().mean().item(), nuc[~flag].max().item())
print('K* - hidden  flagged min/max:', (kstar - hidden)[flag].min().item(), (kstar - hidden)[flag].max().item(), ' others min:', (kstar - hidden)[~flag].min().item())
print('expected hand-backs (K* <= hidden):', int((kstar <= hidden).sum()))
tm_off = (fb_off + 4 * (16 + 64 + B) + 7) & ~7
tm = ses.sample_ws[tm_off:tm_off + 8 * B * tiles].view(torch.int64).cpu().view(B, tiles).double()
tmax = srt[:, :, 0]
exp_s = (torch.exp(zt - tmax[:, :, None]) ).sum(2) * 2**32
rel = ((tm - exp_s).abs() / exp_s)
print('tile mass rel err max', rel.max().item(), 'rows of worst', rel.max(1).values.topk(5))
