import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from codenet_amd import _native as N_, ops
g = torch.Generator().manual_seed(1280)
Co, K = 256, 1024
w = (torch.randn(Co, K, 1, 1, generator=g) * torch.rand(Co, 1, 1, 1, generator=g) * 3).cuda()
gamma = (torch.rand(Co, generator=g) + 0.5).cuda(); beta = (torch.randn(Co, generator=g) * 0.1).cuda()
mean = (torch.randn(Co, generator=g) * 0.1).cuda(); var = (torch.rand(Co, generator=g) + 0.5).cuda()
eps = 1e-5
wq = torch.empty_like(w); sf = torch.empty(Co, device="cuda"); b = torch.empty(Co, device="cuda")
lib = N_.lib()
p = lambda t: t.data_ptr()
rc = lib.cdn_codenet_weight_prep(p(w), Co, K, p(gamma), p(beta), p(mean), p(var), eps, None, 4, p(wq), p(sf), p(b), ops._stream(w))
torch.cuda.synchronize()
std = torch.sqrt(var + eps)
sf_t = gamma / std
print("sf equal", torch.equal(sf, sf_t), (sf - sf_t).abs().max().item())
wf = w * sf_t.reshape(Co, 1, 1, 1)
b_t = (torch.zeros_like(mean) - mean) * sf_t + beta
print("b equal", torch.equal(b, b_t))
w2 = wf.view(Co, -1)
mn, mx = w2.min(1).values, w2.max(1).values
mag = torch.max(torch.stack([mn.abs(), mx.abs()], 1), 1).values
scale = 7 / torch.clamp(mag, min=1e-10)
q = torch.clamp(torch.round(scale.view(-1, 1, 1, 1) * wf - torch.zeros_like(scale).view(-1,1,1,1)), -8, 7)
out = (q + 0) / scale.view(-1, 1, 1, 1)
print("wq equal", torch.equal(out, wq), (out - wq).abs().max().item(), ((out != wq).sum().item()))
bad = (out != wq).nonzero()
if len(bad):
    i = bad[0]; co = i[0].item(); k = i[1].item()
    print(co, k, out[co, k].item(), wq[co, k].item(), w[co, k].item(), sf_t[co].item(), wf[co, k].item(), scale[co].item(), (scale[co] * wf[co, k]).item())
