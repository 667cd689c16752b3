import torch, sys
sys.path.insert(0, ".")
from codenet_amd import harness, pipeline
res, batch = int(sys.argv[1]), int(sys.argv[2])
model = harness.create_model(quantize=True).cuda()
fb = pipeline.FusedBackbone(model)
g = torch.Generator().manual_seed(res + batch)
xs = [torch.randn(batch, 3, res, res, generator=g).cuda() for i in range(3)]
for _ in range(20):
    for x in xs: fb(x)
pipeline.set_running_stat(model, False)
pipeline.cover_frozen_ranges(model, xs, margin=0.05)
fz = pipeline.FrozenBackbone(model)
x = xs[0]
got, gq, hw = fz(x)
print("overflow", fz.overflowed())
B = fz._bufs
m = model
with torch.no_grad():
    ref0 = m.layer0(x)
    act0 = m.layer0[1][1]
    st = act0._device_state(x.device).view(torch.float32)
    want0 = torch.round(st[2] * ref0 - st[3]).permute(0, 2, 3, 1).reshape(batch, -1, 24)
    print("stem diff", (B["x0"][:, :, :24].float() - want0).abs().max().item())
    H = W = ref0.shape[2]
    r = ref0
    x8, x_ld, x_state, in_logical = B["x0"], 32, act0._device_state(x.device).data_ptr(), None
    for name in ("layer1", "layer2", "layer3"):
        nodes = list(getattr(m, name))
        rr = r
        for k in range(len(nodes)):
            rr = nodes[k](rr)
            B["layers"].pop("dbg", None)
            Y, ldc, qs, logical, Ho, Wo = fz._layer("dbg", nodes[:k + 1], x8, x_ld, x_state, in_logical, batch, H, W)
            torch.cuda.synchronize()
            sh = fz._fb._unit(nodes[0])["sh"]
            st = sh._device_state(x.device).view(torch.float32)
            want = torch.round(st[2] * rr - st[3]).permute(0, 2, 3, 1).reshape(batch * Ho * Wo, -1)
            C = want.shape[1]
            inv = torch.empty(C, dtype=torch.long); inv[torch.tensor(logical)] = torch.arange(C)
            gotL = Y[:, :C].float()[:, inv.cuda()]
            d = (gotL - want).abs()
            print(name, "unit", k, "diff frac", (d > 0).float().mean().item(), "max", d.max().item(), "overflow", fz.overflowed())
            if d.max().item() > 3:
                bad = (d > 3).any(0).nonzero().flatten().tolist()
                print("   bad logical channels", bad[:40], len(bad))
                badr = (d > 3).any(1).nonzero().flatten().tolist()
                print("   bad rows", badr[:40], len(badr), "of", d.shape[0])
        x8, x_ld, x_state, in_logical, H, W = fz._layer(name, nodes, x8, x_ld, x_state, in_logical, batch, H, W)
        r = rr

with torch.no_grad():
    r4 = m.layer4(r)
    st4 = m.layer4[1][1]._device_state(x.device).view(torch.float32)
    wantM = torch.round(st4[2] * r4 - st4[3]).permute(0, 2, 3, 1).reshape(batch, -1, r4.shape[1])
    feat, fq, hw2 = fb(x)
    wantF = torch.round(st4[2] * feat - st4[3])
    for nm, a, b in (("frozen-module", got.float(), wantM), ("frozen-fused", got.float(), wantF), ("fused-module", wantF, wantM)):
        d = (a - b).abs()
        print(nm, "frac", (d > 0).float().mean().item(), "max", d.max().item(), "mean", d.mean().item())
