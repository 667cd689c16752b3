"""GraphedTrainStep probe 2: is a REPLAY a deterministic function of the state, and does other work in the process change
it?  One process: capture A; snapshot S0 (parameters, buffers, optimizer state); then rounds of
[restore S0 -> disturbance -> 2 replays -> bit-level record], printed as matches against round 0.
    python tools/experiments/gts_probe2.py [whole|stack]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from codenet_amd import harness, pipeline  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "whole"


def build():
    torch.manual_seed(0)
    if which == "whole":
        m = harness.create_model(quantize=True).cuda().train()
    else:
        m = pipeline.build_hot_path(quantized=True).cuda().train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eval()
    return m, torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1.25e-4, capturable=True)


def loss_fn(net, x):
    if which == "whole":
        out = net(x)[-1]
        if os.environ.get("GTS_TWO_LEVEL") == "1":      # no multi-block (semaphore) reduction: rows of 64, then one block
            terms = [v.square().reshape(-1, 64).sum(1).sum() / v.numel() for v in out.values()]
        else:
            terms = [v.square().mean() for v in out.values()]
        if net is net_a_holder[0] and os.environ.get("GTS_TERMS") == "1":
            TERMS[:] = terms
        return sum(terms)
    return net(x).square().mean()


TERMS = []
net_a_holder = [None]


x = (torch.randn(8, 3, 256, 256, generator=torch.Generator().manual_seed(1)).cuda() if which == "whole" else
     (torch.randn(4, 1024, 8, 8, generator=torch.Generator().manual_seed(1)).abs_() * 1.66).cuda())
net_b, opt_b = build()
torch.manual_seed(3)
net_t = torch.nn.Sequential(torch.nn.Conv2d(3, 24, 3, 2, 1), torch.nn.BatchNorm2d(24), torch.nn.ReLU(),
                            torch.nn.Conv2d(24, 58, 3, 2, 1), torch.nn.BatchNorm2d(58), torch.nn.ReLU(),
                            torch.nn.Conv2d(58, 58, 3, 1, 1, groups=58), torch.nn.Conv2d(58, 116, 1)).cuda().train()
opt_t = torch.optim.Adam(net_t.parameters(), lr=1e-4, capturable=True)
xt = torch.randn(8, 3, 256, 256, device="cuda")
net_a, opt_a = build()
net_a_holder[0] = net_a
step = pipeline.GraphedTrainStep(net_a, opt_a, loss_fn, (x,), warmup=3, unvalidated=True)
torch.cuda.synchronize()


def state_tensors():
    ts = [p for p in net_a.parameters()] + [b for b in net_a.buffers()]
    for st in opt_a.state.values():
        ts += [v for v in st.values() if torch.is_tensor(v)]
    for m in net_a.modules():                       # QuantAct device state words (scale / zero-point scratch)
        s_ = getattr(m, "_state", None)
        if torch.is_tensor(s_):
            ts.append(s_)
    return ts


S0 = [t.detach().clone() for t in state_tensors()]


def restore():
    with torch.no_grad():
        for t, s in zip(state_tensors(), S0):
            t.copy_(s)
    torch.cuda.synchronize()


def record():
    out = []
    for _ in range(2):
        l = step(x)
        torch.cuda.synchronize()
        if TERMS:
            print("      loss %r terms %r" % (l.item(), [t.item() for t in TERMS]))
        out.append(l.detach().clone())
        out += [p.detach().clone() for p in net_a.parameters()]
        out += [p.grad.detach().clone() for p in net_a.parameters() if p.grad is not None]
    return out


def b_whole():
    opt_b.zero_grad(set_to_none=True); loss_fn(net_b, x).backward(); opt_b.step()


def b_torch():
    opt_t.zero_grad(set_to_none=True); net_t(xt).square().mean().backward(); opt_t.step()


def b_sync():
    torch.cuda.synchronize()


def b_alloc():
    t = [torch.full((1 << 24,), float("nan"), device="cuda") for _ in range(16)]
    torch.cuda.synchronize()
    del t


rounds = [("none", b_sync), ("none", b_sync), ("torch model step", b_torch), ("none", b_sync), ("second model first step", b_whole),
          ("second model step", b_whole), ("alloc+nan", b_alloc), ("none", b_sync)]
base = None
names = ["loss"] + ["p:" + n for n, _ in net_a.named_parameters()]
for i, (what, fn) in enumerate(rounds):
    restore()
    fn()
    rec = record()
    if base is None:
        base = rec
        print("round 0 (%s): losses %r %r" % (what, rec[0].item(), rec[len(rec) // 2].item()))
        continue
    bad = [j for j, (u, v) in enumerate(zip(base, rec)) if not torch.equal(u, v)]
    big = sum(1 for u, v in zip(base, rec) if ((u - v).abs().max() / (u.abs().max() + 1e-30)).item() > 1e-3)
    print("      tensors off by > 1e-3: %d" % big)
    worst = max([((u - v).abs().max() / (u.abs().max() + 1e-30)).item() for u, v in zip(base, rec)] + [0.0])
    print("GTS2 %s round %d after [%s]: %d of %d tensors differ from round 0, worst rel %.2e, losses %r %r" % (
        which, i, what, len(bad), len(rec), worst, rec[0].item(), rec[len(rec) // 2].item()), flush=True)
