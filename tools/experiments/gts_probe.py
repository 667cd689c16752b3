"""Root-cause probe for the GraphedTrainStep wrong-loss report: model A captured, then some OTHER
activity in the process, then A's replays -- compared with A captured and replayed alone.  One mode per process:

    python tools/experiments/gts_probe.py alone|repro|poison|b_stack|b_torch|b_fwd|b_fwd_grad|b_noopt|b_after [--steps 4]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from codenet_amd import harness, pipeline  # noqa: E402


def build():
    torch.manual_seed(0)
    m = harness.create_model(quantize=True).cuda().train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eval()
    return m, torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1.25e-4, capturable=True)


def loss_fn(net, x):
    out = net(x)[-1]
    return sum(v.square().mean() for v in out.values())


def main():
    mode = sys.argv[1]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 4
    x = torch.randn(8, 3, 256, 256, generator=torch.Generator().manual_seed(1)).cuda()
    net_b = opt_b = None
    if mode in ("repro", "b_fwd", "b_fwd_grad", "b_noopt"):
        net_b, opt_b = build()
    elif mode == "b_stack":
        torch.manual_seed(3)
        net_b = pipeline.build_hot_path(quantized=True).cuda().train()
        opt_b = torch.optim.Adam(net_b.parameters(), lr=1e-4, capturable=True)
    elif mode == "b_torch":
        torch.manual_seed(3)
        net_b = torch.nn.Sequential(torch.nn.Conv2d(3, 24, 3, 2, 1), torch.nn.BatchNorm2d(24), torch.nn.ReLU(),
                                    torch.nn.Conv2d(24, 58, 3, 2, 1, groups=1), torch.nn.BatchNorm2d(58), torch.nn.ReLU(),
                                    torch.nn.Conv2d(58, 58, 3, 1, 1, groups=58), torch.nn.Conv2d(58, 116, 1)).cuda().train()
        opt_b = torch.optim.Adam(net_b.parameters(), lr=1e-4, capturable=True)
    if mode.startswith("det_"):
        torch.use_deterministic_algorithms(True, warn_only=True)
        mode = mode if mode == "det_alone" else mode[4:]
    net_a, opt_a = build()
    step = pipeline.GraphedTrainStep(net_a, opt_a, loss_fn, (x,), warmup=3, unvalidated=True)
    if mode == "b_after":
        net_b, opt_b = build()

    def b_step():
        if mode in ("repro", "b_after"):
            opt_b.zero_grad(set_to_none=True); l = loss_fn(net_b, x); l.backward(); opt_b.step()
        elif mode == "b_noopt":
            for p in net_b.parameters():
                p.grad = None
            loss_fn(net_b, x).backward()
        elif mode == "b_fwd":
            with torch.no_grad():
                loss_fn(net_b, x)
        elif mode == "b_fwd_grad":
            loss_fn(net_b, x)
        elif mode == "b_stack":
            xs = torch.randn(4, 1024, 8, 8, device="cuda").abs_()
            opt_b.zero_grad(set_to_none=True); net_b(xs).square().mean().backward(); opt_b.step()
        elif mode == "b_torch":
            opt_b.zero_grad(set_to_none=True); net_b(x).square().mean().backward(); opt_b.step()
        elif mode == "poison":
            # every cached-free block of the normal pool, and 4 GB of fresh memory, filled with NaN and released again
            free = torch.cuda.memory_reserved() - torch.cuda.memory_allocated()
            blocks = []
            for sz in (1 << 30, 1 << 26, 1 << 22, 1 << 18, 1 << 14, 1 << 10):
                for _ in range(64):
                    if sum(b.numel() * 4 for b in blocks) > free + (4 << 30):
                        break
                    blocks.append(torch.full((sz // 4,), float("nan"), device="cuda"))
            torch.cuda.synchronize()
            del blocks

    import json
    tiny = torch.nn.Parameter(torch.zeros(4, device="cuda"))
    opt_t = torch.optim.Adam([tiny], lr=1e-3, capturable=True)
    sums = []
    out = []
    for it in range(steps):
        if mode == "b_adam_tiny":
            tiny.grad = torch.ones_like(tiny)
            opt_t.step()
        elif mode == "b_sgd_tiny":
            tiny.data.add_(1.0)
        elif mode == "sleep":
            import time
            time.sleep(0.5)
        elif mode not in ("alone", "det_alone"):
            b_step()
        l = step(x)
        torch.cuda.synchronize()
        out.append("%.6f" % l.item())
        rec = {}
        for n, p_ in net_a.named_parameters():
            rec["p:" + n] = p_.detach().double().sum().item()
            if p_.grad is not None:
                rec["g:" + n] = p_.grad.detach().double().sum().item()
        for n, b_ in net_a.named_buffers():
            rec["b:" + n] = b_.detach().double().sum().item()
        sums.append(rec)
    print("GTS", mode, " ".join(out), flush=True)
    os.makedirs("gpurun_out/r6/gts", exist_ok=True)
    json.dump(sums, open("gpurun_out/r6/gts/%s.json" % mode, "w"))


main()
