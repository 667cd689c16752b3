"""Diagnostic: ticket protocol (a) vs deferred range commit (b) replayed as HIP graphs many times; on a mismatch of a
tracked range print which QuantAct, both values and the true batch extreme of the stage output tensor."""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from codenet_amd import pipeline

dev = torch.device("cuda:0")
planes, res, n = [256, 64, 32, 16], 8, 2
net_a = pipeline.build_hot_path(quantized=True, planes=planes).to(dev).eval()
net_b = copy.deepcopy(net_a)
a, b = pipeline.FusedHotPath(net_a.deconv_layers), pipeline.FusedHotPath(net_b.deconv_layers)
a.deferred, b.deferred = False, True
x = torch.randn(n, planes[0], res, res, device=dev).abs() * 2
ra, rb = a.capture(x, unpack=False), b.capture(x, unpack=False)
acts = lambda net: [m for m in net.modules() if hasattr(m, "x_min") and isinstance(m.x_min, torch.Tensor)]
A, Bq = acts(net_a), acts(net_b)
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    x.mul_(1.001)
    prev = [(m.x_min.item(), m.x_max.item()) for m in A]
    ra()
    rb()
    torch.cuda.synchronize()
    for i, (ma, mb) in enumerate(zip(A, Bq)):
        va, vb = (ma.x_min.item(), ma.x_max.item()), (mb.x_min.item(), mb.x_max.item())
        if va != vb:
            bad += 1
            sa = ma._device_state(dev).view(torch.float32)[4:6].tolist()
            sb = mb._device_state(dev).view(torch.float32)[4:6].tolist()
            true = None
            if i % 3 == 2:
                r = a._bufs["stages"][i // 3]["r"]
                true = (r.min().item(), r.max().item())
            print("iter %d act %d: ticket %r deferred %r | batch extremes ticket %r deferred %r | true(r) %r | prev %r"
                  % (it, i, va, vb, sa, sb, true, prev[i]))
            mb.x_min.copy_(ma.x_min)
            mb.x_max.copy_(ma.x_max)
print("mismatches:", bad)
