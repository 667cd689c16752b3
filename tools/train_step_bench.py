"""Config-e context (BASELINE.json configs[4]): forward + backward of the W4A8 deform stages under
autograd (QAT step of quant_main.py restricted to the hot path): gather forward/backward on the HIP
kernels, 1x1 convolutions and fake-quant (straight-through) on PyTorch-ROCm.  GPU only."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from codenet_amd import pipeline


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)     # lib/opts.py:91 default batch
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    net = pipeline.build_hot_path(quantized=not a.fp32).cuda().train()
    for m in net.modules():                      # BN inside QuantBnConv2d is never called; plain BN in fp32
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    x = pipeline.make_input(a.batch, a.res, device="cuda").requires_grad_(True)
    opt = torch.optim.Adam(net.parameters(), lr=1.25e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        y = net(x)
        loss = y.square().mean()
        loss.backward()
        opt.step()
        return loss

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"config": "CoDeNet1x %dx%d %s QAT step over deconv_layers, batch %d" % (
        a.res, a.res, "fp32" if a.fp32 else "W4A8", a.batch), "ms_per_step": round(dt * 1e3, 3),
        "images_per_s": round(a.batch / dt, 1), "loss": float(loss)}))


if __name__ == "__main__":
    main()
