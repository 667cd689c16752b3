"""Config e (BASELINE.json configs[4]): forward + backward + Adam step of the W4A8 deform stages (the QAT step
of quant_main.py restricted to the hot path).  Per stage ONE native autograd function
(codenet_amd/functions/codenet_stage.py): scale 1x1, QuantAct, gather, QuantAct, pointwise 1x1 forward and
backward on the HIP kernels (f32 MFMA for the 1x1 convolutions' data and weight gradients, straight-through
quantisers); the per-channel weight fake-quantisation / BN fold (tiny tensors), ReLU, Upsample and Adam are torch
ops.  --graph replays the whole step as one HIP graph (a QAT step is ~500 small launches).  GPU only."""
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
    ap.add_argument("--loss", action="store_true", help="surrogate loss on the stage output instead of a fixed upstream gradient")
    ap.add_argument("--graph", action="store_true", help="capture forward + backward + optimizer step in one HIP graph")
    ap.add_argument("--foreach-adam", action="store_true",
                    help="torch.optim.Adam's default multi-tensor form (8 launches per step) instead of fused=True (1)")
    a = ap.parse_args()
    net = pipeline.build_hot_path(quantized=not a.fp32).cuda().train()
    for m in net.modules():                      # BN inside QuantBnConv2d is never called; plain BN in fp32
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    x = pipeline.make_input(a.batch, a.res, device="cuda").requires_grad_(True)
    # lib/opts.py:93 lr; the reference's torch.optim.Adam, in its single-kernel form unless --foreach-adam
    opt = torch.optim.Adam(net.parameters(), lr=1.25e-4, capturable=a.graph, fused=not a.foreach_adam)

    # the stages' output gradient comes from the heads in the real QAT step: a fixed upstream gradient of the
    # output's shape (a loss computed on this 134 MB tensor would add ~160 us of reductions that are not part
    # of the step); --loss keeps the old surrogate loss
    with torch.no_grad():
        go = torch.randn_like(net(x)) * 1e-3

    def step():
        opt.zero_grad(set_to_none=True)
        y = net(x)
        if a.loss:
            loss = y.square().mean()
            loss.backward()
        else:
            loss = y[0, 0, 0, 0].detach()
            y.backward(go)
        opt.step()
        return loss

    if a.graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(g):
            y = net(x)
            if a.loss:
                static_loss = y.square().mean()
                static_loss.backward()
            else:
                static_loss = y[0, 0, 0, 0].detach()
                y.backward(go)
            opt.step()

        def step():   # noqa: F811
            g.replay()
            return static_loss
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"config": "CoDeNet1x %dx%d %s QAT step over deconv_layers, batch %d" % (
        a.res, a.res, "fp32" if a.fp32 else "W4A8", a.batch) + (", one HIP graph" if a.graph else ", eager launches"),
        "ms_per_step": round(dt * 1e3, 3),
        "images_per_s": round(a.batch / dt, 1), "probe": float(loss)}))


if __name__ == "__main__":
    main()
