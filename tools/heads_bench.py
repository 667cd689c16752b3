"""Per-kernel timing of the fused detection heads at the BASELINE shape:
CoDeNet1x 512x512, batch 64 -> heads at 128x128 on the hot path's half-resolution output.  GPU only."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from codenet_amd import harness, ops, pipeline


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--one-stream", action="store_true", help="A/B: the three heads back to back on one stream")
    ap.add_argument("--graph", action="store_true", help="time a captured graph of the heads (as inside the e2e graph)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    model = harness.create_model(quantize=not a.fp32).to(dev)
    feat = pipeline.make_input(a.batch, a.res, device=dev)
    path = pipeline.FusedHotPath(model.deconv_layers)
    heads = pipeline.FusedHeads({h: getattr(model, h) for h in model.heads}, streams=not a.one_stream)
    for _ in range(5):
        heads(*path.forward_nhwc(feat))
    torch.cuda.synchronize()
    r, rq, shape = path.forward_nhwc(feat)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.steps):
        heads(r, rq, shape)
    e1.record()
    torch.cuda.synchronize()
    total = e0.elapsed_time(e1) / a.steps
    if a.graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            heads(r, rq, shape)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            heads(r, rq, shape)
        for _ in range(3):
            g.replay()
        e0.record()
        for _ in range(a.steps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        total = e0.elapsed_time(e1) / a.steps
    with ops.KernelTimer({"head_pw", "head_dw", "head_range", "head_tail_small", "head_tail"}) as kt:
        for _ in range(a.steps):
            heads(r, rq, shape)
    torch.cuda.synchronize()
    per = {"%s%s" % (k[0], list(k[1])): round(sum(v) / len(v) * 1e3 * (len(v) / a.steps), 1)
           for k, v in kt.durations_ms().items()}
    print(json.dumps({"heads_ms": round(total, 4), "images_per_s": round(a.batch / total * 1e3),
                      "kernel_us_per_step": per}))


if __name__ == "__main__":
    main()
