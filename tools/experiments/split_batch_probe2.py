"""Sub-batches as SEPARATE HIP graphs replayed on separate streams (one linear graph per lane)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from codenet_amd import harness, pipeline


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--steps", type=int, default=30)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    images = torch.randn(a.batch, 3, a.res, a.res, generator=torch.Generator().manual_seed(0)).to(dev)
    gen = torch.Generator().manual_seed(7001)
    cal = [torch.randn(images.shape, generator=gen).to(dev) for _ in range(3)]
    model = harness.create_model(quantize=True, seed=317).to(dev)
    pipeline.prepare_serving(model, images, settle=100, margin=0.02, more_batches=cal, sigmas=6.0)
    del cal
    out = {}

    def timed(replay):
        for _ in range(5):
            replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.steps * 1e3

    base = harness.capture_process(model, images)
    out["lanes_1_ms"] = timed(base)
    ref = base()[1].clone()
    for lanes in (2, 4):
        n = a.batch // lanes
        models, replays, streams = [], [], []
        for i in range(lanes):
            m = harness.create_model(quantize=True, seed=317).to(dev)
            m.load_state_dict(model.state_dict())
            pipeline.set_running_stat(m, False)
            m.enable_fused(frozen_codes=True)
            models.append(m)
            replays.append(harness.capture_process(m, images[i * n:(i + 1) * n]))
            streams.append(torch.cuda.Stream())
        main = torch.cuda.current_stream()

        def replay():
            res = []
            for s in streams:
                s.wait_stream(main)
            for r, s in zip(replays, streams):
                with torch.cuda.stream(s):
                    res.append(r()[1])
            for s in streams:
                main.wait_stream(s)
            return res
        out["lanes_%d_ms" % lanes] = timed(replay)
        dets = torch.cat(replay(), 0)
        torch.cuda.synchronize()
        out["lanes_%d_equal" % lanes] = bool(torch.equal(dets, ref))
        out["lanes_%d_overflow" % lanes] = [bool(m.frozen_overflowed()) for m in models]
    print("SPLIT2 " + json.dumps(out))


main()
