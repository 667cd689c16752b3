"""Does the launch-bound byte-code serving network gain from running sub-batches CONCURRENTLY on several streams?
With every QuantAct frozen no value couples two images, so batch 64 = 2 x 32 = 4 x 16 bit for bit; each lane's chain has
the same number of launches on half / a quarter of the work, and the lanes' fixed per-launch latencies overlap.
    python tools/experiments/split_batch_probe.py [--batch 64]"""
import argparse
import copy
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
    print('prepare', flush=True)
    pipeline.prepare_serving(model, images, settle=100, margin=0.02, more_batches=cal, sigmas=6.0)
    print('prepared', flush=True)
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
    print('captured', flush=True)
    out["lanes_1_ms"] = timed(base)
    ref = base()[1].clone()
    for lanes in (2, 4):
        n = a.batch // lanes
        models = []
        for _ in range(lanes):
            m = harness.create_model(quantize=True, seed=317).to(dev)
            m.load_state_dict(model.state_dict())
            pipeline.set_running_stat(m, False)
            m.enable_fused(frozen_codes=True)
            models.append(m)
        print('lanes', lanes, flush=True)
        streams = [torch.cuda.Stream() for _ in range(lanes)]
        bufs = [harness.ProcessBuffers() for _ in range(lanes)]
        main_s = torch.cuda.Stream()
        main_s.wait_stream(torch.cuda.current_stream())

        res_triv = [torch.zeros(8, 100, 6, device=dev) for _ in range(lanes)]

        def run():
            res = [None] * lanes
            cur = torch.cuda.current_stream()
            for s in streams[1:]:
                s.wait_stream(cur)
            for i in range(1, lanes):
                with torch.cuda.stream(streams[i] if os.environ.get("SPLIT_SEQ") != "1" else cur):
                    what = os.environ.get("SPLIT_SIDE", "full")
                    if what == "trivial":
                        res[i] = res_triv[i].add_(1.0)
                        continue
                    if what == "forward":
                        res[i] = models[i](images[i * n:(i + 1) * n])[-1]["hm"]
                        continue
                    if what == "backbone":
                        res[i] = models[i]._fzbackbone(images[i * n:(i + 1) * n])[0]
                        continue
                    res[i] = harness.process(models[i], images[i * n:(i + 1) * n], flip_test=False, bufs=bufs[i])[1]
            res[0] = harness.process(models[0], images[0:n], flip_test=False, bufs=bufs[0])[1]      # lane 0: the capture stream
            for s in streams[1:]:
                cur.wait_stream(s)
            return res
        with torch.cuda.stream(main_s):
            run()
            if os.environ.get("SPLIT_NO_INNER", "1") == "1":      # the lanes' own side streams off: the lanes are the concurrency
                for m in models:
                    for obj in (m._fzheads, m._fzbackbone, m._fheads, m._fbackbone):
                        if obj is not None:
                            if hasattr(obj, "streams"):
                                obj.streams = False
                            if hasattr(obj, "two_streams"):
                                obj.two_streams = False
            for _ in range(3):
                run()
        torch.cuda.current_stream().wait_stream(main_s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main_s):
            res = run()

        def replay():
            g.replay()
            return res
        out["lanes_%d_ms" % lanes] = timed(replay)
        if os.environ.get("SPLIT_SIDE", "full") == "full":
            dets = torch.cat(replay(), 0)
            torch.cuda.synchronize()
            out["lanes_%d_equal" % lanes] = bool(torch.equal(dets, ref))
        out["lanes_%d_overflow" % lanes] = [bool(m.frozen_overflowed()) for m in models]
    print("SPLIT " + json.dumps(out))


main()
