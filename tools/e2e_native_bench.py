"""Whole network + decode on the HIP kernels only (harness.enable_fused + ctdet_decode_native), for timing
and for `rocprofv3 --kernel-trace --stats -- python3 tools/e2e_native_bench.py`.  GPU only."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from codenet_amd import harness


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--graph", action="store_true", help="replay the whole step as one HIP graph")
    ap.add_argument("--no-range-first", action="store_true", help="A/B: the recomputed conv's range pass beside branch 1 "
                    "(before round 4's reordering) instead of in front of the fork")
    ap.add_argument("--no-recompute", action="store_true", help="A/B: layer 1's first 1x1 conv stored (round 3) instead "
                    "of recomputed inside its depthwise")
    ap.add_argument("--no-preload-states", action="store_true", help="A/B: the mixed-generation pointwise without n_gens "
                    "(states loaded at addresses the generation bytes give: a second round trip of its prologue)")
    a = ap.parse_args()
    if a.no_preload_states:
        from codenet_amd import pipeline
        pipeline.FusedBackbone.preload_states = False
    if a.no_recompute:
        from codenet_amd import pipeline
        pipeline.FusedBackbone.recompute_pw1 = False
    if a.no_range_first:
        from codenet_amd import pipeline
        pipeline.FusedBackbone.range_first = False
    dev = torch.device("cuda", 0)
    model = harness.create_model(quantize=not a.fp32).to(dev).enable_fused()
    x = torch.randn(a.batch, 3, a.res, a.res, device=dev)

    def step():
        return harness.process(model, x, flip_test=False)[1]
    if a.graph:
        replay = harness.capture_process(model, x)

        def step():   # noqa: F811
            return replay()[1]
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        dets = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    assert torch.isfinite(dets).all()
    print(json.dumps({"config": "CoDeNet1x %dx%d %s batch %d, whole network + ctdet_decode native" % (
        a.res, a.res, "fp32" if a.fp32 else "W4A8", a.batch) + (", HIP graph" if a.graph else ""),
        "ms_per_batch": round(ms, 4),
        "images_per_s": round(a.batch / ms * 1e3)}))


if __name__ == "__main__":
    main()
