"""Serving mode of the whole W4A8 network (every QuantAct frozen, byte codes from the stem to the heads) as one HIP
graph, batch 64 at 512x512: ms per batch.  The calibration is short (settle 30): an A/B timing tool, the judged
number comes from bench.py's `e2e.frozen` leg (settle 300).  GPU only.
    python tools/e2e_frozen_bench.py [--steps 20] [--stages-only]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from codenet_amd import harness, pipeline


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--no-merge-params", action="store_true", help="A/B: the stages' and the heads' QuantAct parameters in "
                    "launches of their own instead of in the backbone's first launch")
    ap.add_argument("--stages-only", action="store_true", help="backbone on the fp32 kernels, stages on byte codes")
    ap.add_argument("--w2", action="store_true", help="CoDeNet2x (BASELINE cfg4's model; batch 32 per GPU there): stage 0 on "
                    "the fp32 frozen schedule, everything else on byte codes")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    model = harness.create_model(quantize=True, w2=a.w2).to(dev)
    model.merge_frozen_params = not a.no_merge_params
    model.enable_fused()
    images = torch.randn(a.batch, 3, 512, 512, generator=torch.Generator().manual_seed(0)).to(dev)
    report = pipeline.prepare_serving(model, images, settle=30, margin=0.02)
    with torch.no_grad():
        model.enable_fused(frozen_codes=True, frozen_backbone=not a.stages_only)
        replay = harness.capture_process(model, images)
        for _ in range(5):
            replay()
        torch.cuda.synchronize()
        model.frozen_overflowed()
        res = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(a.steps):
                replay()
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / a.steps * 1e3)
    print(json.dumps({"ms_per_batch": [round(r, 4) for r in res], "overflow": bool(model.frozen_overflowed()),
                      "byte_backbone": model._fzbackbone is not None, "calibration_clean": report.get("clean"),
                      "model": "CoDeNet2x" if a.w2 else "CoDeNet1x", "batch": a.batch}))


if __name__ == "__main__":
    main()
