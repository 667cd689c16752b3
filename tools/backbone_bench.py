"""Timing of the fused W4A8 backbone at the BASELINE shape, for rocprofv3
(`rocprofv3 --kernel-trace --stats -- python3 tools/backbone_bench.py`).  GPU only."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from codenet_amd import harness, pipeline


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dev = torch.device("cuda", 0)
    model = harness.create_model(quantize=True).to(dev)
    x = torch.randn(B, 3, 512, 512, device=dev)
    fb = pipeline.FusedBackbone(model)
    for _ in range(5):
        fb(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fb(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print(json.dumps({"backbone_fused_ms": round(ms, 4), "images_per_s": round(B / ms * 1e3)}))


if __name__ == "__main__":
    main()
