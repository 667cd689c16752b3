"""Timing of the native ctdet_decode at the BASELINE shape (batch 64,
20 classes, 128x128, K = 100) against the PyTorch-ROCm composition.  GPU only."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from codenet_amd import harness


def timed(fn, steps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(1)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    logits = (torch.randn(B, 20, 128, 128, generator=g) * 0.05 - 1.75).to(dev)   # scores ~0.148 +- 0.006
    wh = (torch.rand(B, 2, 128, 128, generator=g) * 9).to(dev)
    reg = torch.rand(B, 2, 128, 128, generator=g).to(dev)
    res = {}
    res["torch_ms"] = timed(lambda: harness.ctdet_decode(logits.clone().sigmoid_(), wh, reg=reg, K=100))
    res["native_ms"] = timed(lambda: harness.ctdet_decode_native(logits, wh, reg=reg, K=100, apply_sigmoid=True))
    buf = torch.empty_like(logits)
    res["native_inplace_sigmoid_ms"] = timed(lambda: harness.ctdet_decode_native(
        logits, wh, reg=reg, K=100, apply_sigmoid=True, heat_out=buf))
    print(json.dumps({k: round(v, 4) for k, v in res.items()}))


if __name__ == "__main__":
    main()
