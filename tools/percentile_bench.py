"""--act-percentile hot path (three deform stages, cfg3 shape, batch 64): fused schedule with CDN_X_ACT_PERCENTILE
against the module-by-module path (both on cdn_kth_values), ms per step, eager launches.  GPU only."""
import copy
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from codenet_amd import pipeline
from codenet_amd.portable_quantizer.quant_modules import QuantAct


def timed(fn, steps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    net = pipeline.build_hot_path().cuda().eval()
    for m in net.modules():
        if isinstance(m, QuantAct):
            m.percentile = True
    ref = copy.deepcopy(net)
    fused = pipeline.FusedHotPath(net.deconv_layers)
    x = torch.randn(64, 1024, 16, 16, generator=torch.Generator().manual_seed(0)).cuda()
    with torch.no_grad():
        res = {"fused_percentile_ms": round(timed(lambda: fused(x)), 4), "module_path_percentile_ms": round(timed(lambda: ref(x)), 4)}
        plain = pipeline.build_hot_path().cuda().eval()
        fp = pipeline.FusedHotPath(plain.deconv_layers)
        res["fused_minmax_ms"] = round(timed(lambda: fp(x)), 4)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
