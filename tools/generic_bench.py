"""The NARROWEST drop-in: the reference's own Python composition -- 18-channel offset
tensor + generic ``deform_conv`` -- on cdn_deform_conv_forward (dcn_generic.hip), next to the CoDeNet fast path
(cdn_codenet_dw_forward) on the same stage shapes.  Forward only, batch 64, 512x512 stage shapes.  GPU only."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from codenet_amd import ops
from codenet_amd.functions.dcn_deform_conv import deform_conv
from codenet_amd.modules.dcn_deform_conv import make_anchor_offset


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    g = torch.Generator().manual_seed(0)
    out = {}
    anchor = make_anchor_offset().cuda()
    for (C, H) in ((1024, 16), (256, 32), (128, 64)):
        x = torch.randn(batch, C, H, H, generator=g).cuda()
        s = (torch.randn(batch, 1, H, H, generator=g) * 3 + 1).clamp_(-7, 8).cuda()
        w = (torch.randn(C, 1, 3, 3, generator=g) / 3).cuda()
        with torch.no_grad():
            t_gen = timed(lambda: deform_conv(x, anchor * (s - 1), w, 1, 1, 1, C, 1))
            t_fast = timed(lambda: ops.codenet_dw(x, s, w))
            err = (deform_conv(x, anchor * (s - 1), w, 1, 1, 1, C, 1) - ops.codenet_dw(x, s, w)).abs().max().item()
        out["%dx%dx%d" % (C, H, H)] = {"generic_ms": round(t_gen, 4), "codenet_dw_ms": round(t_fast, 4),
                                       "max_abs_diff": err}
    print(json.dumps({"what": "gather/depthwise forward, batch %d: generic deform_conv (offset tensor, one thread per "
                              "output, dcn_generic.hip) vs the CoDeNet module kernel (LDS planes, codenet_stage.hip)" % batch,
                      "stages": out}))


if __name__ == "__main__":
    main()
