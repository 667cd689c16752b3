"""The NARROWEST drop-in: the reference's own Python composition -- 18-channel offset
tensor + generic ``deform_conv`` -- on cdn_deform_conv_forward (dcn_generic.hip), next to the CoDeNet fast path
(cdn_codenet_dw_forward) on the same stage shapes, batch 64 (argv[1]), 512x512 stage shapes; round 4: also the backward
(cdn_deform_conv_backward_input + _parameters: grad_input, 18-channel grad_offset, grad_weight) next to the module
kernel's (cdn_codenet_dw_backward: grad_x, grad_s, grad_w).  GPU only.
--smooth: a spatially smooth scale map (low-resolution noise, bilinearly up-sampled -- what a scale predicted from an
image looks like) instead of independent noise per pixel: the pixels of a wave instruction then hit neighbouring cells of
the LDS images (the default is the worst case for bank conflicts of the backward kernels' atomics)."""
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
    smooth = "--smooth" in sys.argv
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    batch = int(args[0]) if args else 64
    g = torch.Generator().manual_seed(0)
    out = {}
    anchor = make_anchor_offset().cuda()
    for (C, H) in ((1024, 16), (256, 32), (128, 64)):
        x = torch.randn(batch, C, H, H, generator=g).cuda()
        if smooth:
            low = torch.randn(batch, 1, max(H // 8, 2), max(H // 8, 2), generator=g) * 3 + 1
            s = torch.nn.functional.interpolate(low, size=(H, H), mode="bilinear", align_corners=True).clamp_(-7, 8).cuda()
        else:
            s = (torch.randn(batch, 1, H, H, generator=g) * 3 + 1).clamp_(-7, 8).cuda()
        w = (torch.randn(C, 1, 3, 3, generator=g) / 3).cuda()
        with torch.no_grad():
            t_gen = timed(lambda: deform_conv(x, anchor * (s - 1), w, 1, 1, 1, C, 1))
            t_fast = timed(lambda: ops.codenet_dw(x, s, w))
            err = (deform_conv(x, anchor * (s - 1), w, 1, 1, 1, C, 1) - ops.codenet_dw(x, s, w)).abs().max().item()
        # backward: autograd through the generic Function (two C calls) vs the module op's backward (one)
        go = torch.randn(batch, C, H, H, generator=g).cuda()

        def bwd(fn, leaves):
            for t in leaves:
                t.grad = None
            fn().backward(go)
        xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        og = (anchor * (s - 1)).detach().requires_grad_(True)
        sg = s.clone().requires_grad_(True)
        t_gen_fb = timed(lambda: bwd(lambda: deform_conv(xg, og, wg, 1, 1, 1, C, 1), (xg, og, wg)), iters=10)
        t_fast_fb = timed(lambda: bwd(lambda: ops.codenet_dw(xg, sg, wg), (xg, sg, wg)), iters=10)
        out["%dx%dx%d" % (C, H, H)] = {"generic_ms": round(t_gen, 4), "codenet_dw_ms": round(t_fast, 4),
                                       "max_abs_diff": err,
                                       "generic_bwd_ms": round(t_gen_fb - t_gen, 4),
                                       "codenet_dw_bwd_ms": round(t_fast_fb - t_fast, 4)}
    print(json.dumps({"what": "gather/depthwise, batch %d: generic deform_conv (18-channel offset tensor, "
                              "dcn_generic.hip) vs the CoDeNet module kernel (one scale plane, codenet_stage.hip); "
                              "*_bwd_ms = (forward + backward) - forward, eager launches%s"
                              % (batch, "; SMOOTH scale map" if smooth else ""),
                      "stages": out}))


if __name__ == "__main__":
    main()
