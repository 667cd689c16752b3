"""Times the two backward C calls of the unchanged-reference seam on the CoDeNet stage shapes (batch argv[1], default 64)
with HIP events: cdn_deform_conv_backward_input_scratch (structured offsets; `--generic`: without scratch) and
cdn_deform_conv_backward_parameters.  GPU only; prints one JSON line.  For A/B of variant builds run it under
tools/with_lib.py."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from codenet_amd import _native as N_
from codenet_amd.modules.dcn_deform_conv import make_anchor_offset


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    generic = "--generic" in sys.argv
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    batch = int(args[0]) if args else 64
    lib = N_.lib()
    g = torch.Generator().manual_seed(0)
    anchor = make_anchor_offset().cuda()
    out = {}
    for (C, H) in ((1024, 16), (256, 32), (128, 64)):
        x = torch.randn(batch, C, H, H, generator=g).cuda()
        s = (torch.randn(batch, 1, H, H, generator=g) * 3 + 1).clamp_(-7, 8).cuda()
        w = (torch.randn(C, 1, 3, 3, generator=g) / 3).cuda()
        go = torch.randn(batch, C, H, H, generator=g).cuda()
        off = (anchor * (s - 1)).contiguous()
        gx, goff, gw = torch.zeros_like(x), torch.zeros_like(off), torch.zeros_like(w)
        geom = (batch, C, H, H, C, 3, 3, 1, 1, 1, 1, 1, 1, C, 1)
        need = 0 if generic else lib.cdn_deform_conv_backward_input_scratch_bytes(*geom)
        scratch = torch.empty(max(need // 4, 1), device="cuda")
        st = torch.cuda.current_stream().cuda_stream

        def call_input():
            rc = lib.cdn_deform_conv_backward_input_scratch(
                x.data_ptr(), off.data_ptr(), go.data_ptr(), gx.data_ptr(), goff.data_ptr(), w.data_ptr(), N_.CDN_F32, *geom,
                scratch.data_ptr() if need else None, need, st)
            assert rc == 0

        def call_params():
            rc = lib.cdn_deform_conv_backward_parameters(x.data_ptr(), off.data_ptr(), go.data_ptr(), gw.data_ptr(), N_.CDN_F32,
                                                         *geom, 1.0, st)
            assert rc == 0
        out["%dx%dx%d" % (C, H, H)] = {"input_ms": round(timed(call_input), 4), "parameters_ms": round(timed(call_params), 4)}
    print(json.dumps({"lib": os.path.basename(N_.SO_PATH), "batch": batch, "generic": generic, "stages": out}))


if __name__ == "__main__":
    main()
