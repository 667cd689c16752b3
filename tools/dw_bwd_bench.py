"""Times the gather backward kernels at the QAT step's stage shapes (batch 32, 512 x 512): dw_bwd2_kernel (stage 0,
and stages 1-2 on the materialised up-sampled input) and dw_bwd2u_kernel (stages 1-2 on the stored tensors).
A/B of build variants: python tools/with_lib.py codenet_amd/lib/libcodenet_dcn_<tag>.so tools/dw_bwd_bench.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from codenet_amd import _native as N_


def main():
    lib, dev = N_.lib(), torch.device("cuda", 0)
    batch = 32
    g = torch.Generator().manual_seed(0)
    out = {}
    for C, H, up in ((1024, 16, False), (256, 32, False), (256, 32, True), (128, 64, False), (128, 64, True)):
        Hx = H // 2 if up else H
        x = torch.randn(batch, C, Hx, Hx, generator=g).to(dev)
        s = (torch.rand(batch, 1, Hx, Hx, generator=g) * 2.5 + 0.5).to(dev)
        w = (torch.randn(C, 1, 3, 3, generator=g) * 0.3).to(dev)
        gd = (torch.randn(batch, C, H, H, generator=g) * 1e-3).to(dev)
        gx, gs, gw = torch.empty_like(x), torch.empty_like(s), torch.zeros_like(w)
        st = torch.cuda.current_stream().cuda_stream
        fn = lib.cdn_codenet_dw_up2_backward if up else lib.cdn_codenet_dw_backward

        def run():
            N_.check(fn(x.data_ptr(), s.data_ptr(), w.data_ptr(), gd.data_ptr(), gx.data_ptr(), gs.data_ptr(),
                        gw.data_ptr(), batch, C, H, H, st), "dw backward")
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        out["%dx%d C%d %s" % (H, H, C, "stored" if up else "full")] = round(e0.elapsed_time(e1) / 20 * 1e3, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
