"""Times the training path's f32-MFMA pointwise (cdn_codenet_pointwise_forward: NCHW, Y[n] = W . D[n]) at the six
launches of the QAT step (batch 32, 512 x 512: forward and data gradient of the three stages).  HIP events, 20 calls
each, rotating over 4 buffer sets (the step's tensors are not cache-resident either)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from codenet_amd import _native as N_


def main():
    lib, dev = N_.lib(), torch.device("cuda", 0)
    N = 32
    out = {}
    g = torch.Generator().manual_seed(0)
    tot = 0.0
    for name, C, Co, H in (("fwd0", 1024, 256, 16), ("fwd1", 256, 128, 32), ("fwd2", 128, 64, 64),
                           ("bwd0", 256, 1024, 16), ("bwd1", 128, 256, 32), ("bwd2", 64, 128, 64)):
        HW = H * H
        ds = [torch.randn(N, C, HW, generator=g).to(dev) for _ in range(4)]
        w = (torch.randn(Co, C, generator=g) / C ** 0.5).to(dev)
        ys = [torch.empty(N, Co, HW, device=dev) for _ in range(4)]
        st = torch.cuda.current_stream().cuda_stream

        def run(i):
            N_.check(lib.cdn_codenet_pointwise_forward(ds[i & 3].data_ptr(), w.data_ptr(), None, None, None,
                                                       ys[i & 3].data_ptr(), N, C, Co, HW, 0, st), "pw")
        for i in range(4):
            run(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        ref = torch.einsum("oc,ncp->nop", w.double(), ds[3].double())
        err = (ys[3].double() - ref).abs().max().item()
        flops = 2.0 * N * C * Co * HW
        out[name] = {"us": round(us, 1), "TF": round(flops / us / 1e6, 1), "max_err": err}
        tot += us
    out["sum_us"] = round(tot, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
