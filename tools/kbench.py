"""Per-kernel micro-benchmark of the hot path at the BASELINE stage shapes.
Prints achieved algorithmic GB/s / TFLOP/s per kernel.  GPU only."""
import argparse
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from codenet_amd import ops


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters  # ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--w2", action="store_true")
    ap.add_argument("--bwd", action="store_true")
    a = ap.parse_args()
    c0 = 2153 if a.w2 else 1024
    r = a.res // 32
    stages = [(c0, 256, r, r), (256, 128, 2 * r, 2 * r), (128, 64, 4 * r, 4 * r)]
    N = a.batch
    dev = "cuda"
    rows = []
    for (C, Co, H, W) in stages:
        x = torch.randn(N, C, H, W, device=dev)
        ws = torch.randn(1, C, 1, 1, device=dev) * (2.0 / C ** 0.5)
        bs = torch.ones(1, device=dev)
        wd = torch.randn(C, 1, 3, 3, device=dev) / 3
        wp = torch.randn(Co, C, 1, 1, device=dev) / C ** 0.5
        s = ops.codenet_scale(x, ws, bs, -7.0, 8.0)
        d = ops.codenet_dw(x, s, wd)
        HW = H * W
        t = timeit(lambda: ops.codenet_scale(x, ws, bs, -7.0, 8.0))
        rows.append(("scale", C, Co, H, t, (C + 1) * HW * 4 * N / t / 1e6, None))
        t = timeit(lambda: ops.codenet_dw(x, s, wd))
        rows.append(("dw", C, Co, H, t, (2 * C + 1) * HW * 4 * N / t / 1e6, None))
        t = timeit(lambda: ops.codenet_pointwise(d, wp))
        rows.append(("pointwise", C, Co, H, t, (C + Co) * HW * 4 * N / t / 1e6,
                     2.0 * C * Co * HW * N / t / 1e9))
        xm, xM, st = torch.zeros(1, device=dev), torch.zeros(1, device=dev), ops.quantact_state(dev)
        t = timeit(lambda: ops.quantact_forward(d, xm, xM, st))
        rows.append(("quantact", C, Co, H, t, 3 * C * HW * 4 * N / t / 1e6, None))
        if a.bwd:
            go = torch.randn_like(d)
            gx, gs, gw = torch.empty_like(x), torch.empty_like(s), torch.zeros_like(wd)
            from codenet_amd import _native as N_
            def bwd():
                rc = N_.lib().cdn_codenet_dw_backward(x.data_ptr(), s.data_ptr(), wd.data_ptr(),
                                                      go.data_ptr(), gx.data_ptr(), gs.data_ptr(),
                                                      gw.data_ptr(), N, C, H, W,
                                                      torch.cuda.current_stream().cuda_stream)
                assert rc == 0
            t = timeit(bwd, iters=5, warm=1)
            rows.append(("dw_bwd", C, Co, H, t, (4 * C + 2) * HW * 4 * N / t / 1e6, None))
    tot = 0.0
    for name, C, Co, H, t, gbs, tf in rows:
        if name != "dw_bwd":
            tot += t
        print("%-10s C=%-5d Co=%-4d HxW=%3dx%-3d  %8.3f ms  %8.1f GB/s%s" % (
            name, C, Co, H, H, t, gbs, ("  %7.2f TFLOP/s" % tf) if tf else ""))
    print("sum fwd kernels %.3f ms -> %.0f img/s (N=%d)" % (tot, N / tot * 1e3, N))


if __name__ == "__main__":
    main()
