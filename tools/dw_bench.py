"""Isolated timing of the depthwise 3x3 kernel at the ShuffleNetV2 unit shapes (GPU only):
python tools/dw_bench.py [N C H W stride]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from codenet_amd import _native as N_, ops


def graph_time(run, reps=20):
    """GPU time per launch: `reps` launches captured in one HIP graph (no CPU launch cost in the number)."""
    global stream
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        stream = side.cuda_stream
        for _ in range(3):
            run()
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            for _ in range(reps):
                run()
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / (5 * reps) * 1e6, 1)


def main():
    global stream
    N, C, H, W, stride = (int(v) for v in sys.argv[1:6]) if len(sys.argv) > 5 else (64, 58, 64, 64, 1)
    dev = torch.device("cuda", 0)
    lib = N_.lib()
    aux = lib.cdn_codenet_aux_workspace_bytes()
    ws = torch.zeros(aux // 4 + 64, device=dev)
    wp = (ws.data_ptr() + 255) // 256 * 256
    wb = (ws.numel() * 4 - (wp - ws.data_ptr())) // 256 * 256
    ld = (C + 3) // 4 * 4
    a = torch.randn(N, H * W, ld, device=dev)
    w = torch.randn(C, 9, device=dev)
    b = torch.randn(C, device=dev)
    Ho, Wo = ((H - 1) // 2 + 1, (W - 1) // 2 + 1) if stride == 2 else (H, W)
    out = torch.empty(N, Ho * Wo, ld, device=dev)
    states = torch.zeros(8, dtype=torch.int32, device=dev)
    sf = states.view(torch.float32)
    sf[2], sf[3] = 255.0 / 8.0, -100.0
    xmin, xmax, st = torch.zeros(1, device=dev), torch.zeros(1, device=dev), ops.quantact_state(dev)
    stream = torch.cuda.current_stream().cuda_stream
    res = {}
    for name, aq in (("final", None), ("single", states.data_ptr())):
        def run():
            rc = lib.cdn_codenet_dw3x3_nhwc_forward(
                a.data_ptr(), aq, N, C, H, W, 0, stride, ld, ld, w.data_ptr(), b.data_ptr(), None, None, 0,
                xmin.data_ptr(), xmax.data_ptr(), st.data_ptr(), 8, 0.99, 1, wp, wb, out.data_ptr(), stream)
            N_.check(rc, "dw")
        res[name] = graph_time(run)
    res["shape"] = [N, C, H, W, stride]
    res["MB"] = round((N * H * W * C * 4 + N * Ho * Wo * C * 4) / 1e6, 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
