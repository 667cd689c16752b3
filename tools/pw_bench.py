"""Isolated timing of the pointwise kernels at the ShuffleNetV2 unit shapes (GPU only):
python tools/pw_bench.py [M C Co lda]  -- modes: final / single-state / mixed input, dense or mapped output."""
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
    M, C, Co, lda = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (262144, 116, 58, 116)
    dev = torch.device("cuda", 0)
    lib = N_.lib()
    aux = lib.cdn_codenet_aux_workspace_bytes()
    ws = torch.zeros(aux // 4 + 64, device=dev)
    wp = (ws.data_ptr() + 255) // 256 * 256
    wb = (ws.numel() * 4 - (wp - ws.data_ptr())) // 256 * 256
    g = torch.Generator().manual_seed(0)
    a = torch.randn(M, lda, generator=g).to(dev)
    cpad = (C + 63) // 64 * 64
    q = torch.randint(-8, 8, (Co, C), generator=g)
    codes = torch.zeros(Co, cpad, dtype=torch.int8)
    codes[:, :C] = q.to(torch.int8)
    codes = codes.to(dev)
    scale = (torch.rand(Co, generator=g) * 20 + 1).to(dev)
    bias = torch.randn(Co, generator=g).to(dev)
    w = (q.float() / scale.cpu()[:, None]).contiguous().to(dev)
    colsum = q.sum(1).to(torch.int32).to(dev)
    ldo = (Co + 3) // 4 * 4
    out = torch.empty(M, max(ldo, 2 * Co), device=dev)
    states = torch.zeros(8 * 8, dtype=torch.int32, device=dev)
    sf = states.view(torch.float32).view(8, 8)
    sf[:, 2] = 255.0 / 8.0
    sf[:, 3] = -100.0
    gen = (torch.arange(C) % 5).to(torch.uint8).to(dev)
    omap = (torch.arange(Co) * 2 + 1).to(torch.int32).to(dev)
    xmin, xmax, st = torch.zeros(1, device=dev), torch.zeros(1, device=dev), ops.quantact_state(dev)
    stream = torch.cuda.current_stream().cuda_stream
    res = {}
    for name, aq, ag, om, ld_o, ng in (("final", None, None, None, ldo, 0), ("single", states.data_ptr(), None, None, ldo, 0),
                                       ("mixed", states.data_ptr(), gen.data_ptr(), None, ldo, 0),
                                       ("mixed_n", states.data_ptr(), gen.data_ptr(), None, ldo, 8),
                                       ("mixed_mapped", states.data_ptr(), gen.data_ptr(), omap.data_ptr(), 2 * Co, 0),
                                       ("mixed_mapped_n", states.data_ptr(), gen.data_ptr(), omap.data_ptr(), 2 * Co, 8)):
        def run():
            rc = lib.cdn_codenet_pointwise_mixed_forward_n(
                a.data_ptr(), aq, ag, ng, M, C, Co, lda, ld_o, w.data_ptr(), codes.data_ptr(), scale.data_ptr(),
                colsum.data_ptr(), bias.data_ptr(), None, None, 1, om, xmin.data_ptr(), xmax.data_ptr(),
                st.data_ptr(), 8, 0.99, 1, wp, wb, out.data_ptr(), stream)
            N_.check(rc, "pw")
        res[name] = graph_time(run)
        if hasattr(lib, "cdn_debug_read_stamps") and os.environ.get("CDN_STAMP_MODE", "mixed") == name:
            import ctypes
            import numpy as np
            lib.cdn_debug_clear_stamps()
            stream = torch.cuda.current_stream().cuda_stream
            run()
            torch.cuda.synchronize()
            buf = np.zeros(3 * 2048 * 8 + 64, dtype=np.uint64)
            lib.cdn_debug_read_stamps.argtypes = [ctypes.c_void_p]
            lib.cdn_debug_read_stamps(buf.ctypes.data)
            sd = buf[:3 * 2048 * 8].reshape(3, 2048, 8)[2].astype(np.float64) / 100.0
            sd = sd[sd[:, 0] > 0]
            t0 = sd[:, 0].min()
            ph = ["prologue", "first block k-loop", "remaining blocks + stores", "finish"]
            print("workgroups %d, span %.1f us; starts: p50 %.1f p90 %.1f max %.1f us" % (
                len(sd), sd[:, 4].max() - t0, *np.percentile(sd[:, 0] - t0, [50, 90, 100])))
            print("  start deciles:", np.round(np.percentile(sd[:, 0] - t0, range(0, 101, 10)), 1).tolist())
            for i, nm in enumerate(ph):
                d = sd[:, i + 1] - sd[:, i]
                print("  %-28s mean %.2f  p90 %.2f us" % (nm, d.mean(), np.percentile(d, 90)))
    res["shape"] = [M, C, Co, lda]
    res["MB"] = round((M * C * 4 + M * Co * 4) / 1e6, 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
