"""Experiment: the frozen byte-code schedule has no batch-global dependency, so a batch can be cut into sub-batches
that run on separate streams (inside one HIP graph): HBM-bound kernels (scale, pointwise) of one sub-batch can
overlap the VALU/LDS-bound gather of another.  Prints ms per batch of 64 for 1, 2 and 4 sub-batches."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from codenet_amd import pipeline

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
net = pipeline.build_hot_path(quantized=True).cuda()
x = pipeline.make_input(batch, 512, device="cuda")
pipeline.set_running_stat(net, True)
warm = pipeline.FusedHotPath(net.deconv_layers)
for _ in range(3):
    warm.forward_nhwc(x)
pipeline.set_running_stat(net, False)

for parts in (1, 2, 4):
    n = batch // parts
    xs = [x[i * n:(i + 1) * n].contiguous() for i in range(parts)]
    paths = [pipeline.FrozenHotPath(net.deconv_layers) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]

    def run():
        main = torch.cuda.current_stream()
        outs = []
        for p, xi, st in zip(paths, xs, streams):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                outs.append(p.forward_codes(xi)[0])
        for st in streams:
            main.wait_stream(st)
        return outs
    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = run()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 50 * 1e3
    flags = [p.overflowed() for p in paths]
    print("parts %d: %.4f ms per batch of %d  (%.0f images/s)  overflow %s" % (parts, ms, batch, batch / ms * 1e3, flags))
