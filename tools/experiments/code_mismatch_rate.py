"""Fraction of the fused W4A8 hot path's outputs that differ from the CPU oracle (one code LSB each), and the largest
relative deviation of the nine tracked ranges, at the cfg3 stage shapes (batch 4, three forwards).  Run with
run under tools/with_lib.py <variant .so> to compare library builds."""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch

import test_gpu_real_shapes as T
from codenet_amd import pipeline

planes = T.CFG4 if "--cfg4" in sys.argv else T.CFG3
n = int(sys.argv[sys.argv.index("--n") + 1]) if "--n" in sys.argv else 4
res, forwards, seed = 16, 3, 11
net = pipeline.build_hot_path(quantized=True, planes=planes, seed=seed)
net_cpu = copy.deepcopy(net)
xs = T._inputs(n, planes[0], res, forwards, seed + 100)
ref, ref_ranges = T._oracle(net_cpu, xs, True)
net = net.cuda()
pipeline.set_running_stat(net, True)
fused = pipeline.FusedHotPath(net.deconv_layers)
for it in range(forwards):
    y = fused(xs[it].cuda()).cpu()
    diff = (y - ref[it]).abs()
    got = T._gpu_ranges(net)
    rel = max(max(abs(a0 - b0), abs(a1 - b1)) / max(abs(b0), abs(b1), 1e-30)
              for (a0, a1), (b0, b1) in zip(got, ref_ranges[it]))
    exact = sum(int(a0 == b0) + int(a1 == b1) for (a0, a1), (b0, b1) in zip(got, ref_ranges[it]))
    print("forward %d: %.4g %% of %d outputs differ (max %.3g), ranges: %d / 18 bit-equal to the oracle, max rel dev %.2g"
          % (it, 100 * (diff > 1e-3).float().mean().item(), diff.numel(), diff.max().item(), exact, rel))
