"""Stage-0 int8 pointwise (K = 1024, N = 256) alone, time per launch against M (HIP-graph replay of 20 launches):
is a k-tile's 0.87 us a per-workgroup latency or a per-CU throughput?"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
import torch

import pw_bench as PB
from codenet_amd import _native as N_, ops

dev = torch.device("cuda", 0)
lib = N_.lib()
aux = lib.cdn_codenet_aux_workspace_bytes()
ws = torch.zeros(aux // 4 + 64, device=dev)
wp = (ws.data_ptr() + 255) // 256 * 256
wb = (ws.numel() * 4 - (wp - ws.data_ptr())) // 256 * 256
C, Co = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 256)
g = torch.Generator().manual_seed(0)
q = torch.randint(-8, 8, (Co, C), generator=g)
codes = q.to(torch.int8).contiguous().to(dev)
scale = (torch.rand(Co, generator=g) * 20 + 1).to(dev)
bias = torch.randn(Co, generator=g).to(dev)
w = (q.float() / scale.cpu()[:, None]).contiguous().to(dev)
colsum = q.sum(1).to(torch.int32).to(dev)
states = torch.zeros(8, dtype=torch.int32, device=dev)
sf = states.view(torch.float32)
sf[2], sf[3] = 255.0 / 8.0, -100.0
xmin, xmax, st = torch.zeros(1, device=dev), torch.zeros(1, device=dev), ops.quantact_state(dev)
res = {}
for M in (2048, 4096, 8192, 16384, 32768, 65536, 131072):
    a = torch.randn(M, C, generator=g).to(dev)
    out = torch.empty(M, Co, device=dev)

    def run():
        rc = lib.cdn_codenet_pointwise_mixed_forward(
            a.data_ptr(), states.data_ptr(), None, M, C, Co, C, Co, w.data_ptr(), codes.data_ptr(), scale.data_ptr(),
            colsum.data_ptr(), bias.data_ptr(), None, None, 1, None, xmin.data_ptr(), xmax.data_ptr(),
            st.data_ptr(), 8, 0.99, 1, wp, wb, out.data_ptr(), PB.stream)
        N_.check(rc, "pw")
    PB.stream = torch.cuda.current_stream().cuda_stream
    res[M] = PB.graph_time(run)
print(json.dumps({"K": C, "N": Co, "us_per_launch_by_M": res}))
