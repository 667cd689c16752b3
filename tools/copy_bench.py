"""What a plain copy of a pointwise layer's bytes takes (graph replay of 20 copies): the floor the int8 pointwise launches
of the backbone are compared with.  python tools/copy_bench.py M C Co"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch

import pw_bench as PB

M, C, Co = (int(v) for v in sys.argv[1:4])
dev = torch.device("cuda", 0)
a = torch.randn(M, C, device=dev)
b = torch.empty(M, C, device=dev)
o1 = torch.randn(M, Co, device=dev)
o2 = torch.empty(M, Co, device=dev)


def run():
    # read M x C, write M x Co: one copy of each size / 2 would under-count; copy A->B is read+write of C columns
    b.copy_(a)


def run2():
    o2.copy_(o1)


PB.stream = torch.cuda.current_stream().cuda_stream
t1, t2 = PB.graph_time(run), PB.graph_time(run2)
print(json.dumps({"copy_MxC_us": t1, "copy_MxCo_us": t2, "MB_read_plus_write": round(M * (C + Co) * 4 / 1e6, 1),
                  "floor_us_for_read_C_write_Co": round((t1 + t2) / 2, 1)}))
