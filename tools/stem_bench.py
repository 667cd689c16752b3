import sys, os, json
sys.path.insert(0, ".")
import torch
from codenet_amd import harness, pipeline
dev = torch.device("cuda", 0)
model = harness.create_model(quantize=True).to(dev)
x = torch.randn(64, 3, 512, 512, device=dev)
fb = pipeline.FusedBackbone(model)
for _ in range(3):
    fb(x)
from codenet_amd import _native as N_
m = model
q0, act0 = m.layer0[0], m.layer0[1][1]
w0, b0 = fb._folded(q0)
out = torch.empty(64, 128 * 128, 24, device=dev)
def run():
    rc = N_.lib().cdn_codenet_stem_forward(x.data_ptr(), 64, 512, 512, 24, 4, w0.reshape(24, 27).data_ptr(), b0.data_ptr(), 1,
        *fb._act_args(act0, dev), fb._ws_ptr, fb._ws_bytes, out.data_ptr(), fb._stream)
    N_.check(rc, "stem")
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print(json.dumps({"stem_us": round(e0.elapsed_time(e1) / 20 * 1e3, 1)}))
