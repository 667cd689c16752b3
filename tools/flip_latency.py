"""Latency of the reference's test-time procedure (README test commands: --flip_test, one image + its mirror per call)
on the native network: harness.process(flip_test=True), eager, and the same pair without the flip merge as one HIP graph."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from codenet_amd import harness

res = int(sys.argv[1]) if len(sys.argv) > 1 else 512
model = harness.create_model(quantize=True).cuda()
model.enable_fused()
img = torch.randn(1, 3, res, res, generator=torch.Generator().manual_seed(0)).cuda()
pair = torch.cat([img, torch.flip(img, [3])], 0)
out = {}
for name, fn in (("flip_eager", lambda: harness.process(model, pair, flip_test=True)),
                 ("noflip_eager_batch2", lambda: harness.process(model, pair, flip_test=False))):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    out[name] = round((time.perf_counter() - t0) / 100 * 1e3, 3)
replay = harness.capture_process(model, pair)
for _ in range(20):
    replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    replay()
torch.cuda.synchronize()
out["noflip_graph_batch2"] = round((time.perf_counter() - t0) / 200 * 1e3, 3)
if hasattr(harness, "capture_process") and "flip_test" in harness.capture_process.__code__.co_varnames:
    replay = harness.capture_process(model, pair, flip_test=True)
    for _ in range(20):
        replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        replay()
    torch.cuda.synchronize()
    out["flip_graph"] = round((time.perf_counter() - t0) / 200 * 1e3, 3)
print(json.dumps({"res": res, "ms_per_call": out}))
