"""Profile target: the whole W4A8 network in the frozen byte-code serving mode (batch 64, 512x512), eager launches.
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_fz -- python3 tools/prof_e2e_frozen.py [stages_only]"""
import sys
import torch
sys.path.insert(0, ".")
from codenet_amd import harness, pipeline

stages_only = len(sys.argv) > 1 and sys.argv[1] == "stages_only"
dev = torch.device("cuda", 0)
model = harness.create_model(quantize=True).to(dev)
model.enable_fused()
images = torch.randn(64, 3, 512, 512, generator=torch.Generator().manual_seed(0)).to(dev)
print("calibration", pipeline.prepare_serving(model, images, settle=30, margin=0.02))
with torch.no_grad():
    model.enable_fused(frozen_codes=True, frozen_backbone=not stages_only)
    for _ in range(3):
        model(images)
    torch.cuda.synchronize()
    torch.full((4,), 0.5, device=dev).erfinv_()      # a kernel that occurs nowhere else: tools/steady_stats.py's setup marker
    print("MARK begin")
    for _ in range(20):
        harness.process(model, images, flip_test=False)
    torch.cuda.synchronize()
    print("overflow", model.frozen_overflowed(), "byte backbone", model._fzbackbone is not None)
