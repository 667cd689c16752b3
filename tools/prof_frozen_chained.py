"""Profile target: the byte-code hot-path schedule fed with byte codes at stage 0 and chained scale sums
(`frozen_int8.codes_in_chained_scale` of bench.py: FrozenHotPath(chain_scale=True).forward_codes on an int8
[N, H*W, C] tensor), batch 64, 512 x 512, eager launches.
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <dir> -- python3 tools/prof_frozen_chained.py"""
import sys
import torch
sys.path.insert(0, ".")
from codenet_amd import pipeline
from codenet_amd.portable_quantizer.quant_modules import QuantAct

dev = torch.device("cuda", 0)
batch = 64
net = pipeline.build_hot_path(quantized=True).to(dev).eval()
x = pipeline.make_input(batch, 512, device=dev)
fused = pipeline.FusedHotPath(net.deconv_layers)
act_in = QuantAct(8, quant_mode="asymmetric").to(dev)
with torch.no_grad():
    xq = act_in(x)
    pipeline.set_running_stat(net, True)
    for _ in range(300):
        fused.forward_nhwc(xq)
    pipeline.set_running_stat(net, False)
act_in.running_stat = False
stq = act_in._device_state(dev).view(torch.float32)
Nb, C0, H0, W0 = x.shape
x8 = torch.round(stq[2] * xq - stq[3]).clamp_(-128, 127).to(torch.int8).permute(0, 2, 3, 1).reshape(Nb, H0 * W0, C0).contiguous()
qptr = act_in._device_state(dev).data_ptr()
frz = pipeline.FrozenHotPath(net.deconv_layers, chain_scale=True)
torch.cuda.synchronize()
print("MARK begin")
for _ in range(6):
    frz.forward_codes(x8, qptr, (H0, W0))
torch.cuda.synchronize()
print("overflow", frz.overflowed())
