"""cdn_kth_values (exact radix select) against torch.kthvalue -- the order statistics --act-percentile / --wt-percentile
use as quantisation ranges (portable_quantizer/quantization_utils/quant_utils.py:18-30) -- and the QuantAct percentile
mode on top of it against the reference class's own outputs (`quant_ref.npz`: pact_*)."""
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.gpu
@pytest.mark.parametrize("n,kind", [(1, "randn"), (7, "randn"), (1000, "randn"), (4099, "levels"), (1 << 20, "randn"),
                                    ((1 << 22) + 3, "relu"), (1 << 21, "levels"), (300001, "mixed"), (65536, "const"),
                                    (1 << 20, "unaligned")])
def test_kth_values_equal_torch_kthvalue(n, kind):
    from codenet_amd import ops
    g = torch.Generator().manual_seed(n % 1000 + len(kind))
    if kind == "randn":
        x = torch.randn(n, generator=g) * 3
    elif kind == "levels":                 # 8-bit levels: thousands of duplicates of every value
        x = torch.round(torch.rand(n, generator=g) * 255) / 17 - 4
    elif kind == "relu":                   # half the tensor is exactly zero
        x = torch.relu(torch.randn(n, generator=g))
    elif kind == "mixed":                  # huge dynamic range, infinities, denormals, both zeros
        x = torch.randn(n, generator=g) * torch.exp(torch.randn(n, generator=g) * 8)
        x[::1001] = float("inf")
        x[5::1003] = -float("inf")
        x[7::997] = 1e-42
        x[9::991] = -0.0
    elif kind == "const":
        x = torch.full((n,), 2.5)
    else:                                  # a view that starts 4 bytes into an allocation: scalar tail path
        x = (torch.randn(n + 1, generator=g))[1:]
    xd = x.cuda() if kind != "unaligned" else torch.randn(n + 1, generator=torch.Generator().manual_seed(1)).cuda()[1:]
    if kind == "unaligned":
        x = xd.cpu()
    ks = sorted({1, n, max(1, round(n * 0.001)), max(1, round(n * 0.999)), (n + 1) // 2})
    for k_lo in ks:
        for k_hi in (ks[-1], ks[len(ks) // 2]):
            lo, hi = ops.kth_values(xd, k_lo, k_hi)
            ref_lo, ref_hi = torch.kthvalue(x, k_lo).values, torch.kthvalue(x, k_hi).values
            # (== : -0.0 and +0.0 are the same order statistic)
            assert lo.item() == ref_lo.item() and hi.item() == ref_hi.item(), (n, kind, k_lo, k_hi)
    with pytest.raises(RuntimeError):
        ops.kth_values(xd, 0, 1)


@pytest.mark.gpu
def test_quantact_percentile_mode_on_the_native_order_statistics_matches_the_reference_class():
    from codenet_amd.portable_quantizer.quant_modules import QuantAct
    from codenet_amd.portable_quantizer.quantization_utils import quant_utils
    z = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(G, "quant_ref.npz")).items()}
    act = QuantAct(8, quant_mode="asymmetric", percentile=True).cuda()
    for it in range(2):
        x = z["pact_x%d" % it].cuda()
        lo, hi = quant_utils.get_percentile_min_max(x.view(-1), 0.1, 99.9, output_tensor=True)
        rlo, rhi = quant_utils.get_percentile_min_max(x.cpu().view(-1), 0.1, 99.9, output_tensor=True)
        assert lo.item() == rlo.item() and hi.item() == rhi.item()
        y = act(x)
        assert torch.equal(act.x_min.cpu(), z["pact_min%d" % it]) and torch.equal(act.x_max.cpu(), z["pact_max%d" % it])
        assert torch.equal(y.cpu(), z["pact_y%d" % it])
