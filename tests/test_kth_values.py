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
    if kind == "mixed":                    # NaNs of either sign order last, as in torch.kthvalue
        xn = x.clone()
        xn[3::5003] = float("nan")
        xn[4::7001] = -float("nan")
        n_nan = int(torch.isnan(xn).sum())
        lo, hi = ops.kth_values(xn.cuda(), n - n_nan, n)
        assert lo.item() == torch.kthvalue(xn, n - n_nan).values.item() and torch.isnan(hi).item()


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


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4, 64, 16, 16), (64, 1024, 16, 16)])
def test_fused_stages_with_percentile_ranges_match_the_module_path(shape):
    """--act-percentile (quant_modules.py:203-210): the fused stage schedule with CDN_X_ACT_PERCENTILE against the same
    stages module by module (each QuantAct's order statistics from cdn_kth_values / torch.kthvalue), three forwards so
    that the EMA has history.  The ranges must agree to rounding (they are order statistics of tensors that agree to
    a code flip), the outputs to the usual flip noise."""
    import copy
    from codenet_amd import pipeline
    from codenet_amd.portable_quantizer.quant_modules import QuantAct
    Nb, C, H, W = shape
    planes = [C, 256, 128, 64] if C == 1024 else [C, 32, 16, 8]
    net = pipeline.build_hot_path(planes=planes).cuda().eval()
    acts = [m for m in net.modules() if isinstance(m, QuantAct)]
    for a in acts:
        a.percentile = True
    ref = copy.deepcopy(net)
    assert pipeline.FusedHotPath.supported(net.deconv_layers, shape)
    fused = pipeline.FusedHotPath(net.deconv_layers)
    g = torch.Generator().manual_seed(5)
    for it in range(3):
        x = (torch.randn(shape, generator=g) * (1.0 + 0.2 * it)).cuda()
        with torch.no_grad():
            y = fused(x).clone()
            yr = ref(x)
        for a, b in zip(acts, [m for m in ref.modules() if isinstance(m, QuantAct)]):
            lsb = (b.x_max - b.x_min).item() / 255
            assert abs(a.x_min.item() - b.x_min.item()) <= 0.51 * lsb and abs(a.x_max.item() - b.x_max.item()) <= 0.51 * lsb
            # and they are NOT the extremes: a percentile range is strictly inside min / max of a large tensor
        d = (y - yr).abs()
        std = yr.std().item()
        assert d.mean().item() < 0.02 * std and (d > 0.25 * std).float().mean().item() < 1e-3, (it, d.mean().item(), std)
    # the same network with min / max ranges ends up with wider ranges: the flag really changes the statistic
    net2 = pipeline.build_hot_path(planes=planes).cuda().eval()
    f2 = pipeline.FusedHotPath(net2.deconv_layers)
    with torch.no_grad():
        f2(x)
    a_pct = acts[1]
    a_mm = [m for m in net2.modules() if isinstance(m, QuantAct)][1]
    assert (a_mm.x_max - a_mm.x_min).item() > 0


@pytest.mark.gpu
def test_act_percentile_model_runs_its_stages_fused():
    from codenet_amd import harness
    model = harness.create_model(quantize=True, act_percentile=True).cuda().eval()
    import copy
    m2 = copy.deepcopy(model).enable_fused()
    # (every QuantAct input needs >= 500 elements: below that the 0.1 % rank is 0 and torch.kthvalue -- the reference --
    # raises; the smallest tensor here is stage 0's scale plane, 8 x 16 x 16)
    x = torch.randn(8, 3, 512, 512, generator=torch.Generator().manual_seed(3)).cuda()
    with torch.no_grad():
        a = model(x)[-1]
        b = m2(x)[-1]
    assert m2._fpath is not None and m2._fheads is None          # stages fused, heads / backbone module by module
    for k in a:
        std = a[k].std().item() + 1e-6
        d = (a[k] - b[k]).abs()
        assert d.mean().item() < 0.12 * std, (k, d.mean().item(), std)
