"""GPU parity of the backbone building blocks against plain PyTorch fp32
references of the same ops, then of the whole fused backbone against the module path."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ws(lib, dev):
    aux = lib.cdn_codenet_aux_workspace_bytes()
    ws = torch.zeros(aux // 4 + 64, device=dev)
    ptr = (ws.data_ptr() + 255) // 256 * 256
    return ws, ptr, (ws.numel() * 4 - (ptr - ws.data_ptr())) // 256 * 256


@pytest.mark.parametrize("N,C,H,W,up,stride,ld", [
    (2, 58, 12, 10, 0, 1, 60), (2, 58, 12, 10, 0, 2, 60), (1, 24, 9, 7, 0, 2, 24), (2, 116, 8, 8, 0, 1, 116),
    (2, 16, 6, 6, 1, 1, 16), (1, 232, 5, 5, 0, 2, 232)])
def test_dw3x3_strides_and_row_strides(N, C, H, W, up, stride, ld):
    from codenet_amd import _native as N_, ops
    lib, dev = N_.lib(), torch.device("cuda", 0)
    ws, wp, wb = _ws(lib, dev)
    g = torch.Generator().manual_seed(C * 7 + stride)
    x = torch.randn(N, C, H, W, generator=g).to(dev)
    a = torch.full((N, H * W, ld), 3.0, device=dev)              # padding channels: finite garbage
    a[:, :, :C] = x.permute(0, 2, 3, 1).reshape(N, H * W, C)
    w = torch.randn(C, 1, 3, 3, generator=g).to(dev)
    b = torch.randn(C, generator=g).to(dev)
    xu = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    ref = F.conv2d(xu, w, b, stride, 1, 1, C)
    Ho, Wo = ref.shape[2:]
    out = torch.full((N, Ho * Wo, ld), -7.0, device=dev)
    xmin, xmax, st = torch.zeros(1, device=dev), torch.zeros(1, device=dev), ops.quantact_state(dev)
    rc = lib.cdn_codenet_dw3x3_nhwc_forward(
        a.data_ptr(), None, N, C, H, W, up, stride, ld, ld, w.data_ptr(), b.data_ptr(), None, None, 0,
        xmin.data_ptr(), xmax.data_ptr(), st.data_ptr(), 8, 0.99, 1, wp, wb, out.data_ptr(),
        torch.cuda.current_stream().cuda_stream)
    N_.check(rc, "dw3")
    got = out[:, :, :C].reshape(N, Ho, Wo, C).permute(0, 3, 1, 2)
    assert (got - ref).abs().max().item() < 1e-5
    # the range covers the real channels only (first call: += initialisation)
    assert abs(xmin.item() - ref.min().item()) < 1e-5 and abs(xmax.item() - ref.max().item()) < 1e-5


def test_pointwise_row_strides_and_interleave():
    from codenet_amd import _native as N_
    lib, dev = N_.lib(), torch.device("cuda", 0)
    g = torch.Generator().manual_seed(3)
    M, C, h = 300, 116, 58
    y = torch.randn(M, C, generator=g).to(dev)                  # a unit input: x1 = y[:, :h], x2 = y[:, h:]
    w = torch.randn(h, h, generator=g).to(dev)
    b = torch.randn(h, generator=g).to(dev)
    t = torch.zeros(M, 60, device=dev)                           # padded intermediate (ld 60)
    x2 = y[:, h:]
    rc = lib.cdn_codenet_pointwise_nhwc_forward(
        x2.data_ptr(), None, M, h, h, C, 60, w.data_ptr(), None, None, None, b.data_ptr(), None, None, 1,
        None, None, None, 8, 0.99, 0, None, 0, t.data_ptr(), torch.cuda.current_stream().cuda_stream)
    N_.check(rc, "pw")
    ref = torch.relu(x2 @ w.t() + b)
    assert (t[:, :h] - ref).abs().max().item() < 1e-4 and t[:, h:].abs().max().item() == 0.0
    out = torch.zeros(M, C, device=dev)
    rc = lib.cdn_codenet_interleave_forward(y.data_ptr(), C, None, t.data_ptr(), 60, None, M, h,
                                            out.data_ptr(), C, torch.cuda.current_stream().cuda_stream)
    N_.check(rc, "interleave")
    from codenet_amd.portable_quantizer.quant_modules import channel_shuffle
    exp = channel_shuffle(torch.cat((y[:, :h], ref), 1).view(M, C, 1, 1), 2).view(M, C)
    assert (out - exp).abs().max().item() < 1e-4


@pytest.mark.parametrize("M,K,Co,affine,lda,ldo", [
    (2048, 1024, 256, True, 0, 0),       # cfg2 stage 0 (32 images at 8 x 8): 256 workgroups x 4 k slices of 256
    (8192, 256, 128, True, 0, 0),        # cfg2 stage 1
    (32768, 128, 64, False, 0, 0),       # cfg2 stage 2: one 32-k window per wave
    (2001, 512, 200, True, 520, 208),    # ragged m / co tiles (clamped loads, masked stores), row strides
    (77, 384, 40, False, 0, 0),          # fewer than 8 m tiles: the plain part of the tile map; odd window count
])
def test_pointwise_f32_wave_split_k(M, K, Co, affine, lda, ldo):
    """pws_kernel (round 5): the f32 pointwise for launches with few output tiles -- the four waves of a workgroup split K
    and stream their MFMA operands from global memory, partial tiles added in a fixed order -- against float64
    arithmetic, incl. bias, BN affine, ReLU, the range epilogue, and bitwise reproducibility from call to call."""
    from codenet_amd import _native as N_, ops
    lib, dev = N_.lib(), torch.device("cuda", 0)
    ws, wp, wb = _ws(lib, dev)
    g = torch.Generator().manual_seed(M + K)
    lda, ldo = lda or K, ldo or Co
    a = torch.randn(M, lda, generator=g).to(dev)
    w = (torch.randn(Co, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(Co, generator=g).to(dev)
    es = (torch.rand(Co, generator=g) + 0.5).to(dev) if affine else None
    eh = torch.randn(Co, generator=g).to(dev) if affine else None
    ref = a[:, :K].double() @ w.double().t() + b.double()
    if affine:
        ref = ref * es.double() + eh.double()
    ref = torch.relu(ref)
    xmin, xmax, st = torch.zeros(1, device=dev), torch.zeros(1, device=dev), ops.quantact_state(dev)
    outs = []
    for _ in range(2):
        out = torch.full((M, ldo), -7.0, device=dev)
        xmin.zero_()
        xmax.zero_()
        rc = lib.cdn_codenet_pointwise_nhwc_forward(
            a.data_ptr(), None, M, K, Co, lda, ldo, w.data_ptr(), None, None, None, b.data_ptr(),
            es.data_ptr() if affine else None, eh.data_ptr() if affine else None, 1,
            xmin.data_ptr(), xmax.data_ptr(), st.data_ptr(), 8, 0.99, 1, wp, wb, out.data_ptr(),
            torch.cuda.current_stream().cuda_stream)
        N_.check(rc, "pw")
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    got = outs[0][:, :Co].double()
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() < 2e-6 * scale + 1e-6
    if ldo > Co:
        assert (outs[0][:, Co:] == -7.0).all()
    assert xmin.item() == got.min().item() and xmax.item() == got.max().item()      # "+=" initialisation from (0, 0)


@pytest.mark.parametrize("stride,R", [(4, 64), (2, 50), (4, 37)])
def test_stem_conv(stride, R):
    from codenet_amd import _native as N_, ops
    lib, dev = N_.lib(), torch.device("cuda", 0)
    ws, wp, wb = _ws(lib, dev)
    g = torch.Generator().manual_seed(R)
    img = torch.randn(2, 3, R, R, generator=g).to(dev)
    w = (torch.randn(24, 3, 3, 3, generator=g) * 0.3).to(dev)
    b = torch.randn(24, generator=g).to(dev)
    ref = torch.relu(F.conv2d(img, w, b, stride, 1))
    Ho, Wo = ref.shape[2:]
    out = torch.empty(2, Ho * Wo, 24, device=dev)
    xmin, xmax, st = torch.zeros(1, device=dev), torch.zeros(1, device=dev), ops.quantact_state(dev)
    rc = lib.cdn_codenet_stem_forward(img.data_ptr(), 2, R, R, 24, stride, w.data_ptr(), b.data_ptr(), 1,
                                      xmin.data_ptr(), xmax.data_ptr(), st.data_ptr(), 8, 0.99, 1, wp, wb,
                                      out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    N_.check(rc, "stem")
    got = out.view(2, Ho, Wo, 24).permute(0, 3, 1, 2)
    assert (got - ref).abs().max().item() < 1e-5
    assert abs(xmax.item() - ref.max().item()) < 1e-5 and xmin.item() == 0.0


@pytest.mark.parametrize("res,batch,shuffle_free", [(64, 2, True), (96, 3, True), (64, 2, False)])
def test_fused_backbone_matches_modules(res, batch, shuffle_free):
    """FusedBackbone (stem, 16 ShuffleNetV2 units, layer4 on the HIP kernels) vs the mirrored modules on
    PyTorch-ROCm over 3 forwards: every QuantAct range, and the layer4 output after its QuantAct."""
    import copy
    from codenet_amd import harness, pipeline
    model = harness.create_model(quantize=True)
    ma, mb = copy.deepcopy(model).cuda(), copy.deepcopy(model).cuda()
    assert pipeline.FusedBackbone.supported(mb)
    fb = pipeline.FusedBackbone(mb, shuffle_free=shuffle_free)
    g = torch.Generator().manual_seed(res)
    for it in range(3):
        x = (torch.randn(batch, 3, res, res, generator=g) * (1.0 + 0.3 * it)).cuda()
        with torch.no_grad():
            ref = ma.layer4(ma.layer3(ma.layer2(ma.layer1(ma.layer0(x)))))
        feat, fq, hw = fb(x)
        act4 = mb.layer4[1][1]
        from codenet_amd import ops
        got, _ = ops.quantact_forward(feat.clone(), act4.x_min.clone(), act4.x_max.clone(),
                                      act4._device_state(x.device).clone(), bits=8, momentum=0.99, running=False)
        got = got.view(batch, hw[0], hw[1], -1).permute(0, 3, 1, 2)
        assert got.shape == ref.shape
        diff = (got - ref).abs()
        lsb = (act4.x_max - act4.x_min).item() / 255.0
        # The int8 pointwise sums are exact integers, the module path's fp32 convs round: pre-quantisation
        # values differ at 1e-7 relative, so about one value per 1e5 crosses a rounding boundary.  Such a
        # one-LSB code flip is re-amplified by every following unit (observed: one flip in layer1 unit 3
        # -> 11 % of the layer4 outputs off by more than 1.5 LSB, ranges off by 3e-4 relative; with no
        # flip -- e.g. seed 64, 128x128 -- all 40 ranges agree to 6e-7 and the output is identical).
        # Bounded, not exact; the whole-model test in test_harness.py checks the end result against the
        # reference model at the reference's own re-association floor.
        assert diff.max().item() < 16 * lsb + 1e-4, (diff.max().item(), lsb)
        assert diff.mean().item() < 0.6 * lsb
    acts_a = [m for m in ma.modules() if hasattr(m, "x_min") and isinstance(m.x_min, torch.Tensor)]
    acts_b = [m for m in mb.modules() if hasattr(m, "x_min") and isinstance(m.x_min, torch.Tensor)]
    checked = 0
    for a, b in zip(acts_a, acts_b):
        if a.x_max.item() == 0.0 and a.x_min.item() == 0.0:
            continue                      # heads / deform stages were not run
        tol = 3e-2 * (a.x_max.abs().item() + a.x_min.abs().item()) + 1e-4
        assert (a.x_min - b.x_min).abs().item() < tol and (a.x_max - b.x_max).abs().item() < tol
        checked += 1
    assert checked >= 16 * 2 + 3


@pytest.mark.parametrize("shuffle_free", [False, True])
def test_fused_units_match_reference_golden(shuffle_free):
    """Two chained ShuffleNetV2 units (stride 2, stride 1; shared block-output QuantAct) on the HIP kernels
    against the reference's own QuantBaseNode outputs over 3 forwards (tests/golden/base_nodes.npz)."""
    import numpy as np
    import os
    import torch.nn as nn
    from codenet_amd import pipeline
    from codenet_amd.portable_quantizer import quant_modules as qm
    z = {k: torch.from_numpy(v) for k, v in
         np.load(os.path.join(os.path.dirname(__file__), "golden", "base_nodes.npz")).items()}

    def bn(t):
        b = nn.BatchNorm2d(t.shape[1])
        b.weight.data, b.bias.data, b.running_mean, b.running_var = t[0].clone(), t[1].clone(), t[2].clone(), t[3].clone()
        return b

    def conv(w, s=1, groups=1):
        c = nn.Conv2d(w.shape[1] * groups, w.shape[0], w.shape[2], s, w.shape[2] // 2, groups=groups, bias=False)
        c.weight.data = w.clone()
        return c

    class Node(nn.Module):
        def __init__(self, u, stride):
            super().__init__()
            self.stride = stride
            h = z["u%d_w1" % u].shape[0]
            self.b2 = nn.Sequential(conv(z["u%d_w1" % u]), bn(z["u%d_bn1" % u]), nn.ReLU(inplace=True),
                                    conv(z["u%d_w2" % u], stride, h), bn(z["u%d_bn2" % u]),
                                    conv(z["u%d_w3" % u]), bn(z["u%d_bn3" % u]), nn.ReLU(inplace=True))
            if stride == 2:
                inp = z["u%d_w4" % u].shape[0]
                self.b1 = nn.Sequential(conv(z["u%d_w4" % u], 2, inp), bn(z["u%d_bn4" % u]),
                                        conv(z["u%d_w5" % u]), bn(z["u%d_bn5" % u]), nn.ReLU(inplace=True))
    shared = qm.QuantAct(8, quant_mode="asymmetric")
    nodes = []
    for u, stride in ((0, 2), (1, 1)):
        q = qm.QuantBaseNode(4, 8, act_percentile=False, wt_quant_mode="symmetric", act_quant_mode="asymmetric",
                             per_channel=True, weight_percentile=False)
        q.set_param(Node(u, stride).eval())
        q.set_act(shared)
        nodes.append(q.eval().cuda())
    fb = pipeline.FusedBackbone(None)
    assert fb.mixed_supported(nodes) or not shuffle_free
    for it in range(3):
        x = z["x%d" % it].cuda()
        Nb, C, H, W = x.shape
        a = x.permute(0, 2, 3, 1).contiguous().view(-1, C)
        if shuffle_free:
            lay = fb.run_units_mixed(nodes, a, C, None, Nb, H, W)
            y, Co, Ho, Wo = fb.materialize(lay), lay["C"], lay["H"], lay["W"]
        else:
            y, Co, Ho, Wo = fb.run_units(nodes, a, C, None, Nb, H, W)
        got = y.view(Nb, Ho, Wo, Co).permute(0, 3, 1, 2).cpu()
        sh = nodes[0].quant_act
        assert (sh.x_min.cpu() - z["shared_min%d" % it]).abs().item() < 1e-4
        assert (sh.x_max.cpu() - z["shared_max%d" % it]).abs().item() < 1e-4
        for u, names in ((0, ("quant_act1", "quant_act2", "quant_act4")), (1, ("quant_act1", "quant_act2"))):
            for k in names:
                a_ = getattr(nodes[u], k)
                assert (a_.x_max.cpu() - z["u%d_%s_max%d" % (u, k, it)]).abs().item() < 1e-4
                assert (a_.x_min.cpu() - z["u%d_%s_min%d" % (u, k, it)]).abs().item() < 1e-4
        lsb = (sh.x_max - sh.x_min).item() / 255.0
        d = (got - z["y1_%d" % it]).abs()
        assert d.max().item() <= 2.05 * lsb and (d > 1e-5).float().mean().item() < 0.02


@pytest.mark.parametrize("seed", range(6))
def test_dw3x3_random_shapes_with_quant_on_load(seed):
    """Random shapes / strides / paddings of the row strides, WITH fake-quantisation on load, against
    conv2d on the fake-quantised input (cdn_quantact_forward gives the same expression)."""
    import random
    from codenet_amd import _native as N_, ops
    lib, dev = N_.lib(), torch.device("cuda", 0)
    ws, wp, wb = _ws(lib, dev)
    rnd = random.Random(seed)
    N, C = rnd.randint(1, 3), rnd.choice([4, 12, 20, 58, 64, 100])
    H, W = rnd.randint(3, 20), rnd.randint(3, 20)
    stride = rnd.choice([1, 2])
    up = rnd.choice([0, 1]) if stride == 1 else 0
    ld = (C + 3) // 4 * 4 + rnd.choice([0, 4])
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C, H, W, generator=g).to(dev) * 2
    xmin, xmax, st = torch.zeros(1, device=dev), torch.zeros(1, device=dev), ops.quantact_state(dev)
    xq, _ = ops.quantact_forward(x, xmin, xmax, st)                      # tracks the range, returns fq(x)
    a = torch.zeros(N, H * W, ld, device=dev)
    a[:, :, :C] = x.permute(0, 2, 3, 1).reshape(N, H * W, C)
    w = torch.randn(C, 1, 3, 3, generator=g).to(dev)
    b = torch.randn(C, generator=g).to(dev)
    xu = F.interpolate(xq, scale_factor=2, mode="nearest") if up else xq
    ref = torch.relu(F.conv2d(xu, w, b, stride, 1, 1, C))
    Ho, Wo = ref.shape[2:]
    out = torch.zeros(N, Ho * Wo, ld, device=dev)
    rc = lib.cdn_codenet_dw3x3_nhwc_forward(
        a.data_ptr(), st.data_ptr(), N, C, H, W, up, stride, ld, ld, w.data_ptr(), b.data_ptr(), None, None, 1,
        None, None, None, 8, 0.99, 0, None, 0, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    N_.check(rc, "dw3")
    got = out[:, :, :C].reshape(N, Ho, Wo, C).permute(0, 3, 1, 2)
    assert (got - ref).abs().max().item() < 2e-5 * (1 + ref.abs().max().item())


def test_range_epilogue_stress_exact_extremes():
    """The producers' range epilogue (workgroup extremes + arrival ticket on one group line, last arriver
    reduces) must see EVERY workgroup's extremes: identity depthwise pass over planes whose minimum and
    maximum sit in single random elements, back-to-back launches, batch extremes compared bit for bit."""
    from codenet_amd import _native as N_, ops
    lib, dev = N_.lib(), torch.device("cuda", 0)
    ws, wp, wb = _ws(lib, dev)
    g = torch.Generator().manual_seed(11)
    shapes = [(8, 64, 64, 64), (2, 116, 40, 40), (1, 32, 7, 9), (16, 24, 48, 48)]
    stream = torch.cuda.current_stream().cuda_stream
    got, exp = [], []
    for it in range(240):
        N, C, H, W = shapes[it % len(shapes)]
        x = torch.rand(N, H * W, C, generator=g)
        flat = x.view(-1)
        i0, i1 = torch.randint(0, flat.numel(), (2,), generator=g).tolist()
        flat[i0], flat[i1] = -3.0 - it, 5.0 + it
        a = x.to(dev)
        w = torch.zeros(C, 1, 3, 3)
        w[:, 0, 1, 1] = 1.0
        w = w.to(dev)
        b = torch.zeros(C, device=dev)
        out = torch.empty_like(a)
        xmin, xmax, st = torch.zeros(1, device=dev), torch.zeros(1, device=dev), ops.quantact_state(dev)
        rc = lib.cdn_codenet_dw3x3_nhwc_forward(
            a.data_ptr(), None, N, C, H, W, 0, 1, C, C, w.data_ptr(), b.data_ptr(), None, None, 0,
            xmin.data_ptr(), xmax.data_ptr(), st.data_ptr(), 8, 0.99, 1, wp, wb, out.data_ptr(), stream)
        N_.check(rc, "dw3")
        got.append((xmin, xmax, st, out, a))
        exp.append((min(-3.0 - it, flat.min().item()), 5.0 + it))
    torch.cuda.synchronize()
    for (xmin, xmax, st, out, a), (lo, hi) in zip(got, exp):
        assert torch.equal(out, a)
        assert xmin.item() == lo and xmax.item() == hi
        sf = st.view(torch.float32)
        assert sf[4].item() == lo and sf[5].item() == hi
    # the arrival lines are left zeroed for the next producer
    assert int((ws.view(torch.int32) != 0).sum().item()) == 0


@pytest.mark.parametrize("M,C,Co,lda,off,ldo,relu", [
    (4099, 58, 58, 116, 58, 60, 1), (1500, 116, 116, 232, 116, 116, 1), (700, 232, 232, 464, 232, 232, 1),
    (300, 464, 1024, 464, 0, 1024, 1), (257, 24, 58, 24, 0, 60, 0), (130, 37, 70, 37, 0, 70, 1),
    (16384, 464, 1024, 464, 0, 1024, 1), (40000, 80, 58, 80, 0, 58, 1)])
def test_pointwise_bf16_split_is_exact_product(M, C, Co, lda, off, ldo, relu):
    """Final-valued input + 4-bit weight codes runs on the bf16 x 3 split kernel: products are exact, so the
    result matches a float64 reference to fp32 accumulation rounding (and the f32-MFMA kernel).
    (16384 x 464 -> 1024 = layer4 at batch 64, 512 x 512, and 40000 x 80: an ODD number of 32-channel windows with
    more than one 32-row block per wave of the persistent streaming kernel -- the prefetch buffers of pwd3_kernel
    used to come out swapped from the second block on; round 3.)"""
    from codenet_amd import _native as N_, ops
    lib, dev = N_.lib(), torch.device("cuda", 0)
    ws, wp, wb = _ws(lib, dev)
    g = torch.Generator().manual_seed(M + C)
    full = (torch.randn(M, lda, generator=g) * 3.0).to(dev)
    a = full[:, off:off + C]
    cpad = (C + 63) // 64 * 64
    q = torch.randint(-8, 8, (Co, C), generator=g)
    codes = torch.zeros(Co, cpad, dtype=torch.int8)
    codes[:, :C] = q.to(torch.int8)
    codes = codes.to(dev)
    scale = (torch.rand(Co, generator=g) * 20 + 1).to(dev)
    bias = torch.randn(Co, generator=g).to(dev)
    w = (q.to(dev).float() / scale[:, None]).contiguous()
    colsum = q.sum(1).to(torch.int32).to(dev)
    outs = []
    for use_codes in (True, False):
        out = torch.full((M, ldo), -7.0, device=dev)
        xmin, xmax, st = torch.zeros(1, device=dev), torch.zeros(1, device=dev), ops.quantact_state(dev)
        rc = lib.cdn_codenet_pointwise_nhwc_forward(
            a.data_ptr(), None, M, C, Co, lda, ldo, w.data_ptr(), codes.data_ptr() if use_codes else None,
            scale.data_ptr() if use_codes else None, colsum.data_ptr() if use_codes else None, bias.data_ptr(),
            None, None, relu, xmin.data_ptr(), xmax.data_ptr(), st.data_ptr(), 8, 0.99, 1, wp, wb,
            out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        N_.check(rc, "pw")
        outs.append((out, xmin.item(), xmax.item()))
    ref = a.double() @ (q.to(dev).double() / scale.double()[:, None]).t() + bias.double()
    ref = torch.relu(ref) if relu else ref
    mag = (a.double().abs() @ (q.to(dev).double().abs() / scale.double()[:, None]).t()).max().item()
    for out, lo, hi in outs:
        got = out[:, :Co].double()
        assert (got - ref).abs().max().item() < 2e-6 * mag
        assert (out[:, Co:] == -7.0).all()
        assert lo == out[:, :Co].min().item() and hi == out[:, :Co].max().item()
    assert (outs[0][0][:, :Co] - outs[1][0][:, :Co]).abs().max().item() < 2e-6 * mag


def _states(dev, ngen, g):
    """An array of QuantAct states (scale, zero point at words 2, 3) with different grids."""
    S = torch.zeros(ngen, 8)
    S[:, 2] = torch.rand(ngen, generator=g) * 40 + 5
    S[:, 3] = torch.round(torch.rand(ngen, generator=g) * 200 - 100)
    return S.to(dev)


def _avoid_ties(x, sc, zp):
    """Nudge the values whose code sc*x - zp lies within 1e-3 of a rounding tie (x.5): there the fp32 rounding
    of the product decides the code, and a torch composition is no reference for a single kernel."""
    t = sc.double() * x.double() - zp.double()
    near = ((t - torch.floor(t)) - 0.5).abs() < 1e-3
    return torch.where(near, x + 0.05 / sc, x)


@pytest.mark.parametrize("seed", range(12))
def test_depthwise_mixed_generations_random_shapes(seed):
    """Row-streaming depthwise kernels (x strips: C <= 128; channel chunks: wider) with per-channel QuantAct
    generations applied on load, stride 1 / 2, odd sizes, padded rows -- vs conv2d on the per-channel
    fake-quantised input."""
    import random
    from codenet_amd import _native as N_, ops
    lib, dev = N_.lib(), torch.device("cuda", 0)
    ws, wp, wb = _ws(lib, dev)
    rnd = random.Random(seed)
    g = torch.Generator().manual_seed(100 + seed)
    N = rnd.choice([1, 2, 3])
    C = rnd.choice([3, 8, 24, 58, 116, 130, 232, 260])
    H, W = rnd.choice([5, 8, 16, 33, 64]), rnd.choice([4, 9, 16, 31, 64, 130])
    stride = rnd.choice([1, 2])
    ld_in = (C + 3) // 4 * 4 + rnd.choice([0, 4])
    ld_out = (C + 3) // 4 * 4 + rnd.choice([0, 4])
    ngen = rnd.choice([1, 3, 7])
    S = _states(dev, ngen, g)
    gen = torch.randint(0, ngen, (C,), generator=g).to(torch.uint8).to(dev)
    x = (torch.randn(N, C, H, W, generator=g) * 2).to(dev)
    sc, zp = S[gen.long(), 2].view(1, C, 1, 1), S[gen.long(), 3].view(1, C, 1, 1)
    x = _avoid_ties(x, sc, zp)
    xq = (torch.round(sc * x - zp) + zp) / sc
    a = torch.full((N, H * W, ld_in), 1.5, device=dev)
    a[:, :, :C] = x.permute(0, 2, 3, 1).reshape(N, H * W, C)
    w = torch.randn(C, 1, 3, 3, generator=g).to(dev)
    b = torch.randn(C, generator=g).to(dev)
    ref = torch.relu(F.conv2d(xq, w, b, stride, 1, 1, C))
    Ho, Wo = ref.shape[2:]
    out = torch.full((N, Ho * Wo, ld_out), -7.0, device=dev)
    xmin, xmax, st = torch.zeros(1, device=dev), torch.zeros(1, device=dev), ops.quantact_state(dev)
    rc = lib.cdn_codenet_dw3x3_mixed_forward(
        a.data_ptr(), S.data_ptr(), gen.data_ptr(), N, C, H, W, 0, stride, ld_in, ld_out, w.data_ptr(),
        b.data_ptr(), None, None, 1, xmin.data_ptr(), xmax.data_ptr(), st.data_ptr(), 8, 0.99, 1, wp, wb,
        out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    N_.check(rc, "dw mixed")
    got = out[:, :, :C].reshape(N, Ho, Wo, C).permute(0, 3, 1, 2)
    tol = 2e-5 * (1 + ref.abs().max().item())
    assert (got - ref).abs().max().item() < tol
    assert (out[:, :, C:] == -7.0).all() or ld_out % 4 == 0          # padding is written only by whole-quad stores
    assert abs(xmin.item() - got.min().item()) < 1e-6 and abs(xmax.item() - got.max().item()) < 1e-6


@pytest.mark.parametrize("seed", range(10))
def test_pointwise_mixed_generations_and_output_map(seed):
    """Mixed-generation pointwise (streaming and LDS-tiled bf16-split kernels) with an output channel map
    vs a float64 matmul on the per-channel fake-quantised input; untouched output slots stay untouched."""
    import random
    from codenet_amd import _native as N_, ops
    lib, dev = N_.lib(), torch.device("cuda", 0)
    ws, wp, wb = _ws(lib, dev)
    rnd = random.Random(seed)
    g = torch.Generator().manual_seed(200 + seed)
    M = rnd.choice([1, 31, 129, 1000, 4100])
    C = rnd.choice([4, 24, 58, 116, 232, 464])
    Co = rnd.choice([1, 2, 29, 58, 116, 130, 300])
    lda = C + rnd.choice([0, 4, 6]) if C % 4 == 0 else C + 3
    ldo = Co * 2 + rnd.choice([0, 1])
    ngen = rnd.choice([1, 2, 9])
    S = _states(dev, ngen, g)
    gen = torch.randint(0, ngen, (C,), generator=g).to(torch.uint8).to(dev)
    a = (torch.randn(M, lda, generator=g) * 2).to(dev)
    sc, zp = S[gen.long(), 2].view(1, C), S[gen.long(), 3].view(1, C)
    a[:, :C] = _avoid_ties(a[:, :C], sc, zp)
    aq = (torch.round(sc * a[:, :C] - zp) + zp) / sc
    cpad = (C + 63) // 64 * 64
    q = torch.randint(-8, 8, (Co, C), generator=g)
    q[:, torch.rand(C, generator=g) < 0.4] = 0                       # pass-through columns
    codes = torch.zeros(Co, cpad, dtype=torch.int8)
    codes[:, :C] = q.to(torch.int8)
    codes = codes.to(dev)
    scale = (torch.rand(Co, generator=g) * 20 + 1).to(dev)
    bias = torch.randn(Co, generator=g).to(dev)
    wf = (q.to(dev).float() / scale[:, None]).contiguous()
    colsum = q.sum(1).to(torch.int32).to(dev)
    omap = torch.randperm(ldo, generator=g)[:Co].to(torch.int32).to(dev)
    out = torch.full((M, ldo), -7.0, device=dev)
    xmin, xmax, st = torch.zeros(1, device=dev), torch.zeros(1, device=dev), ops.quantact_state(dev)
    rc = lib.cdn_codenet_pointwise_mixed_forward(
        a.data_ptr(), S.data_ptr(), gen.data_ptr(), M, C, Co, lda, ldo, wf.data_ptr(), codes.data_ptr(),
        scale.data_ptr(), colsum.data_ptr(), bias.data_ptr(), None, None, 1, omap.data_ptr(), xmin.data_ptr(),
        xmax.data_ptr(), st.data_ptr(), 8, 0.99, 1, wp, wb, out.data_ptr(),
        torch.cuda.current_stream().cuda_stream)
    N_.check(rc, "pw mixed")
    ref = torch.relu(aq.double() @ (q.to(dev).double() / scale.double()[:, None]).t() + bias.double())
    mag = (aq.double().abs() @ (q.to(dev).double().abs() / scale.double()[:, None]).t()).max().item() + 1.0
    got = out[:, omap.long()].double()
    assert (got - ref).abs().max().item() < 3e-6 * mag
    untouched = torch.ones(ldo, dtype=torch.bool, device=dev)
    untouched[omap.long()] = False
    assert (out[:, untouched] == -7.0).all()
    assert xmin.item() == out[:, omap.long()].min().item() and xmax.item() == out[:, omap.long()].max().item()


@pytest.mark.parametrize("M,C,Co,relu", [(1000, 1024, 256, 1), (16384, 1024, 256, 1), (333, 512, 200, 0),
                                         (4096, 544, 256, 1), (700, 256, 128, 1), (5000, 128, 64, 1)])
def test_int8_pointwise_exact_integer_sums(M, C, Co, relu):
    """int8-MFMA pointwise on integer codes (one QuantAct state on the input) at the stage shapes: the integer
    sums are exact, so the result equals a float64 evaluation of
    sum_c L_c * qw / (s * sw) + b to fp32 rounding of the final scale."""
    from codenet_amd import _native as N_, ops
    lib, dev = N_.lib(), torch.device("cuda", 0)
    ws, wp, wb = _ws(lib, dev)
    g = torch.Generator().manual_seed(M + C + Co)
    a = (torch.randn(M, C, generator=g) * 2).to(dev)
    S = torch.zeros(8, device=dev)
    S[2], S[3] = 17.3, -41.0
    a = _avoid_ties(a, S[2], S[3])
    L = torch.round(S[2] * a - S[3]) + S[3]                      # integer levels
    cpad = (C + 63) // 64 * 64
    q = torch.randint(-8, 8, (Co, C), generator=g)
    codes = torch.zeros(Co, cpad, dtype=torch.int8)
    codes[:, :C] = q.to(torch.int8)
    codes = codes.to(dev)
    scale = (torch.rand(Co, generator=g) * 20 + 1).to(dev)
    bias = torch.randn(Co, generator=g).to(dev)
    wf = (q.to(dev).float() / scale[:, None]).contiguous()
    colsum = q.sum(1).to(torch.int32).to(dev)
    out = torch.empty(M, Co, device=dev)
    xmin, xmax, st = torch.zeros(1, device=dev), torch.zeros(1, device=dev), ops.quantact_state(dev)
    Si = S.view(torch.int32).clone()
    rc = lib.cdn_codenet_pointwise_nhwc_forward(
        a.data_ptr(), Si.data_ptr(), M, C, Co, 0, 0, wf.data_ptr(), codes.data_ptr(), scale.data_ptr(),
        colsum.data_ptr(), bias.data_ptr(), None, None, relu, xmin.data_ptr(), xmax.data_ptr(), st.data_ptr(), 8,
        0.99, 1, wp, wb, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    N_.check(rc, "pw int8")
    isum = L.double() @ q.to(dev).double().t()                   # exact integers in float64
    rinv = (1.0 / (S[2] * scale)).double()                       # fp32 product, fp32 reciprocal (as the kernel)
    ref = isum * rinv[None, :] + bias.double()
    ref = torch.relu(ref) if relu else ref
    err = (out.double() - ref).abs().max().item()
    assert err < 2e-6 * (ref.abs().max().item() + 1.0), err
    assert xmin.item() == out.min().item() and xmax.item() == out.max().item()




@pytest.mark.parametrize("res,batch,maxpool", [(64, 2, False), (128, 3, False), (256, 2, False), (128, 2, True),
                                               (256, 3, True)])
def test_frozen_backbone_on_byte_codes_matches_fp32_frozen_schedule(res, batch, maxpool):
    """pipeline.FrozenBackbone (stem, 16 units, layer4 on BYTE CODES: cdn_codenet_stem_q8 / dw3x3_q8 /
    pointwise_q8_strided, every QuantAct frozen) against FusedBackbone on the same model with running_stat False.
    The depthwise chains and the stem are the fp32 kernels' arithmetic on (q + zp) / scale; a unit's first 1x1 conv is an
    exact integer sum here where the running-range schedule accumulates exact products in fp32, so single codes can
    flip, and a flip early in this deep random-weight network moves many codes behind it.  The yardstick is the
    difference between FusedBackbone and the MODULE path (torch ops) on the same frozen model: the byte schedule must
    be at least that close to FusedBackbone.  The kernels themselves are pinned exactly in
    test_dw3x3_q8_is_the_fp32_kernel_on_code_values, test_gpu_frozen.py::test_stem_q8_* and *_strided_*.  No overflow
    (ranges covered by pipeline.cover_frozen_ranges); no range moves."""
    import copy
    from codenet_amd import harness, pipeline
    model = harness.create_model(quantize=True, maxpool=maxpool).cuda()      # maxpool: README configs b / e stems
    fb = pipeline.FusedBackbone(model)
    g = torch.Generator().manual_seed(res + batch)
    xs = [torch.randn(batch, 3, res, res, generator=g).cuda() for i in range(3)]
    for _ in range(20):
        for x in xs:
            fb(x)                                           # running ranges settle
    pipeline.set_running_stat(model, False)
    # the byte grid must hold every code: widen the EMA ranges over what the frozen network feeds each QuantAct
    assert pipeline.cover_frozen_ranges(model, xs, margin=0.05) > 0
    assert pipeline.FrozenBackbone.supported(model)
    ranges = [(a.x_min.item(), a.x_max.item()) for a in model.modules() if hasattr(a, "x_min") and isinstance(a.x_min, torch.Tensor)]
    fz = pipeline.FrozenBackbone(model)
    act4 = model.layer4[1][1]
    for x in xs:
        feat, fq, hw = fb(x)
        st = act4._device_state(x.device).view(torch.float32)
        want = torch.round(st[2] * feat - st[3])            # codes of the fp32 schedule's pre-quantisation output
        with torch.no_grad():                               # the module path (torch ops): the noise yardstick
            r = x
            for name in ("layer0", "layer1", "layer2", "layer3", "layer4"):
                r = getattr(model, name)(r)
        mod = torch.round(st[2] * r - st[3]).permute(0, 2, 3, 1).reshape(want.shape)
        got, gq, ghw = fz(x)
        assert got.dtype == torch.int8 and tuple(got.shape) == tuple(feat.shape) and ghw == hw and gq == fq
        assert not fz.overflowed()
        d, noise = (got.float() - want).abs(), (want - mod).abs()
        print("   res %d: %.2e of the layer4 codes differ, mean %.3f max %d LSB   (fp32 schedule vs module path: "
              "%.2e, mean %.3f max %d)" % (res, (d > 0).float().mean().item(), d.mean().item(), int(d.max().item()),
                                           (noise > 0).float().mean().item(), noise.mean().item(), int(noise.max().item())))
        # Frozen images are independent.  An image without a flipped code is bit-identical; a flipped code early in this
        # 50-layer random-weight network moves many codes behind it IN THAT IMAGE, by what two accepted implementations
        # of the reference (the fp32 schedule and the module path, printed above) differ by where they differ: a
        # fraction of a LSB on average, a dozen at most.  A plumbing error (slot map, strides, states) is far outside.
        for i in range(d.shape[0]):
            di = d[i]
            assert di.max().item() == 0 or (di.mean().item() <= 1.5 and di.max().item() <= 24), (i, di.mean().item())
    assert ranges == [(a.x_min.item(), a.x_max.item()) for a in model.modules()
                      if hasattr(a, "x_min") and isinstance(a.x_min, torch.Tensor)]


@pytest.mark.parametrize("N,C,H,W,stride,ld", [(2, 58, 9, 11, 1, 64), (1, 24, 16, 16, 2, 32), (3, 116, 7, 8, 2, 128),
                                                 (2, 232, 5, 5, 1, 240)])
def test_dw3x3_q8_is_the_fp32_kernel_on_code_values(N, C, H, W, stride, ld):
    """cdn_codenet_dw3x3_q8_forward (byte codes in, byte codes out) against the fp32 depthwise kernel fed with the
    values (q + zp) / scale and the same quantiser state (frozen), its output quantised with the output state:
    identical codes (same accumulation chain), padding channels ignored."""
    from codenet_amd import _native as N_, ops
    lib, dev = N_.lib(), torch.device("cuda", 0)
    ws, wp, wb = _ws(lib, dev)
    g = torch.Generator().manual_seed(C + H)
    codes = torch.randint(-128, 128, (N, H * W, ld), generator=g, dtype=torch.int32).to(torch.int8).to(dev)
    sa = ops.quantact_state(dev); sr = ops.quantact_state(dev)
    sa.view(torch.float32)[2], sa.view(torch.float32)[3] = 31.7, -17.0
    sr.view(torch.float32)[2], sr.view(torch.float32)[3] = 9.3, -101.0
    w = (torch.randn(C, 1, 3, 3, generator=g) * 0.4).to(dev)
    b = (torch.randn(C, generator=g) * 0.2).to(dev)
    Ho, Wo = ((H - 1) // 2 + 1, (W - 1) // 2 + 1) if stride == 2 else (H, W)
    out8 = torch.zeros(N, Ho * Wo, ld, dtype=torch.int8, device=dev)
    of = torch.zeros(1, dtype=torch.int32, device=dev)
    rc = lib.cdn_codenet_dw3x3_q8_forward(codes.data_ptr(), sa.data_ptr(), N, C, H, W, stride, ld, ld, w.data_ptr(),
                                          b.data_ptr(), 0, sr.data_ptr(), out8.data_ptr(), of.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream)
    N_.check(rc, "dw q8")
    # fp32 reference path: the same values through the fp32 kernel (no input quantiser: final values), then the code
    vals = ((codes.float() + sa.view(torch.float32)[3]) / sa.view(torch.float32)[2]).contiguous()
    out = torch.empty(N, Ho * Wo, ld, device=dev)
    rc = lib.cdn_codenet_dw3x3_nhwc_forward(vals.data_ptr(), None, N, C, H, W, 0, stride, ld, ld, w.data_ptr(),
                                            b.data_ptr(), None, None, 0, None, None, None, 8, 0.99, 0, wp, wb,
                                            out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    N_.check(rc, "dw fp32")
    s2, z2 = sr.view(torch.float32)[2], sr.view(torch.float32)[3]
    want = torch.round(s2 * out[:, :, :C] - z2)
    inside = (want >= -128) & (want <= 127)
    assert torch.equal(out8[:, :, :C].float()[inside], want[inside])
    assert bool(of.item()) == bool((~inside).any().item())


@pytest.mark.parametrize("res,batch,scale2", [(64, 2, 1.0), (128, 3, 1.0), (256, 4, 1.0), (512, 8, 1.0), (128, 2, 40.0)])
def test_layer1_first_conv_recomputed_in_the_depthwise_is_bit_identical(res, batch, scale2):
    """Round 4 (VERDICT r3 "next" #1b): the first unit of layer 1 no longer stores the output of its first 1x1 conv
    (58 channels at the stem's resolution, fp32: 243 MB at batch 64, 512 x 512) -- a range-only pass of the int8
    pointwise kernel fixes the QuantAct, pwdwx_kernel recomputes the conv (exact integer sums by v_dot4c_i32_i8, the
    same epilogue expression) inside the stride-2 depthwise.  Whole backbone with and without it over three forwards
    with moving ranges: the layer4 tensor and all 40 QuantAct ranges are bit-identical (odd and even row / column
    counts, several x strips per row at 512).  scale2 = 40: the second batch is 40x the first, so the stem's codes are
    too wide for the nibble split (state word [6]) -- both sides then take their fp32 branches (f32-MFMA there, an fp32
    FMA chain here): equal to fp32 rounding, not bit for bit."""
    import copy
    from codenet_amd import harness, pipeline
    model = harness.create_model(quantize=True)
    ma, mb = copy.deepcopy(model).cuda(), copy.deepcopy(model).cuda()
    fa, fb = pipeline.FusedBackbone(ma), pipeline.FusedBackbone(mb)
    fa.recompute_pw1 = False
    assert fb.recompute_pw1
    g = torch.Generator().manual_seed(res + batch)
    for it in range(3):
        x = (torch.randn(batch, 3, res, res, generator=g) * (scale2 if it == 1 else 1.0 + 0.3 * it)).cuda()
        ya = fa(x)[0].clone()
        yb = fb(x)[0].clone()
        if scale2 == 1.0:
            assert torch.equal(ya, yb), "forward %d" % it
        else:
            assert (ya - yb).abs().max().item() <= 2e-2 * ya.abs().max().item() + 1e-5, "forward %d" % it
    if scale2 == 1.0:
        for (na, ba), (nb, bb) in zip(ma.named_buffers(), mb.named_buffers()):
            if na.endswith(("x_min", "x_max")):
                assert torch.equal(ba, bb), na
