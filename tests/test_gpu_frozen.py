"""Frozen-range schedule with byte codes in HBM (codenet_frozen.hip, pipeline.FrozenHotPath) -- the serving mode
``QuantAct.running_stat = False`` (a plain attribute in the reference, quant_modules.py:172,181; the update
:203-219 is skipped):

  * BIT-IDENTICAL to the fp32 fused schedule (cdn_codenet_stage_fused_forward, running = 0) while no code leaves
    its 8-bit grid -- same scale sums, same bilinear arithmetic, exact integer pointwise sums, same epilogue;
  * a saturated code (the reference does not clamp, a byte must) raises the overflow flag;
  * against the CPU oracle with frozen ranges at the BASELINE stage shapes: the standard W4A8 acceptance;
  * the int8 pointwise kernel alone against an exact integer evaluation.
"""
import copy

import pytest
import torch
import torch.nn.functional as F

from oracle import quant as Q

pytestmark = pytest.mark.gpu


def _inputs(n, c, res, k, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(n, c, res, res, generator=g).abs_() * (1.66 * (1.0 - 0.04 * i)) for i in range(k)]


def _warm_and_freeze(net, xs, margin=0.5):
    """Ranges from running-mode passes over the inputs, widened by `margin` of their width on both sides (the
    gather output depends on the input through the sampling positions too, so a slightly different batch can
    exceed the extremes of the warm-up batches), then frozen."""
    from codenet_amd import pipeline
    from codenet_amd.portable_quantizer.quant_modules import QuantAct
    pipeline.set_running_stat(net, True)
    warm = pipeline.FusedHotPath(net.deconv_layers)
    for _ in range(2):
        for x in xs:
            warm(x.cuda())
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, QuantAct):
                w = (m.x_max - m.x_min) * margin
                m.x_min.sub_(w)
                m.x_max.add_(w)
    pipeline.set_running_stat(net, False)


@pytest.mark.parametrize("planes,res,n", [([64, 32, 16, 8], 8, 3), ([128, 64, 32, 16], 8, 2),
                                          ([1024, 256, 128, 64], 16, 4), ([1024, 256, 128, 64], 16, 64),
                                          ([1024, 256, 128, 64], 8, 32)])
def test_frozen_codes_bit_identical_to_fp32_frozen_schedule(planes, res, n):
    from codenet_amd import pipeline
    net = pipeline.build_hot_path(quantized=True, planes=planes, seed=41).cuda()
    xs = _inputs(n, planes[0], res, 3, 141)
    _warm_and_freeze(net, xs)
    ref = pipeline.FusedHotPath(net.deconv_layers)
    frz = pipeline.FrozenHotPath(net.deconv_layers)
    for x in xs:                                   # later inputs are slightly smaller: inside the frozen ranges
        xg = x.cuda()
        a = ref(xg).clone()
        b = frz(xg).clone()
        assert not frz.overflowed()
        assert torch.equal(a, b)
        # the byte codes themselves against the fp32 schedule's pre-quantisation output + its quantiser state
        r, rq, shape = ref.forward_nhwc(xg)
        r8, rq8, shape8 = frz.forward_codes(xg)
        assert rq == rq8 and r8.dtype == torch.int8 and tuple(r8.shape) == tuple(r.shape)
        act = list(net.deconv_layers)[-2][1]
        st = act._device_state(xg.device).view(torch.float32)
        scale, zp = st[2], st[3]
        assert torch.equal(r8.float(), torch.round(scale * r - zp))
    # graph replay of the byte-code schedule
    xbuf = xs[0].cuda()
    replay = frz.capture(xbuf)
    for x in xs:
        xbuf.copy_(x.cuda())
        got = replay().clone()
        assert torch.equal(got, frz.forward_codes(xbuf)[0])


def test_frozen_overflow_flag_when_codes_leave_the_grid():
    from codenet_amd import pipeline
    net = pipeline.build_hot_path(quantized=True, planes=[64, 32, 16, 8], seed=42).cuda()
    xs = _inputs(2, 64, 8, 1, 142)
    _warm_and_freeze(net, xs)
    frz = pipeline.FrozenHotPath(net.deconv_layers)
    frz(xs[0].cuda())
    assert not frz.overflowed()
    frz(xs[0].cuda() * 4.0)                         # far outside the frozen ranges
    assert frz.overflowed()
    assert not frz.overflowed()                     # reading resets the flag
    # each of the three quantisers of a stage on its own: shrink one range, everything else untouched
    for which in ("quant_identity_deform", "post"):
        net2 = copy.deepcopy(net)
        st0 = list(net2.deconv_layers)
        act = st0[0].quant_identity_deform if which == "quant_identity_deform" else st0[1][1]
        with torch.no_grad():
            mid, half = (act.x_max + act.x_min) / 2, (act.x_max - act.x_min) / 2
            act.x_min.copy_(mid - 0.4 * half)
            act.x_max.copy_(mid + 0.4 * half)
        f2 = pipeline.FrozenHotPath(net2.deconv_layers)
        f2(xs[0].cuda())
        assert f2.overflowed(), which


def test_frozen_nhwc_input_with_quantiser_state():
    """Stage 0 fed by a native backbone: channels-last pre-quantisation values + their QuantAct state."""
    from codenet_amd import pipeline
    from codenet_amd.portable_quantizer.quant_modules import QuantAct
    net = pipeline.build_hot_path(quantized=True, planes=[64, 32, 16, 8], seed=43).cuda()
    xs = _inputs(3, 64, 8, 2, 143)
    # the backbone's last QuantAct: run it once in running mode so that its state holds (scale, zp)
    act_in = QuantAct(8, quant_mode="asymmetric").cuda()
    xq = act_in(xs[0].cuda())                       # fake-quantised NCHW tensor: what a PyTorch backbone hands over
    _warm_and_freeze(net, [xq.cpu()])
    act_in.running_stat = False
    x_nhwc = xs[0].cuda().permute(0, 2, 3, 1).reshape(3, 64, 64).contiguous()
    qptr = act_in._device_state(x_nhwc.device).data_ptr()
    ref = pipeline.FusedHotPath(net.deconv_layers)
    frz = pipeline.FrozenHotPath(net.deconv_layers)
    a = ref.forward_nhwc(x_nhwc, qptr, (8, 8))[0].clone()
    r8, rq, _ = frz.forward_codes(x_nhwc, qptr, (8, 8))
    assert not frz.overflowed()
    st = list(net.deconv_layers)[-2][1]._device_state(x_nhwc.device).view(torch.float32)
    assert torch.equal(r8.float(), torch.round(st[2] * a - st[3]))
    b = frz.forward_nhwc(x_nhwc, qptr, (8, 8))[0]
    assert torch.equal(b, (torch.round(st[2] * a - st[3]) + st[3]) / st[2])


@pytest.mark.parametrize("planes,res,n", [([64, 32, 16, 8], 8, 3), ([1024, 256, 128, 64], 16, 8)])
def test_frozen_byte_code_input_equals_its_expanded_values(planes, res, n):
    """Stage 0 fed with BYTE CODES of the backbone's last QuantAct (x_kind 2: what a frozen byte-code backbone hands
    over; 1 byte per element instead of the 4-byte pre-quantisation values read twice) -- the codes of every stage
    must equal those computed from the fp32 channels-last input holding the same values (q + zp) / scale with the same
    state (fake-quantising them on load returns them unchanged)."""
    from codenet_amd import pipeline
    from codenet_amd.portable_quantizer.quant_modules import QuantAct
    net = pipeline.build_hot_path(quantized=True, planes=planes, seed=45).cuda()
    xs = _inputs(n, planes[0], res, 2, 145)
    act_in = QuantAct(8, quant_mode="asymmetric").cuda()
    xq = act_in(xs[0].cuda())
    _warm_and_freeze(net, [xq.cpu()])
    act_in.running_stat = False
    st = act_in._device_state(xq.device).view(torch.float32)
    x_nhwc = xq.permute(0, 2, 3, 1).reshape(n, res * res, planes[0]).contiguous()     # final values (q + zp) / scale
    codes = torch.round(st[2] * x_nhwc - st[3])
    assert codes.abs().max().item() <= 128
    x8 = codes.clamp(-128, 127).to(torch.int8).contiguous()
    qptr = act_in._device_state(xq.device).data_ptr()
    frz = pipeline.FrozenHotPath(net.deconv_layers)
    a = frz.forward_codes(x_nhwc, qptr, (res, res))[0].clone()
    b = frz.forward_codes(x8, qptr, (res, res))[0].clone()
    assert not frz.overflowed() and a.dtype == torch.int8 and torch.equal(a, b)
    replay = frz.capture(x8, x_qstate=qptr, hw=(res, res))
    assert torch.equal(replay(), a)


def test_frozen_w2_stage0_runs_on_fp32_schedule():
    """CoDeNet2x: C = 2153 is not a multiple of 4 -> stage 0 on the fp32 frozen schedule, stages 1-2 on codes."""
    from codenet_amd import pipeline
    net = pipeline.build_hot_path(quantized=True, planes=[2153, 256, 128, 64], seed=44).cuda()
    xs = _inputs(2, 2153, 8, 2, 144)
    _warm_and_freeze(net, xs)
    ref = pipeline.FusedHotPath(net.deconv_layers)
    frz = pipeline.FrozenHotPath(net.deconv_layers)
    for x in xs:
        assert torch.equal(ref(x.cuda()), frz(x.cuda()))
        assert not frz.overflowed()


@pytest.mark.parametrize("chain_scale", [False, True])
def test_frozen_matches_oracle_with_frozen_ranges_real_shapes(chain_scale):
    """cfg3 stage shapes against the CPU oracle with running = False (oracle/quant.py::stage_w4a8).  chain_scale: the
    declared variant in which stage k's pointwise epilogue accumulates the exact integer sums of stage k+1's scale
    prediction (one rounding of the exact sum instead of an fp32 sum of C products) -- same acceptance: a scale code
    that lands on the other side of a rounding boundary moves that pixel's outputs, like any other code flip."""
    from codenet_amd import pipeline
    planes, res, n = [1024, 256, 128, 64], 16, 4
    net = pipeline.build_hot_path(quantized=True, planes=planes, seed=45)
    xs = _inputs(n, planes[0], res, 2, 145)
    net = net.cuda()
    _warm_and_freeze(net, xs)
    net_cpu = copy.deepcopy(net).cpu()
    mods = list(net_cpu.deconv_layers)
    frz = pipeline.FrozenHotPath(net.deconv_layers, chain_scale=chain_scale)
    for x in xs:
        cur = x
        with torch.no_grad():
            for i in range(0, len(mods), 3):
                q, post = mods[i], mods[i + 1]
                bnm = q.quant_conv_channel_bn.bn
                bn = (bnm.weight, bnm.bias, bnm.running_mean, bnm.running_var, bnm.eps)
                acts = [Q.QuantActState(x_min=a.x_min.item(), x_max=a.x_max.item())
                        for a in (q.quant_act[1], q.quant_identity_deform, post[1])]
                r = Q.stage_w4a8(cur, q.quant_conv_scale.weight, q.quant_conv_scale.bias, q.quant_deform_conv.weight,
                                 q.quant_conv_channel_bn.conv.weight, bn, acts[0], acts[1], running=False)
                cur = F.interpolate(acts[2](torch.relu(r["y"]), running=False), scale_factor=2, mode="nearest")
        y = frz(x.cuda()).cpu()
        assert not frz.overflowed()
        last = list(net.deconv_layers)[-2][1]
        lsb = (last.x_max - last.x_min).item() / 255.0
        diff = (y - cur).abs()
        assert diff.max().item() <= 1.05 * lsb + 1e-3
        assert (diff > 1e-3).float().mean().item() < 2e-3


@pytest.mark.parametrize("n", [4, 64])
def test_frozen_chained_scale_against_the_default_frozen_schedule(n):
    """FrozenHotPath(chain_scale=True) vs the default frozen schedule at the cfg3 shapes: the scale planes of stages
    1 and 2 come from exact integer sums left by the previous pointwise instead of a scale launch; every output code
    equals the default schedule's except around the (rare) pixels whose 8-bit scale code rounds the other way
    (fraction of differing outputs < 1e-3, none by more than a few LSB), no overflow, and a replayed HIP graph of the
    chained schedule reproduces its eager result bit for bit (the integer atomics are order-independent)."""
    from codenet_amd import pipeline
    planes, res = [1024, 256, 128, 64], 16
    net = pipeline.build_hot_path(quantized=True, planes=planes, seed=46).cuda()
    xs = _inputs(n, planes[0], res, 2, 146)
    _warm_and_freeze(net, xs)
    base = pipeline.FrozenHotPath(net.deconv_layers)
    chained = pipeline.FrozenHotPath(net.deconv_layers, chain_scale=True)
    x = xs[0].cuda()
    a = base.forward_codes(x)[0].clone()
    b = chained.forward_codes(x)[0].clone()
    assert not base.overflowed() and not chained.overflowed()
    assert chained._bufs["stages"][0]["sums"] is not None and chained._bufs["stages"][1]["sums"] is not None
    d = (a.int() - b.int()).abs()
    frac = (d > 0).float().mean().item()
    print("   chained scale: %.2e of the output codes differ, max %d LSB" % (frac, d.max().item()))
    assert frac < 1e-3 and d.max().item() <= 8
    replay = chained.capture(x)
    for _ in range(3):
        assert torch.equal(replay(), b)


@pytest.mark.parametrize("M,C,Co", [(300, 64, 20), (4096, 1024, 256), (1000, 128, 64), (513, 256, 128), (77, 36, 5)])
def test_pointwise_q8_exact_integer_sums(M, C, Co):
    """cdn_codenet_pointwise_q8_forward against float64 arithmetic on the integer codes (the sum is exact in
    both; the epilogue is one fmaf + the code expression): bytes equal except where the fp32 epilogue lands
    within rounding distance of a code boundary (none expected: compared through the same fp32 expressions)."""
    from codenet_amd import _native as N_
    g = torch.Generator().manual_seed(M + C)
    a = torch.randint(-128, 128, (M, C), generator=g, dtype=torch.int32).to(torch.int8).cuda()
    qw = torch.randint(-8, 8, (Co, C), generator=g, dtype=torch.int32)
    cpad = (C + 63) // 64 * 64
    codes = torch.zeros(Co, cpad, dtype=torch.int8)
    codes[:, :C] = qw.to(torch.int8)
    codes = codes.cuda()
    wscale = (torch.rand(Co, generator=g) * 20 + 5).cuda()
    colsum = qw.sum(1).to(torch.int32).cuda()
    bias = (torch.randn(Co, generator=g) * 0.1).cuda()
    a_state = torch.zeros(8, dtype=torch.float32).cuda()
    r_state = torch.zeros(8, dtype=torch.float32).cuda()
    a_state[2], a_state[3] = 37.5, 11.0
    # output range chosen from the data so that nothing saturates
    isum = (a.cpu().double() + 11.0) @ qw.double().t()          # levels = codes + zero-point of the A quantiser
    v64 = torch.relu(isum / (37.5 * wscale.cpu().double()) + bias.cpu().double())
    r_scale = 255.0 / max(v64.max().item(), 1e-6) * 0.98
    r_state[2], r_state[3] = r_scale, 128.0
    out8 = torch.empty(M, Co, dtype=torch.int8).cuda()
    outf = torch.empty(M, Co).cuda()
    flag = torch.zeros(1, dtype=torch.int32).cuda()
    st = torch.cuda.current_stream().cuda_stream
    lib = N_.lib()
    N_.check(lib.cdn_codenet_pointwise_q8_forward(a.data_ptr(), a_state.data_ptr(), M, C, Co, codes.data_ptr(),
                                                  wscale.data_ptr(), colsum.data_ptr(), bias.data_ptr(), 1,
                                                  r_state.data_ptr(), None, outf.data_ptr(), flag.data_ptr(), st), "q8 f")
    N_.check(lib.cdn_codenet_pointwise_q8_forward(a.data_ptr(), a_state.data_ptr(), M, C, Co, codes.data_ptr(),
                                                  wscale.data_ptr(), colsum.data_ptr(), bias.data_ptr(), 1,
                                                  r_state.data_ptr(), out8.data_ptr(), None, flag.data_ptr(), st), "q8 b")
    torch.cuda.synchronize()
    assert flag.item() == 0
    # fp32 epilogue: fmaf(float(isum), 1 / (qs * sw), bias) -- reproduce it in fp32 on the exact integer sum
    rinv = 1.0 / (torch.tensor(37.5) * wscale.cpu())
    want = torch.relu(torch.addcmul(bias.cpu(), isum.float(), rinv))
    assert (outf.cpu() - want).abs().max().item() <= 2e-6 * max(1.0, want.abs().max().item())
    assert torch.equal(out8.cpu().float(), torch.round(r_state[2].cpu() * outf.cpu() - r_state[3].cpu()))
    # saturation is reported
    r_state[2] = r_scale * 4
    N_.check(lib.cdn_codenet_pointwise_q8_forward(a.data_ptr(), a_state.data_ptr(), M, C, Co, codes.data_ptr(),
                                                  wscale.data_ptr(), colsum.data_ptr(), bias.data_ptr(), 1,
                                                  r_state.data_ptr(), out8.data_ptr(), None, flag.data_ptr(), st), "q8 s")
    torch.cuda.synchronize()
    assert flag.item() == 1 and out8.max().item() == 127


@pytest.mark.parametrize("M,C,Co,lda,ldo", [(300, 58, 58, 64, 128), (1000, 24, 58, 32, 128), (513, 232, 232, 240, 464),
                                             (77, 116, 116, 128, 240), (4096, 464, 1024, 464, 1024)])
def test_pointwise_q8_strided_rows_and_output_map(M, C, Co, lda, ldo):
    """cdn_codenet_pointwise_q8_strided_forward: rows of lda bytes holding C codes (the bytes behind them are junk
    paired with zero weight codes), outputs scattered through out_map into rows of ldo bytes whose other slots stay
    untouched -- the same bytes as the dense entry point on the compacted operands."""
    from codenet_amd import _native as N_
    g = torch.Generator().manual_seed(M + C + ldo)
    a = torch.randint(-128, 128, (M, lda), generator=g, dtype=torch.int32).to(torch.int8).cuda()
    qw = torch.randint(-8, 8, (Co, C), generator=g, dtype=torch.int32)
    cpad = (C + 63) // 64 * 64
    codes = torch.zeros(Co, cpad, dtype=torch.int8)
    codes[:, :C] = qw.to(torch.int8)
    codes = codes.cuda()
    wscale = (torch.rand(Co, generator=g) * 20 + 5).cuda()
    colsum = qw.sum(1).to(torch.int32).cuda()
    bias = (torch.randn(Co, generator=g) * 0.1).cuda()
    a_state = torch.zeros(8, dtype=torch.float32).cuda()
    r_state = torch.zeros(8, dtype=torch.float32).cuda()
    a_state[2], a_state[3] = 37.5, -23.0
    r_state[2], r_state[3] = 255.0 / (C * 0.35), 128.0
    omap = torch.randperm(ldo, generator=g)[:Co].to(torch.int32).cuda()
    lib, st = N_.lib(), torch.cuda.current_stream().cuda_stream
    flag = torch.zeros(1, dtype=torch.int32).cuda()
    dense_a = a[:, :C].contiguous() if C % 4 == 0 else None
    out = torch.full((M, ldo), 77, dtype=torch.int8).cuda()
    N_.check(lib.cdn_codenet_pointwise_q8_strided_forward(
        a.data_ptr(), a_state.data_ptr(), M, C, Co, lda, ldo, codes.data_ptr(), wscale.data_ptr(), colsum.data_ptr(),
        bias.data_ptr(), 1, omap.data_ptr(), r_state.data_ptr(), out.data_ptr(), None, flag.data_ptr(), st), "strided")
    # expected: exact integer sum -> the fp32 epilogue expressions -> code
    isum = (a[:, :C].cpu().double() - 23.0) @ qw.double().t()
    rinv = 1.0 / (torch.tensor(37.5) * wscale.cpu())
    v = torch.relu(torch.addcmul(bias.cpu(), isum.float(), rinv))
    want = torch.round(r_state[2].cpu() * v - r_state[3].cpu()).clamp(-128, 127)
    got = out.cpu()
    touched = torch.zeros(ldo, dtype=torch.bool)
    touched[omap.cpu().long()] = True
    assert torch.equal(got[:, omap.cpu().long()].float(), want)
    assert bool((got[:, ~touched] == 77).all())
    if dense_a is not None:
        dense = torch.empty(M, Co, dtype=torch.int8).cuda()
        N_.check(lib.cdn_codenet_pointwise_q8_forward(dense_a.data_ptr(), a_state.data_ptr(), M, C, Co, codes.data_ptr(),
                                                      wscale.data_ptr(), colsum.data_ptr(), bias.data_ptr(), 1,
                                                      r_state.data_ptr(), dense.data_ptr(), None, flag.data_ptr(), st), "dense")
        assert torch.equal(dense.cpu(), got[:, omap.cpu().long()])


@pytest.mark.parametrize("N,R,stride", [(2, 64, 2), (3, 96, 4), (1, 250, 2)])
def test_stem_q8_is_the_fp32_stem_then_the_code(N, R, stride):
    """cdn_codenet_stem_q8_forward against cdn_codenet_stem_forward (frozen state, running = 0) followed by the code
    expression: identical bytes, padding bytes of the 32-byte rows untouched."""
    from codenet_amd import _native as N_, ops
    lib, dev = N_.lib(), torch.device("cuda", 0)
    g = torch.Generator().manual_seed(R)
    img = torch.randn(N, 3, R, R, generator=g).to(dev)
    w = (torch.randn(24, 27, generator=g) * 0.3).to(dev)
    b = (torch.randn(24, generator=g) * 0.1).to(dev)
    Ho = (R + 2 - 3) // stride + 1
    aux = lib.cdn_codenet_aux_workspace_bytes()
    ws = torch.zeros(aux // 4 + 64, device=dev)
    wp = (ws.data_ptr() + 255) // 256 * 256
    sr = ops.quantact_state(dev)
    sr.view(torch.float32)[2], sr.view(torch.float32)[3] = 18.0, 128.0
    st = torch.cuda.current_stream().cuda_stream
    out = torch.empty(N, Ho * Ho, 24, device=dev)
    N_.check(lib.cdn_codenet_stem_forward(img.data_ptr(), N, R, R, 24, stride, w.data_ptr(), b.data_ptr(), 1, None, None,
                                          None, 8, 0.99, 0, wp, (ws.numel() * 4 - 256) // 256 * 256, out.data_ptr(), st),
             "stem fp32")
    out8 = torch.full((N, Ho * Ho, 32), 55, dtype=torch.int8, device=dev)
    of = torch.zeros(1, dtype=torch.int32, device=dev)
    N_.check(lib.cdn_codenet_stem_q8_forward(img.data_ptr(), N, R, R, 24, stride, w.data_ptr(), b.data_ptr(), 1,
                                             sr.data_ptr(), out8.data_ptr(), 32, of.data_ptr(), st), "stem q8")
    want = torch.round(18.0 * out - 128.0)
    inside = (want >= -128) & (want <= 127)
    assert torch.equal(out8[:, :, :24].float()[inside], want[inside])
    assert bool((out8[:, :, 24:] == 55).all())
    assert bool(of.item()) == bool((~inside).any().item())


@pytest.mark.parametrize("N,Hs,Ws,classes", [(2, 16, 16, 2), (3, 8, 32, 20), (1, 12, 20, 2), (2, 16, 48, 4)])
def test_head_tail_on_byte_codes_equals_the_fp32_form(N, Hs, Ws, classes):
    """cdn_codenet_head_tail_small_q8_forward (y1 as the byte codes of its QuantAct) against
    cdn_codenet_head_tail_small_forward on the pre-quantisation fp32 tensor that produced those codes: identical
    outputs (a code decodes to the value the fp32 form's fake-quantisation gives)."""
    from codenet_amd import _native as N_, ops
    lib, dev = N_.lib(), torch.device("cuda", 0)
    g = torch.Generator().manual_seed(Hs * Ws + classes)
    y1 = (torch.rand(N, Hs * Ws, 64, generator=g) * 4.0).to(dev)
    q1 = ops.quantact_state(dev); q2 = ops.quantact_state(dev)
    q1.view(torch.float32)[2], q1.view(torch.float32)[3] = 255.0 / 4.2, 128.0 - 3.0
    q2.view(torch.float32)[2], q2.view(torch.float32)[3] = 11.0, 128.0
    s1, z1 = q1.view(torch.float32)[2], q1.view(torch.float32)[3]
    codes = torch.round(s1 * y1 - z1)
    assert codes.min().item() >= -128 and codes.max().item() <= 127
    y8 = codes.to(torch.int8).contiguous()
    w_dw = (torch.randn(64, 9, generator=g) * 0.3).to(dev)
    b_dw = (torch.randn(64, generator=g) * 0.1).to(dev)
    wq = torch.randint(-8, 8, (classes, 64), generator=g, dtype=torch.int32)
    w_codes = wq.to(torch.int8).to(dev)
    w_scale = (torch.rand(classes, generator=g) * 10 + 3).to(dev)
    w_colsum = wq.sum(1).to(torch.int32).to(dev)
    bias = (torch.randn(classes, generator=g) * 0.1).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    a = torch.zeros(N, classes, 2 * Hs, 2 * Ws, device=dev)
    b = torch.zeros_like(a)
    of = torch.zeros(1, dtype=torch.int32, device=dev)
    N_.check(lib.cdn_codenet_head_tail_small_forward(y1.data_ptr(), q1.data_ptr(), N, 64, Hs, Ws, w_dw.data_ptr(),
                                                     b_dw.data_ptr(), q2.data_ptr(), w_codes.data_ptr(),
                                                     w_scale.data_ptr(), w_colsum.data_ptr(), bias.data_ptr(), classes,
                                                     a.data_ptr(), st), "fp32 tail")
    N_.check(lib.cdn_codenet_head_tail_small_q8_forward(y8.data_ptr(), q1.data_ptr(), N, 64, Hs, Ws, w_dw.data_ptr(),
                                                        b_dw.data_ptr(), q2.data_ptr(), w_codes.data_ptr(),
                                                        w_scale.data_ptr(), w_colsum.data_ptr(), bias.data_ptr(),
                                                        classes, b.data_ptr(), of.data_ptr(), st), "byte tail")
    assert a.abs().max().item() > 0 and torch.equal(a, b) and of.item() == 0
    if classes > 4 or Ws % 16 == 0:            # the matrix-core form: a y2 level outside its nibble split is reported
        q2.view(torch.float32)[2] = 4000.0
        N_.check(lib.cdn_codenet_head_tail_small_q8_forward(y8.data_ptr(), q1.data_ptr(), N, 64, Hs, Ws, w_dw.data_ptr(),
                                                            b_dw.data_ptr(), q2.data_ptr(), w_codes.data_ptr(),
                                                            w_scale.data_ptr(), w_colsum.data_ptr(), bias.data_ptr(),
                                                            classes, b.data_ptr(), of.data_ptr(), st), "byte tail")
        assert of.item() == 1


@pytest.mark.parametrize("N,C,H,W,stride,Co,ldo", [
    (2, 58, 64, 64, 1, 58, 128), (2, 116, 32, 32, 1, 116, 240), (3, 232, 16, 16, 1, 232, 464), (2, 24, 128, 128, 2, 58, 128),
    (2, 58, 128, 128, 2, 58, 128), (1, 116, 64, 64, 2, 116, 240), (2, 232, 32, 32, 2, 232, 464), (2, 36, 16, 16, 1, 20, 0),
    (1, 58, 64, 128, 1, 58, 64)])
def test_dwpw_q8_equals_depthwise_then_pointwise(N, C, H, W, stride, Co, ldo):
    """cdn_codenet_dwpw_q8_forward (the depthwise computed into the LDS operand tile of the 1x1 conv) against
    cdn_codenet_dw3x3_q8_forward + cdn_codenet_pointwise_q8_strided_forward: identical bytes, slots outside the output
    map untouched, the overflow flag of either stage reported."""
    from codenet_amd import _native as N_, ops
    lib, dev = N_.lib(), torch.device("cuda", 0)
    assert lib.cdn_codenet_dwpw_q8_supported(C, H, W, stride, Co) == 1
    g = torch.Generator().manual_seed(C + H + Co)
    ld = (C + 15) // 16 * 16
    a8 = torch.randint(-128, 128, (N, H * W, ld), generator=g, dtype=torch.int32).to(torch.int8).to(dev)
    sa, sd, sr = ops.quantact_state(dev), ops.quantact_state(dev), ops.quantact_state(dev)
    sa.view(torch.float32)[2], sa.view(torch.float32)[3] = 31.7, -17.0
    sd.view(torch.float32)[2], sd.view(torch.float32)[3] = 6.0, 21.0
    sr.view(torch.float32)[2], sr.view(torch.float32)[3] = 255.0 / (C * 0.9), 128.0
    w_dw = (torch.randn(C, 9, generator=g) * 0.4).to(dev)
    b_dw = (torch.randn(C, generator=g) * 0.2).to(dev)
    qw = torch.randint(-8, 8, (Co, C), generator=g, dtype=torch.int32)
    cpad = (C + 63) // 64 * 64
    codes = torch.zeros(Co, cpad, dtype=torch.int8)
    codes[:, :C] = qw.to(torch.int8)
    codes = codes.to(dev)
    wscale = (torch.rand(Co, generator=g) * 20 + 5).to(dev)
    colsum = qw.sum(1).to(torch.int32).to(dev)
    bias = (torch.randn(Co, generator=g) * 0.1).to(dev)
    ldo_ = ldo if ldo else Co
    omap = torch.randperm(ldo_, generator=g)[:Co].to(torch.int32).to(dev) if ldo else None
    Ho, Wo = ((H - 1) // 2 + 1, (W - 1) // 2 + 1) if stride == 2 else (H, W)
    M = N * Ho * Wo
    st = torch.cuda.current_stream().cuda_stream
    of1 = torch.zeros(1, dtype=torch.int32, device=dev)
    of2 = torch.zeros(1, dtype=torch.int32, device=dev)
    t2 = torch.zeros(M, ld, dtype=torch.int8, device=dev)
    want = torch.full((M, ldo_), 77, dtype=torch.int8, device=dev)
    got = torch.full((M, ldo_), 77, dtype=torch.int8, device=dev)
    N_.check(lib.cdn_codenet_dw3x3_q8_forward(a8.data_ptr(), sa.data_ptr(), N, C, H, W, stride, ld, ld, w_dw.data_ptr(),
                                              b_dw.data_ptr(), 0, sd.data_ptr(), t2.data_ptr(), of1.data_ptr(), st), "dw")
    N_.check(lib.cdn_codenet_pointwise_q8_strided_forward(
        t2.data_ptr(), sd.data_ptr(), M, C, Co, ld, ldo_, codes.data_ptr(), wscale.data_ptr(), colsum.data_ptr(),
        bias.data_ptr(), 1, omap.data_ptr() if omap is not None else None, sr.data_ptr(), want.data_ptr(), None,
        of1.data_ptr(), st), "pw")
    N_.check(lib.cdn_codenet_dwpw_q8_forward(
        a8.data_ptr(), sa.data_ptr(), N, C, H, W, stride, ld, w_dw.data_ptr(), b_dw.data_ptr(), 0, sd.data_ptr(), Co,
        codes.data_ptr(), wscale.data_ptr(), colsum.data_ptr(), bias.data_ptr(), 1, ldo, omap.data_ptr() if omap is not None
        else None, sr.data_ptr(), got.data_ptr(), of2.data_ptr(), st), "dwpw")
    assert torch.equal(got, want)
    assert of1.item() == of2.item()
    assert want.float().std().item() > 1.0          # (a non-degenerate case)


@pytest.mark.parametrize("N,C,H,W,ld", [(2, 24, 9, 11, 32), (1, 24, 64, 64, 32), (3, 58, 7, 8, 64)])
def test_maxpool_q8_is_the_code_of_the_pooled_values(N, C, H, W, ld):
    """cdn_codenet_maxpool3x3s2_q8_forward against F.max_pool2d(3, 2, 1) of the code tensor (the pool follows the
    QuantAct and fake-quantisation is monotone: the maximum of the codes is the code of the maximum)."""
    import torch.nn.functional as F
    from codenet_amd import _native as N_
    lib, dev = N_.lib(), torch.device("cuda", 0)
    g = torch.Generator().manual_seed(H * W + C)
    a8 = torch.randint(-128, 128, (N, H * W, ld), generator=g, dtype=torch.int32).to(torch.int8).to(dev)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.full((N, Ho * Wo, ld), 55, dtype=torch.int8, device=dev)
    N_.check(lib.cdn_codenet_maxpool3x3s2_q8_forward(a8.data_ptr(), N, C, H, W, ld, ld, out.data_ptr(),
                                                     torch.cuda.current_stream().cuda_stream), "maxpool q8")
    x = a8[:, :, :C].float().reshape(N, H, W, C).permute(0, 3, 1, 2)
    want = F.max_pool2d(x, 3, 2, 1).permute(0, 2, 3, 1).reshape(N, Ho * Wo, C)
    cq4 = (C + 3) // 4 * 4
    assert torch.equal(out[:, :, :C].float(), want)
    assert bool((out[:, :, cq4:] == 55).all())
