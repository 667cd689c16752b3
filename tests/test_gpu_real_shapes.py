"""Fused hot path (codenet_fused.hip: scale -> dw2 / dw2u -> pwi8 / pw3 -> unpack) against the CPU oracle at the
BASELINE configurations' REAL stage shapes and batches (VERDICT r1 "next" #1):

    cfg3  CoDeNet1x 512x512 W4A8   planes [1024,256,128,64], 16^2 / 32^2 / 64^2, N = 4 and N = 64
          -> dw2_kernel<64,NCHW>, dw2u_kernel<64> (stage 1), dw2u_kernel<32> (stage 2: the stored 32x32 plane
             forces 32-channel chunks), all three pwi8 tilings; eager launches and HIP-graph replay
    cfg4  CoDeNet2x per-rank shard   planes [2153,256,128,64], N = 32
    cfg2  CoDeNet1x 256x256 fp32    planes [1024,256,128,64], 8^2 / 16^2 / 32^2, N = 32

Stage shapes: lib/models/networks/shufflenetv2_dcn.py:199-202,293-301.  Oracle: oracle/quant.py::stage_w4a8 /
stage_fp32 over oracle/dcn_oracle.c (im2col + per-group contraction), three consecutive forwards on different
inputs so that the running QuantAct ranges ("+=" initialisation, then two EMA steps) are part of the check.

Acceptance (same as tests/test_gpu_parity.py::test_fused_hot_path_matches_modules_and_oracle): fp32 within
1e-3; W4A8 outputs are 8-bit fake-quantised values -- fp32 re-association between CPU and GPU (~1e-6) can move a
pre-quantisation value across a rounding boundary, i.e. flip a code by ONE LSB: <= 1 LSB, on < 0.02 % of the
elements (measured 0.001-0.004 %, tools/experiments/code_mismatch_rate.py); every tracked range within 3e-6 of its
magnitude (+1e-6; measured <= 6.3e-7).

That the summation ORDER is the only cause of those flips -- not a tie rule, an edge weight or a quantiser expression --
is demonstrated in tests/test_gpu_exact_codes.py (round 5): on inputs where every order gives the same fp32 value the
same kernels match the oracle with torch.equal (s, d, relu(y), all ranges, output codes; running and frozen ranges,
graph replay, byte-code schedule), and on THESE random inputs the oracle's QuantAct applied to the GPU's own
pre-quantisation tensors reproduces every device range bit for bit and every consumer's output to a tenth of one
flipped code's effect.
"""
import copy

import pytest
import torch
import torch.nn.functional as F

from oracle import quant as Q

pytestmark = pytest.mark.gpu


def _inputs(n, c, res, forwards, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(n, c, res, res, generator=g).abs_() * (1.66 * (1.0 + 0.15 * i)) for i in range(forwards)]


def _oracle(net_cpu, xs, quantized):
    """-> (outputs per forward, per-forward list of [(x_min, x_max)] * 9 for the QuantActs in module order)."""
    mods = list(net_cpu.deconv_layers)
    outs, ranges = [], []
    acts = [[Q.QuantActState(), Q.QuantActState(), Q.QuantActState()] for _ in range(len(mods) // 3)]
    with torch.no_grad():
        for x in xs:
            cur = x
            if not quantized:
                for i in range(0, len(mods), 4):
                    op, bn = mods[i], mods[i + 1]
                    r = Q.stage_fp32(cur, op.conv_scale.weight, op.conv_scale.bias, op.conv.weight,
                                     op.conv_channel.weight)
                    cur = F.interpolate(torch.relu(bn(r["y"])), scale_factor=2, mode="nearest")
            else:
                for k, i in enumerate(range(0, len(mods), 3)):
                    q = mods[i]
                    bnm = q.quant_conv_channel_bn.bn
                    bn = (bnm.weight, bnm.bias, bnm.running_mean, bnm.running_var, bnm.eps)
                    r = Q.stage_w4a8(cur, q.quant_conv_scale.weight, q.quant_conv_scale.bias,
                                     q.quant_deform_conv.weight, q.quant_conv_channel_bn.conv.weight, bn,
                                     acts[k][0], acts[k][1])
                    cur = F.interpolate(acts[k][2](torch.relu(r["y"])), scale_factor=2, mode="nearest")
                ranges.append([(a.x_min.item(), a.x_max.item()) for st in acts for a in st])
            outs.append(cur)
    return outs, ranges


def _gpu_ranges(net):
    """(x_min, x_max) of the nine QuantActs in the oracle's order: per stage s, d, r."""
    out = []
    mods = list(net.deconv_layers)
    for i in range(0, len(mods), 3):
        q, post = mods[i], mods[i + 1]
        for a in (q.quant_act[1], q.quant_identity_deform, post[1]):
            out.append((a.x_min.item(), a.x_max.item()))
    return out


def _check(y, ref, quantized, lsb, what):
    assert y.shape == ref.shape, what
    diff = (y.cpu() - ref).abs()
    if not quantized:
        assert diff.max().item() < 1e-3, "%s: max |diff| %g" % (what, diff.max().item())
        return
    assert diff.max().item() <= 1.05 * lsb + 1e-3, "%s: max |diff| %g vs LSB %g" % (what, diff.max().item(), lsb)
    frac = (diff > 1e-3).float().mean().item()
    assert frac < 2e-4, "%s: %.3g of the outputs differ" % (what, frac)     # measured: 1e-5 .. 4e-5


def _check_ranges(got, want, what):
    for k, ((a0, a1), (b0, b1)) in enumerate(zip(got, want)):
        tol = 3e-6 * max(abs(b0), abs(b1), 1.0) + 1e-6      # measured: <= 6.3e-7 relative
        assert abs(a0 - b0) <= tol and abs(a1 - b1) <= tol, \
            "%s: QuantAct %d range (%g, %g) vs oracle (%g, %g)" % (what, k, a0, a1, b0, b1)


def _run_case(planes, res, n, quantized, forwards, seed, graph=False):
    from codenet_amd import pipeline
    net = pipeline.build_hot_path(quantized=quantized, planes=planes, seed=seed)
    net_cpu = copy.deepcopy(net)
    xs = _inputs(n, planes[0], res, forwards, seed + 100)
    ref, ref_ranges = _oracle(net_cpu, xs, quantized)
    net = net.cuda()
    if quantized:
        pipeline.set_running_stat(net, True)          # reference-faithful: ranges keep moving in eval()
    fused = pipeline.FusedHotPath(net.deconv_layers)
    what = "planes %s res %d N %d %s%s" % (planes, res, n, "W4A8" if quantized else "fp32", " graph" if graph else "")
    if graph:
        # one eager pass (forward 0) allocates and warms; forwards 1.. replay ONE captured graph over a static
        # input buffer, exactly what bench.py times
        xbuf = xs[0].cuda()
        y = fused(xbuf).clone()
        last = list(net.deconv_layers)[-2][1] if quantized else None
        _check(y, ref[0], quantized, (last.x_max - last.x_min).item() / 255.0 if quantized else 0.0, what + " fwd 0")
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):                      # capturing records the launches, it does not run them
            out = fused(xbuf)
        for it in range(1, forwards):
            xbuf.copy_(xs[it].cuda())
            g.replay()
            torch.cuda.synchronize()
            lsb = (last.x_max - last.x_min).item() / 255.0 if quantized else 0.0
            _check(out, ref[it], quantized, lsb, what + " fwd %d" % it)
            if quantized:
                _check_ranges(_gpu_ranges(net), ref_ranges[it], what + " fwd %d" % it)
        return
    for it in range(forwards):
        y = fused(xs[it].cuda())
        lsb = 0.0
        if quantized:
            last = list(net.deconv_layers)[-2][1]
            lsb = (last.x_max - last.x_min).item() / 255.0
            _check_ranges(_gpu_ranges(net), ref_ranges[it], what + " fwd %d" % it)
        _check(y, ref[it], quantized, lsb, what + " fwd %d" % it)


CFG3 = [1024, 256, 128, 64]
CFG4 = [2153, 256, 128, 64]


@pytest.mark.parametrize("n", [4, 64])
def test_cfg3_fused_w4a8_real_stage_shapes(n):
    """BASELINE configs[2]: CoDeNet1x config-c 512x512 W4A8, batch 64 (and 4): 16^2 -> 32^2 -> 64^2."""
    _run_case(CFG3, 16, n, True, 3, seed=31)


def test_cfg3_fused_w4a8_real_stage_shapes_graph_replay():
    """The same schedule as ONE captured HIP graph replayed over a static input buffer (what bench.py times)."""
    _run_case(CFG3, 16, 4, True, 3, seed=32, graph=True)


def test_cfg3_fused_fp32_real_stage_shapes():
    _run_case(CFG3, 16, 4, False, 1, seed=33)


def test_cfg4_fused_w4a8_per_rank_shard():
    """BASELINE configs[3]: CoDeNet2x 512x512 W4A8, 32 images per rank (stage 0: C = 2153, odd)."""
    _run_case(CFG4, 16, 32, True, 2, seed=34)


def test_cfg2_fused_fp32_batch32():
    """BASELINE configs[1]: CoDeNet1x config-a 256x256 fp32 batch 32: 8^2 -> 16^2 -> 32^2."""
    _run_case(CFG3, 8, 32, False, 1, seed=35)


@pytest.mark.parametrize("planes,H,N", [(None, 8, 32), (None, 16, 8), ([64, 48, 40, 24], 12, 3)])
def test_fp32_chained_scale_sums_equal_the_scale_kernel_to_rounding(planes, H, N):
    """Round 6: in the fp32 model nothing sits between a stage's BN + ReLU and the next stage's
    scale prediction, so the streaming f32 pointwise kernel leaves that prediction's partial sums per column tile and the
    next stage runs without its scale launch (cdn_codenet_stage_fused_forward_chain: 10 -> 8 launches at cfg2).  The
    chained schedule against the unchained one on the same weights: the scale planes differ by fp32 re-association only
    (<= 1e-5 of |s|'s range), the final outputs by what that moves the samples (<= 2e-4 relative) -- both sit inside the
    1e-3 bound against the oracle that test_cfg2_fused_fp32_batch32 checks with the chain ON (the default).  The third
    shape has no streaming-kernel form at every stage (K % 32 != 0): the chain is simply not taken there."""
    from codenet_amd import _native as N_, pipeline
    planes = planes or CFG3
    net = pipeline.build_hot_path(quantized=False, planes=planes, seed=41).cuda()
    x = _inputs(N, planes[0], H, 1, 141)[0].cuda()
    a = pipeline.FusedHotPath(net.deconv_layers, chain_scale=True)
    b = pipeline.FusedHotPath(net.deconv_layers, chain_scale=False)
    ya, yb = a(x).clone(), b(x).clone()
    parts = [sb["parts"] for sb in a._bufs["stages"]]
    lib = N_.lib()
    assert parts[:-1] == [int(lib.cdn_codenet_stage_chain_parts(N, sb["C"], sb["Co"], sb["H"], sb["W"]))
                          for sb in a._bufs["stages"][:-1]] and parts[-1] == 0
    if planes is CFG3:
        assert all(p > 0 for p in parts[:-1])                  # cfg2 / cfg3-fp32: both boundaries chained
    scale = yb.abs().max().item()
    assert (ya - yb).abs().max().item() <= 2e-4 * scale, (ya - yb).abs().max().item() / scale
    assert torch.equal(a(x), ya)                                # fixed summation order: repeatable bit for bit
    g = a.capture(x)                                            # and capturable
    assert torch.equal(g(), ya)


def test_cfg2_modules_fp32_batch32():
    """configs[1] as BASELINE words it -- 'deform-conv HIP kernel only, PyTorch-ROCm for the rest': the
    nn.Module chain (DeformConvWithOffsetScaleBoundPositive -> BatchNorm2d -> ReLU -> Upsample) at batch 32."""
    from codenet_amd import pipeline
    net = pipeline.build_hot_path(quantized=False, planes=CFG3, seed=36)
    xs = _inputs(32, 1024, 8, 1, 136)
    ref, _ = _oracle(copy.deepcopy(net), xs, False)
    with torch.no_grad():
        y = net.cuda()(xs[0].cuda())
    _check(y, ref[0], False, 0.0, "cfg2 module path")


def test_cfg3_modules_w4a8_batch64_matches_fused():
    """Module-by-module W4A8 chain (the drop-in path) vs the fused schedule at batch 64: ranges to 1e-5,
    outputs within one LSB on < 0.2 %."""
    from codenet_amd import pipeline
    net = pipeline.build_hot_path(quantized=True, planes=CFG3, seed=37)
    xs = [x.cuda() for x in _inputs(64, 1024, 16, 2, 137)]
    a, b = copy.deepcopy(net).cuda(), copy.deepcopy(net).cuda()
    fused = pipeline.FusedHotPath(b.deconv_layers)
    for it, x in enumerate(xs):
        with torch.no_grad():
            ya = a(x)
        yb = fused(x)
        last = list(b.deconv_layers)[-2][1]
        lsb = (last.x_max - last.x_min).item() / 255.0
        diff = (ya - yb).abs()
        assert diff.max().item() <= 1.05 * lsb + 1e-3
        assert (diff > 1e-3).float().mean().item() < 2e-3
        _check_ranges(_gpu_ranges(b), _gpu_ranges(a), "modules vs fused fwd %d" % it)


@pytest.mark.parametrize("res,n,quantized", [(16, 64, True), (16, 33, True), (8, 32, False), (8, 40, True)])
def test_persistent_dma_gather_is_bit_identical_to_per_item_gather(res, n, quantized):
    """Stage 0 (NCHW input): dw0p_kernel -- one persistent workgroup per CU, the next item's planes in flight by LDS-DMA
    (global_load_lds) during the gather, XOR-swizzled raw buffer -> [cell][channel] image -- against dw2_kernel (one
    workgroup per item, register staging).  Same staged values, same gather routine: outputs and all nine QuantAct
    ranges must be bit-identical over three forwards, for even (4 / 2 items per workgroup) and ragged (33 x 16 = 528,
    40 x 16 = 640 items over 256 workgroups) item counts.  Both also run against the oracle in the tests above (the
    automatic mode picks the persistent kernel at N = 64 / 32)."""
    from codenet_amd import _native as N_, pipeline
    lib = N_.lib()
    net = pipeline.build_hot_path(quantized=quantized, planes=CFG3, seed=41)
    xs = [x.cuda() for x in _inputs(n, 1024, res, 3, 141)]
    a, b = copy.deepcopy(net).cuda(), copy.deepcopy(net).cuda()
    if quantized:
        pipeline.set_running_stat(a, True)
        pipeline.set_running_stat(b, True)
    fa, fb = pipeline.FusedHotPath(a.deconv_layers), pipeline.FusedHotPath(b.deconv_layers)
    fa.gather_flag, fb.gather_flag = pipeline.GATHER_PER_ITEM, pipeline.GATHER_PERSISTENT     # a per-call argument
    for x in xs:
        ya = fa(x).clone()
        yb = fb(x).clone()
        assert torch.equal(ya, yb)
        if quantized:
            assert _gpu_ranges(a) == _gpu_ranges(b)


@pytest.mark.parametrize("n", [32, 5])
def test_ragged_channel_count_on_the_persistent_gather_is_bit_identical(n):
    """CoDeNet2x stage 0 (C = 2153 = 33 * 64 + 41; VERDICT r3 "next" #4b): round 3 kept it on dw2_kernel with scalar
    stores and the int8 pointwise on guarded scalar loads (rows of 2153 floats are not 16-byte aligned).  Now the rows of
    d are padded to 2176, the persistent LDS-DMA gather takes the ragged last chunk (missing planes / weights = the last
    real channel again) and the int8 pointwise runs its whole-tile form against zero-padded weight codes.  Same values:
    outputs and all nine QuantAct ranges bit-identical to the per-item schedule (the unpadded round-3 path) over three
    forwards; both are checked against the oracle by test_cfg4_* above."""
    from codenet_amd import pipeline
    net = pipeline.build_hot_path(quantized=True, planes=CFG4, seed=43)
    xs = [x.cuda() for x in _inputs(n, 2153, 16, 3, 143)]
    a, b = copy.deepcopy(net).cuda(), copy.deepcopy(net).cuda()
    pipeline.set_running_stat(a, True)
    pipeline.set_running_stat(b, True)
    fa, fb = pipeline.FusedHotPath(a.deconv_layers), pipeline.FusedHotPath(b.deconv_layers)
    fa.gather_flag = pipeline.GATHER_PER_ITEM
    for x in xs:
        ya = fa(x).clone()
        yb = fb(x).clone()
        assert torch.equal(ya, yb)
        assert _gpu_ranges(a) == _gpu_ranges(b)


def test_persistent_dma_gather_frozen_codes_bit_identical():
    """The same for the frozen byte-code schedule (OUT8 instantiation of dw0p_kernel): codes and overflow flag."""
    from codenet_amd import _native as N_, pipeline
    lib = N_.lib()
    net = pipeline.build_hot_path(quantized=True, planes=CFG3, seed=42).cuda()
    xs = [x.cuda() for x in _inputs(64, 1024, 16, 2, 142)]
    pipeline.set_running_stat(net, True)
    fused = pipeline.FusedHotPath(net.deconv_layers)
    for x in xs:
        fused(x)
    pipeline.set_running_stat(net, False)
    frz = pipeline.FrozenHotPath(net.deconv_layers)
    frz.gather_flag = pipeline.GATHER_PER_ITEM
    ca = frz.forward_codes(xs[0])[0].clone()
    frz.gather_flag = pipeline.GATHER_PERSISTENT
    cb = frz.forward_codes(xs[0])[0].clone()
    assert ca.dtype == torch.int8 and torch.equal(ca, cb) and not frz.overflowed()
