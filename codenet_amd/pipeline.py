"""The hot path as the reference composes it: ``PoseShuffleNetV2.deconv_layers``
(lib/models/networks/shufflenetv2_dcn.py:286-312) -- three
[DeformConvWithOffsetScaleBoundPositive, BatchNorm2d, ReLU, Upsample x2] groups -- in fp32, or after
``quantize_shufflenetv2_dcn`` (quantize_model.py:70-82) three
[QuantDeformConvWithOffsetScaleBoundPositive, Sequential(ReLU, QuantAct), Upsample x2] groups.

Used by bench.py, the smoke test and the multi-process tests.  Data-parallel inference shards
independent image batches over one process per GPU; the only collective is a start-up broadcast
of parameters and buffers (weights, BN statistics, QuantAct ranges) from rank 0 over RCCL
(SURVEY.md section 8e).
"""
import os

import torch
import torch.nn as nn

from .modules.dcn_deform_conv import DeformConvWithOffsetScaleBoundPositive
from .portable_quantizer.quantization_utils.quantize_model import quantize_deform_stages

BN_MOMENTUM = 0.1


def stage_shapes(input_res=512, w2=False):
    """[(C_in, C_out, H, W)] of the three deform stages."""
    c0 = 2153 if w2 else 1024          # shufflenetv2_dcn.py:199-202,293-296
    r = input_res // 32
    return [(c0, 256, r, r), (256, 128, 2 * r, 2 * r), (128, 64, 4 * r, 4 * r)]


class DeconvLayers(nn.Module):
    """Container with the reference's attribute name so the quantiser's surgery applies."""

    def __init__(self, w2=False, planes=None):
        super().__init__()
        planes = planes or [2153 if w2 else 1024, 256, 128, 64]   # C_in of stage 0, then every C_out
        layers = []
        for cin, cout in zip(planes[:-1], planes[1:]):
            layers += [
                DeformConvWithOffsetScaleBoundPositive(cin, cout, 3, 1, 1, groups=cout, bias=False,
                                                        hidden_state=128, BN_MOMENTUM=BN_MOMENTUM),
                nn.BatchNorm2d(cout, momentum=BN_MOMENTUM),
                nn.ReLU(inplace=True),
                nn.Upsample(scale_factor=2, mode="nearest"),
            ]
        self.deconv_layers = nn.Sequential(*layers)

    def forward(self, x):
        from .functions.codenet_stage import forward_stage_blocks
        return forward_stage_blocks(self.deconv_layers, x)      # (== self.deconv_layers(x); fused blocks in the QAT step)


def build_hot_path(w2=False, quantized=True, seed=317, scale_std=3.0, planes=None, wt_percentile=False):
    """Seeded synthetic weights: reference initialisers, except a non-degenerate
    conv_scale (weight ~ N(0, scale_std/sqrt(C)), bias 1 => s ~ N(1, scale_std) on unit-power inputs,
    clipped to [-7, 8] with ~1 % of pixels at each clamp) and non-trivial BN running statistics."""
    g = torch.Generator().manual_seed(seed)
    net = DeconvLayers(w2=w2, planes=planes)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, DeformConvWithOffsetScaleBoundPositive):
                C = m.in_channels
                m.conv_scale.weight.copy_(torch.randn(1, C, 1, 1, generator=g) * (scale_std / C ** 0.5))
                bound = 1.0 / (9 * C) ** 0.5
                m.conv.weight.copy_(torch.empty(C, 1, 3, 3).uniform_(-bound, bound, generator=g))
                m.conv_channel.weight.copy_(
                    torch.randn(m.out_channels, C, 1, 1, generator=g) * (2.0 / C) ** 0.5)
            elif isinstance(m, nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
                m.weight.copy_(torch.rand(m.num_features, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    if quantized:
        # (the README's QAT command passes --wt-percentile, its test commands do not: README.md:87-116)
        quantize_deform_stages(net, 4, 8, "symmetric", "asymmetric", True, bool(wt_percentile), False)
    return net.eval()


def make_input(batch, input_res=512, w2=False, seed=0, device="cpu"):
    """Stage-0 input: what layer4 (conv1x1 + BN + ReLU [+ QuantAct]) hands over -- non-negative,
    unit power."""
    C, _, H, W = stage_shapes(input_res, w2)[0]
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, C, H, W, generator=g).abs_() * 1.66   # E[x^2] ~ 1 after the fold
    return x.to(device)


def set_running_stat(net, flag):
    """running_stat=True is the reference's behaviour even in eval();
    False freezes the QuantAct ranges."""
    from .portable_quantizer.quant_modules import QuantAct
    for m in net.modules():
        if isinstance(m, QuantAct):
            m.running_stat = flag


def cover_frozen_ranges(net, batches, forward=None, margin=0.02, passes=2, spread=None):
    """Deployment step for the byte-code serving mode.  With ``running_stat = False`` the reference keeps the EMA
    ranges it trained with (quant_modules.py:203-219 skipped) and a value outside such a range simply becomes a code
    beyond the 8-bit grid (the fake-quantised float has no clamp, quant_utils.py:132-171).  A byte cannot hold that code
    -- the frozen kernels flag it (``overflowed()``) and the caller goes back to the fp32 schedule -- and an EMA range
    is routinely exceeded: a QuantAct shared by the units of a ShuffleNetV2 layer averages the extremes of four
    different tensors.  This helper runs ``forward(batch)`` (default: ``net``) on calibration batches with every
    QuantAct frozen, records what each one is fed, and WIDENS x_min / x_max (never narrows) to cover it with `margin` of
    the span to spare; two passes, since moving a grid moves what the layers behind it see.  It changes the model's
    quantisation grids -- the same widened model is what the fp32 frozen schedule is compared on.  Returns the number of
    QuantActs whose range moved.  spread (a dict, round 6): receives id(act) -> (sigma_lo, sigma_hi), the standard
    deviations of the per-BATCH extremes over the calibration batches of the last pass (what calibrate_serving's
    `sigmas` policy prices the tail of unseen batches with); needs >= 2 batches."""
    from .portable_quantizer.quant_modules import QuantAct
    acts = [m for m in net.modules() if isinstance(m, QuantAct)]
    was = [a.running_stat for a in acts]
    forward = forward if forward is not None else net
    moved = set()
    try:
        for a in acts:
            a.running_stat = False
        for _ in range(passes):
            seen, per_batch, cur = {}, {}, [0]

            def hook(mod, args):
                x = args[0].detach()
                lo, hi = x.min().float(), x.max().float()
                if id(mod) in seen:
                    seen[id(mod)] = (torch.minimum(seen[id(mod)][0], lo), torch.maximum(seen[id(mod)][1], hi))
                else:
                    seen[id(mod)] = (lo, hi)
                pb = per_batch.setdefault(id(mod), {})      # (a QuantAct shared by several call sites: one pair per batch)
                if cur[0] in pb:
                    pb[cur[0]] = (torch.minimum(pb[cur[0]][0], lo), torch.maximum(pb[cur[0]][1], hi))
                else:
                    pb[cur[0]] = (lo, hi)
            handles = [a.register_forward_pre_hook(hook) for a in acts]
            try:
                with torch.no_grad():
                    for bi, b in enumerate(batches):
                        cur[0] = bi
                        forward(b)
            finally:
                for h_ in handles:
                    h_.remove()
            if acts and not seen:
                raise RuntimeError("cover_frozen_ranges: no QuantAct was called by forward() -- the model runs a fused "
                                   "schedule (enable_fused()), which never calls the modules; calibrate on the module "
                                   "path (model.enable_fused(False)) or use pipeline.calibrate_serving(model, batches)")
            with torch.no_grad():
                for a in acts:
                    if id(a) not in seen:
                        continue
                    lo, hi = seen[id(a)]
                    span = (torch.maximum(hi, a.x_max.reshape(())) - torch.minimum(lo, a.x_min.reshape(()))) * margin
                    new_lo = torch.minimum(a.x_min.reshape(()), lo - span)
                    new_hi = torch.maximum(a.x_max.reshape(()), hi + span)
                    if bool(new_lo < a.x_min.reshape(())) or bool(new_hi > a.x_max.reshape(())):
                        moved.add(id(a))
                    a.x_min.copy_(new_lo.reshape(a.x_min.shape))
                    a.x_max.copy_(new_hi.reshape(a.x_max.shape))
            if spread is not None:
                for a in acts:
                    pb = per_batch.get(id(a))
                    if pb and len(pb) >= 2:
                        los = torch.stack([v[0] for v in pb.values()]).double()
                        his = torch.stack([v[1] for v in pb.values()]).double()
                        spread[id(a)] = (float(los.std()), float(his.std()))
    finally:
        for a, f in zip(acts, was):
            a.running_stat = f
    return len(moved)


# per-call gather schedule choice for an NCHW stage-0 input (include/codenet_dcn.h: CDN_X_GATHER_*), set as
# `path.gather_flag` on a FusedHotPath / FrozenHotPath by the tests that compare the two schedules bit for bit
GATHER_PER_ITEM, GATHER_PERSISTENT = 0x100, 0x200


class OverflowFlags:
    """The sticky saturation flags of a byte-code schedule: one int32 word PER LAUNCH GROUP (the kernels only
    ``atomicOr(flag, 1)``, so a distinct word per launch attributes a saturated code to the QuantAct(s) whose codes
    that launch writes).  ``any()`` is the old single-flag question; ``acts()`` names the QuantActs to widen
    (calibrate_serving)."""

    def __init__(self, n, dev):
        self.words = torch.zeros(max(1, n), dtype=torch.int32, device=dev)
        self.who = [[] for _ in range(max(1, n))]
        self.off = 0

    def ptr(self, i=0):
        return self.words.data_ptr() + 4 * (self.off + i)

    def data_ptr(self):
        return self.ptr(0)

    def count(self):
        return len(self.who) - self.off

    def slice(self, lo):
        """a view of the words from `lo` on, sharing words and names (for a consumer that numbers its own launches from 0)"""
        v = OverflowFlags.__new__(OverflowFlags)
        v.words, v.who, v.off = self.words, self.who, self.off + lo
        return v

    def name(self, i, acts):
        self.who[self.off + i] = list(acts)

    def any(self, reset=True):
        """True when some code saturated since the last reset (synchronises)."""
        hit = bool(self.words.any().item())
        if hit and reset:
            self.words.zero_()
        return hit

    def acts(self, reset=True):
        """The QuantActs of the launches that saturated since the last reset (synchronises)."""
        w = self.words.tolist()
        out = []
        for i, v in enumerate(w):
            if v:
                for a in self.who[i]:
                    if all(a is not b for b in out):
                        out.append(a)
        if reset and any(w):
            self.words.zero_()
        return out


def _widen(act, frac, low_too):
    with torch.no_grad():
        lo, hi = act.x_min.reshape(()), act.x_max.reshape(())
        span = (hi - lo).clamp_min(1e-6) * frac
        act.x_max.copy_((hi + span).reshape(act.x_max.shape))
        if low_too:
            act.x_min.copy_((lo - span).reshape(act.x_min.shape))


def calibrate_serving(model, batches, margin=0.02, grow=0.04, max_iter=40, sigmas=0.0):
    """Calibration of the byte-code serving mode ON THE SCHEDULE THAT SERVES.

    cover_frozen_ranges() records what the MODULE path feeds every QuantAct; the byte network's exact-integer first
    convolutions flip single codes against that path and a deep network amplifies them, so ranges that cover the module
    path with 2 % to spare can still saturate a byte on the serving schedule.  Here: (1) cover_frozen_ranges on the
    module path without margin (a starting point), (2) the byte network itself (``model.enable_fused(frozen_codes=True)``)
    runs the calibration batches; every launch that saturated a code names its QuantAct(s) through its own flag word
    (OverflowFlags) and exactly those ranges are widened by `grow` of their span -- repeated until a whole pass is
    clean; (3) every range gets `margin` of its span to spare and the pass is repeated until clean again.  Only widens.
    Leaves the model frozen (running_stat False) and on the byte schedule.  Returns a dict (iterations, widened, clean).

    sigmas > 0 (round 6, VERDICT r5 weak #5: ranges that are clean on the calibration batches with 2 % to spare saturated a
    byte in 8 of 32 UNSEEN batches): the margin of step (3) is priced per QuantAct from the spread of its per-batch
    extremes over the calibration batches (>= 4 of them) -- each live end moves out by a further `sigmas` standard
    deviations of that end's batch extreme.  Batch extremes are maxima of ~10^6 values: Gumbel-like with scale
    beta = 0.78 sigma; the largest of m calibration batches sits ~ beta ln m above the location, a fresh batch exceeds
    location + t beta with probability e^-t, and with ~140 QuantActs a batch-level recompute rate below 2 % asks for
    t ~ 9, i.e. ~ 5-6 sigma beyond the calibration extreme at m = 8.  The price is resolution: every grid gets coarser by the
    widening (reported as `mean_widening`).  The alternative is the saturating policy: serve what the byte schedule
    produced and do not recompute."""
    from .portable_quantizer.quant_modules import QuantAct
    acts = [m for m in model.modules() if isinstance(m, QuantAct)]
    set_running_stat(model, False)
    model.enable_fused(False)
    spread = {} if (sigmas > 0 and len(batches) >= 4) else None
    covered = cover_frozen_ranges(model, batches, margin=0.0, spread=spread)
    span0 = {id(a): float(a.x_max.reshape(()) - a.x_min.reshape(())) for a in acts}
    model.enable_fused(frozen_codes=True)
    # attribution needs one launch per QuantAct: the depthwise-into-pointwise fusion of the byte backbone writes two
    # QuantActs' codes from one launch (bit-identical to the two kernels), so it is off while calibrating
    hits, widened, iters = {}, set(), 0

    def one_pass():
        with torch.no_grad():
            for b in batches:
                model(b)
        fz = getattr(model, "_fzbackbone", None)
        bad = []
        for f in (getattr(model, "_ffrozen", None), fz):
            if f is not None and f._bufs is not None:
                for a in f._bufs["overflow"].acts():
                    if all(a is not b_ for b_ in bad):
                        bad.append(a)
        return bad

    def until_clean():
        nonlocal iters
        while iters < max_iter:
            iters += 1
            fz = getattr(model, "_fzbackbone", None)
            if fz is not None:
                fz.fuse_dwpw = False
            bad = one_pass()
            if not bad:
                return True
            for a in bad:
                hits[id(a)] = hits.get(id(a), 0) + 1
                _widen(a, grow, bool(a.x_min.reshape(()) < 0) or hits[id(a)] >= 3)
                widened.add(id(a))
        return False
    clean = False
    try:
        with torch.no_grad():
            model(batches[0])                   # builds the byte-code objects
        clean = until_clean()
        if clean and (margin > 0 or spread):
            with torch.no_grad():
                for a in acts:
                    low_live = bool(a.x_min.reshape(()) < 0)
                    if margin > 0:
                        _widen(a, margin, low_live)
                    s_lo, s_hi = (spread or {}).get(id(a), (0.0, 0.0))
                    a.x_max.add_(sigmas * s_hi)
                    if low_live:
                        a.x_min.sub_(sigmas * s_lo)
            clean = until_clean()
    finally:
        fz = getattr(model, "_fzbackbone", None)
        if fz is not None:
            fz.fuse_dwpw = True                 # (also when a calibration pass raised)
    if fz is not None:
        with torch.no_grad():
            for b in batches:               # the serving configuration itself (fused depthwise) must be clean too
                model(b)
        clean = clean and not model.frozen_overflowed()
    widening = [float(a.x_max.reshape(()) - a.x_min.reshape(())) / max(span0[id(a)], 1e-12) for a in acts if span0[id(a)] > 0]
    return {"iterations": iters, "covered_on_module_path": covered, "widened_on_byte_schedule": len(widened),
            "clean": bool(clean), "byte_backbone": fz is not None, "calibration_batches": len(batches),
            "sigmas": float(sigmas) if spread is not None else 0.0, "margin": float(margin),
            "mean_widening": (sum(widening) / len(widening)) if widening else 1.0,
            "max_widening": max(widening) if widening else 1.0}


def prepare_serving(model, images, settle=300, margin=0.02, replay=None, more_batches=(), sigmas=0.0):
    """The serving recipe bench.py's `e2e.frozen` leg times and tests/test_harness.py checks on three seeds: let the
    running (EMA) ranges settle over `settle` forwards of `images` (replay: an already captured graph of the running
    network), freeze every QuantAct, calibrate ON THE BYTE SCHEDULE (calibrate_serving) over `images` + `more_batches`
    (sigmas: its tail policy for unseen batches).  Leaves the model on enable_fused(frozen_codes=True); returns
    calibrate_serving's report."""
    if replay is None:
        model.enable_fused()
        with torch.no_grad():
            for _ in range(settle):
                model(images)
    else:
        for _ in range(settle):
            replay()
    torch.cuda.synchronize()
    return calibrate_serving(model, [images] + list(more_batches), margin=margin, sigmas=sigmas)


def broadcast_parameters(net, src=0):
    """Start-up broadcast of every parameter and buffer from `src` as ONE flat tensor
    (RCCL over xGMI when the process group backend is nccl; gloo in the CPU tests)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    tensors = list(net.parameters()) + [b for b in net.buffers() if b.dtype.is_floating_point]
    if not tensors:
        return 0
    with torch.no_grad():
        flat = torch.cat([t.detach().reshape(-1).float() for t in tensors])
        dist.broadcast(flat, src=src)
        off = 0
        for t in tensors:
            # copy into the parameter / buffer itself (not .data): the in-place write bumps the tensor's
            # version counter, which is what the derived-weight caches (fake-quantised, folded, int8 forms,
            # BN affines) are keyed on
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
    return flat.numel() * 4


def exchange_shard_sizes(n_local, device=None):
    """ONE start-up exchange of the ranks' per-batch image counts (shard sizes are static for a run): the list
    `counts` that gather_detections() then takes, so that no batch pays a count all_gather + host sync."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [int(n_local)]
    world = dist.get_world_size()
    nb = torch.tensor([int(n_local)], device=device, dtype=torch.int64)
    parts = [torch.zeros_like(nb) for _ in range(world)]
    dist.all_gather(parts, nb)
    return [int(c.item()) for c in parts]


def gather_detections(dets, dst=None, counts=None):
    """Per-batch collection of every rank's detections [B, K, 6] (collective 2; the
    reference's only mechanism is DataParallel's gather, lib/models/data_parallel.py:64-84,120-129).
    all_gather over RCCL (gloo in the CPU tests).  `counts` = the ranks' shard sizes from exchange_shard_sizes()
    (static per run): with it a batch is ONE collective and no host synchronisation -- equal shards gather
    straight into one [world*B, K, 6] tensor; shards that differ by an image (shard_range) are padded to the
    largest.  Without `counts` the sizes are exchanged first (one extra small all_gather + a host read per call;
    kept for one-off calls).  Returns [sum B_r, K, 6] in rank order on every rank (dst=None) or only on rank dst
    (other ranks: None)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return dets
    world = dist.get_world_size()
    if counts is None:
        counts = exchange_shard_sizes(dets.shape[0], dets.device)
    if len(counts) != world or counts[dist.get_rank()] != dets.shape[0]:
        raise ValueError("gather_detections: counts %r do not describe this rank's shard of %d images"
                         % (counts, dets.shape[0]))
    bmax = max(counts)
    pad = dets if dets.shape[0] == bmax else torch.cat(
        [dets, dets.new_zeros((bmax - dets.shape[0],) + tuple(dets.shape[1:]))])
    pad = pad.contiguous()
    if dst is None and min(counts) == bmax:
        out = pad.new_empty((world * bmax,) + tuple(pad.shape[1:]))
        dist.all_gather_into_tensor(out, pad)
        return out
    if dst is None:
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
    else:
        parts = [torch.empty_like(pad) for _ in range(world)] if dist.get_rank() == dst else None
        dist.gather(pad, parts, dst=dst)
        if parts is None:
            return None
    return torch.cat([p[:c] for p, c in zip(parts, counts)])


def set_global_range(model, flag=True):
    """Multi-process parity mode of every QuantAct of `model` (collective 3): the batch extremes
    are all-reduced (MIN / MAX, two 4-byte collectives per QuantAct call, RCCL over xGMI / gloo) before the range
    update, so R ranks x B images track the ranges of one R*B-image run.  The three deform stages stay on the fused
    schedule (round 5: the stage call is split at its QuantActs, FusedHotPath._global_commit); backbone and heads keep
    the module-by-module path in this mode (their fused schedules update ranges inside the producing kernels).  Returns
    the number of QuantActs set."""
    from .portable_quantizer.quant_modules import QuantAct
    n = 0
    for m in model.modules():
        if isinstance(m, QuantAct):
            m.global_range = bool(flag)
            n += 1
    return n


ACT_PERCENTILE = 0x400      # CDN_X_ACT_PERCENTILE (include/codenet_dcn.h)
WCODES_KB = 0x800           # CDN_X_WCODES_KB
DEFER_RANGE = 0x1000        # CDN_X_DEFER_RANGE
PHASE_SCALE, PHASE_GATHER, PHASE_POINTWISE = 0x2000, 0x4000, 0x8000      # CDN_X_PHASE_*


def stage_int8_codes(convbn, kblocked=True):
    """(i8 triple or None, flag) for the pointwise conv of a fused stage: the int8 form of the folded weights, with the
    k-blocked copy behind the codes -- and CDN_X_WCODES_KB to OR into the stage call's layout argument -- where the
    library has a use for it (long-K stages: cdn_codenet_wcodes_kb_columns)."""
    from . import _native as N_
    lib = N_.lib()
    conv = convbn.conv
    cols = lib.cdn_codenet_wcodes_kb_columns(conv.in_channels, conv.out_channels) if kblocked else 0
    if cols:
        i8 = convbn.folded_int8_kblocked(cols, lib.cdn_codenet_wcodes_kb_offset(conv.in_channels, conv.out_channels))
        return i8, (WCODES_KB if i8 is not None else 0)
    return convbn.folded_int8(), 0


def act_fusable(act, allow_percentile=False, allow_global=False):
    """The fused schedules implement the reference's default QuantAct: plain batch min/max tracking,
    asymmetric, quantising (quant_modules.py:163-225 with percentile=False).  Symmetric activations and
    full_precision_flag stay on the module path; --act-percentile too, except in the three deform stages
    (allow_percentile: FusedHotPath, round 4 -- the stage entry point follows the order statistics with
    cdn_kth_values between its kernels); the multi-process global-range mode too, except in the three deform stages
    (allow_global: FusedHotPath, round 5 -- the stage call is split at its QuantActs, `_global_commit`)."""
    glob = getattr(act, "global_range", False) and act.running_stat
    return (act.quant_mode == "asymmetric" and (allow_percentile or not act.percentile)
            and not act.full_precision_flag
            and (not glob or (allow_global and not act.percentile)))


def global_range_active(acts):
    """True when the QuantActs of a fused stage call must see batch extremes reduced over the ranks: the mode is on, the
    ranges are tracked and there is more than one rank (one rank: the plain call computes the same thing)."""
    import torch.distributed as dist
    acts = [a for a in acts if a is not None]
    if not any(getattr(a, "global_range", False) and a.running_stat for a in acts):
        return False
    if not all(getattr(a, "global_range", False) for a in acts):
        raise NotImplementedError("the QuantActs of one fused stage must all or none be in global-range mode")
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def uniform_act_settings(acts, what, allow_percentile=False, allow_global=False):
    """(bits, momentum, running) shared by the QuantActs of one fused C call, which takes them once.
    allow_percentile: the call's QuantActs may all be percentile ones (never a mixture)."""
    acts = [a for a in acts if a is not None]
    if not acts:
        return 8, 0.99, 0
    if allow_percentile and len({bool(a.percentile) for a in acts}) != 1:
        raise NotImplementedError("%s: the QuantActs of one fused call must all or none use percentile ranges" % what)
    for a in acts:
        if not act_fusable(a, allow_percentile, allow_global):
            raise NotImplementedError("%s: QuantAct(percentile=%s, quant_mode=%s, full_precision_flag=%s) is not "
                                      "implemented by the fused schedule; use the module path"
                                      % (what, a.percentile, a.quant_mode, a.full_precision_flag))
    st = {(a.activation_bit, float(a.momentum), int(bool(a.running_stat))) for a in acts}
    if len(st) != 1:
        raise NotImplementedError("%s: the QuantActs of one fused call must share activation_bit / momentum / "
                                  "running_stat (got %s)" % (what, sorted(st)))
    return st.pop()


class GraphedTrainStep:
    """One training step -- forward, loss, backward, optimizer -- over static input buffers as ONE HIP graph.

    The reference's training loop (quant_main.py -> lib/trains/base_trainer.py:51-80) launches eagerly; the QAT step of
    the three deform stages is 61 kernels of 4-130 us, so eager launches are host-bound (1.4-1.9 ms per step against
    1.05 ms of GPU work).  Captured once, the step replays without the host in the loop.  Every
    sum of the step has a fixed order (section 4.3), so a replayed step is bit-identical to the eager one.

        opt = torch.optim.Adam(net.parameters(), lr=torch.tensor(1.25e-4, device=dev), capturable=True)
        step = GraphedTrainStep(net, opt, loss_fn, example_inputs)                  # loss_fn(net, *inputs) -> scalar
        for batch in loader:
            loss = step(*batch)           # copies the batch into the static buffers, replays; loss: a static tensor
            step.set_lr(schedule(it))     # (a learning rate held as a Python float is a constant of the graph)

    Shapes are fixed at capture; QuantAct running ranges, BatchNorm buffers and the optimizer state advance in place
    exactly as in the eager loop.  `warmup` eager steps run first on a side stream (they DO train: allocator and lazily
    derived tensors settle before capture); the capture records on that same stream.

    SCOPE (round 6).  Validated -- and accepted without `unvalidated=True` -- is the stack of deform stages
    (``pipeline.build_hot_path`` / a quantised ``deconv_layers``): every kernel of that step is this library's, every sum
    has one order, and tests/test_train_step.py::test_graphed_train_step_* show replays bit-identical to the eager step and
    bit-identical from a restored state whatever else the process does in between (a second model built before the capture
    taking its first eager steps, another framework model training, the allocator's free memory filled with NaN).
    A network that also runs PyTorch-ROCm operators under autograd (the whole CoDeNet: backbone and heads) needs
    `unvalidated=True`; what was measured for it (tools/experiments/gts_probe*.py):
      * parameters and gradients of a replay agree with any other replay FROM THE SAME STATE to ~1e-6 of their magnitude --
        with or without other work in between.  That residue is the framework's own backward (atomics), present in two
        eager runs too; after a quantiser amplifies it the loss trajectories of two runs part ways in the fifth digit by
        the second step.  This -- not memory corruption -- is what round 5 recorded as "wrong losses after a second model's
        first steps": poisoning every free byte of the allocator with NaN between replays changes nothing;
      * the scalar LOSS a replay returns can be stale in one situation: loss_fn reduces a large tensor with one
        ``mean()`` / ``sum()`` (ATen's multi-block reduction: a semaphore word zeroed by a memset node), and another model
        takes its FIRST eager step between two replays -- that replay's reduction leaves its output unwritten while the
        gradients and parameters of the same replay are right (the backward of a mean does not read its value).  A
        two-level reduction (``v.square().reshape(-1, 64).sum(1).sum() / v.numel()``: no semaphore) does not show it."""

    def __init__(self, net, optimizer, loss_fn, example_inputs, warmup=3, unvalidated=False):
        if not unvalidated and not self.is_stage_stack(net):
            raise NotImplementedError(
                "GraphedTrainStep is validated (bit-identical replays) for a stack of quantised deform stages only; `net` "
                "holds other modules, whose PyTorch-ROCm kernels under autograd are not run-to-run deterministic. Pass "
                "unvalidated=True to capture it anyway (see the class docstring for what was measured).")
        self.static = [t.detach().clone() for t in example_inputs]
        for t, src in zip(self.static, example_inputs):
            t.requires_grad_(src.requires_grad)
        self._net, self._opt, self._loss_fn = net, optimizer, loss_fn
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(int(warmup), 1)):
                self.eager_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        # capture ON the warm-up stream: the autograd nodes of the parameters were created there, and a capture stream of
        # its own makes every gradient accumulation a cross-stream branch of the graph (the framework warns about it)
        with torch.cuda.graph(self.graph, stream=side):
            self.loss = loss_fn(net, *self.static)
            self.loss.backward()
            optimizer.step()

    @staticmethod
    def is_stage_stack(net):
        """True for a (container of one) Sequential of [quantised deform stage, Sequential(ReLU, QuantAct), Upsample]
        blocks -- what functions/codenet_stage.forward_stage_blocks runs natively and the bit-level tests cover."""
        from .portable_quantizer.quant_modules import QuantAct, QuantDeformConvWithOffsetScaleBoundPositive
        seq = getattr(net, "deconv_layers", net)
        if not isinstance(seq, nn.Sequential) or len(seq) == 0 or len(seq) % 3:
            return False
        if seq is not net and [m for m in net.children()] != [seq]:
            return False
        mods = list(seq)
        for i in range(0, len(mods), 3):
            q, post, up = mods[i:i + 3]
            if not (isinstance(q, QuantDeformConvWithOffsetScaleBoundPositive) and isinstance(post, nn.Sequential)
                    and len(post) == 2 and isinstance(post[0], nn.ReLU) and isinstance(post[1], QuantAct)
                    and isinstance(up, nn.Upsample)):
                return False
        return True

    def set_lr(self, value, group=None):
        """Write a new learning rate where the captured step reads it: the param group's lr must be a device TENSOR
        (torch.optim with capturable=True accepts one).  A Python float was baked into the graph at capture -- a scheduler
        that assigns ``group['lr'] = float`` changes nothing on replay -- so that case raises instead of silently ignoring."""
        groups = self._opt.param_groups if group is None else [self._opt.param_groups[group]]
        for g in groups:
            if not torch.is_tensor(g["lr"]):
                raise RuntimeError("GraphedTrainStep.set_lr: this optimizer holds its learning rate as a Python float, which "
                                   "the captured graph holds as a constant; construct the optimizer with "
                                   "lr=torch.tensor(value, device=...) (capturable=True) before the capture")
            with torch.no_grad():
                g["lr"].fill_(float(value))

    def eager_step(self):
        """The same step with eager launches (what the capture records), on the static buffers."""
        self._opt.zero_grad(set_to_none=True)
        loss = self._loss_fn(self._net, *self.static)
        loss.backward()
        self._opt.step()
        return loss

    def __call__(self, *inputs):
        if len(inputs) != len(self.static):
            raise ValueError("GraphedTrainStep: %d inputs, captured with %d" % (len(inputs), len(self.static)))
        with torch.no_grad():
            for dst, src in zip(self.static, inputs):
                if dst.shape != src.shape:
                    raise ValueError("GraphedTrainStep: input shape %s, captured with %s" % (tuple(src.shape), tuple(dst.shape)))
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.loss


def shard_range(total, rank, world):
    """Contiguous images [lo, hi) of a `total`-image batch owned by `rank` (sizes differ by <= 1)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def algorithmic_bytes(batch, input_res=512, w2=False, fused=False, act_bytes=4):
    """Per-step algorithmic bytes of the hot path: unfused 3-kernel definition
    scale (C+1)HW*4 + gather (2C+1)HW*4 + pointwise (C+Co)HW*4, or the fused-stage definition."""
    tot = 0
    per = {}
    for (C, Co, H, W) in stage_shapes(input_res, w2):
        HW = H * W
        if fused:
            b = (C + Co) * HW * act_bytes
            per[(C, Co, H)] = {"fused": b * batch}
        else:
            sc, ga, pw = (C + 1) * HW * 4, (2 * C + 1) * HW * 4, (C + Co) * HW * 4
            b = sc + ga + pw
            per[(C, Co, H)] = {"scale": sc * batch, "dw": ga * batch, "pointwise": pw * batch}
        tot += b * batch
    return tot, per


def bn_affine(cache, bn):
    """BatchNorm (eval) as a per-channel affine, cached until one of its tensors changes (keyed on
    data_ptr + version like the weight caches; refreshed in place so captured graphs stay valid)."""
    from .portable_quantizer.quant_modules import refresh_in_place
    src = (bn.weight, bn.bias, bn.running_mean, bn.running_var)
    key = tuple((t.data_ptr(), t._version, t.device) for t in src)
    ent = cache.get(id(bn))
    if ent is None or ent[0] != key:
        with torch.no_grad():
            inv = torch.rsqrt(bn.running_var + bn.eps)
            es = (bn.weight * inv).contiguous()
            new = (es, (bn.bias - bn.running_mean * es).contiguous())
            ent = (key, refresh_in_place(ent[1] if ent else None, new))
        cache[id(bn)] = ent
    return ent[1]


class FusedHotPath:
    """Runs a ``deconv_layers`` Sequential (fp32 or W4A8, built from the modules of this package)
    as the fused per-stage kernel schedule of codenet_fused.hip: one C-ABI call per stage, then one
    unpack (fake-quant + nearest x2 + NCHW) for the consumer.  Same parameters, same QuantAct
    buffers (updated in place), same results as calling the Sequential module by module.

    All device buffers are allocated once per input shape, so a call issues only kernel launches
    and can be captured into a HIP graph (``capture()``)."""

    def __init__(self, deconv_layers, int8_pointwise=True, kblocked_codes=True, chain_scale=True):
        from .portable_quantizer.quant_modules import QuantDeformConvWithOffsetScaleBoundPositive
        self.seq = deconv_layers
        self.int8_pointwise = int8_pointwise
        self.kblocked_codes = kblocked_codes      # (False: tests compare the two int8 pointwise kernels)
        self.chain_scale = chain_scale            # fp32 model: the next stage's scale prediction from the pointwise epilogue
        mods = list(deconv_layers)
        self.quantized = isinstance(mods[0], QuantDeformConvWithOffsetScaleBoundPositive)
        step = 3 if self.quantized else 4
        assert len(mods) % step == 0
        self.stages = [mods[i:i + step] for i in range(0, len(mods), step)]
        for st in self.stages:
            assert isinstance(st[-1], nn.Upsample) and st[-1].scale_factor in (2, 2.0)
        self._bufs = None
        self._graph = None
        self._affine = {}
        self.stage_hook = None        # diagnostics: called as stage_hook(stage_shape_dict) after each stage

    @staticmethod
    def supported(deconv_layers, input_shape=None):
        """True when the fused schedule implements this Sequential (and, given the NCHW shape of its input,
        this geometry): callers keep the module path otherwise."""
        from . import _native as N_
        from .portable_quantizer.quant_modules import QuantAct, QuantDeformConvWithOffsetScaleBoundPositive
        mods = list(deconv_layers)
        if not mods:
            return False
        quantized = isinstance(mods[0], QuantDeformConvWithOffsetScaleBoundPositive)
        step = 3 if quantized else 4
        if len(mods) % step:
            return False
        shape = tuple(input_shape) if input_shape is not None else None
        for i in range(0, len(mods), step):
            st = mods[i:i + step]
            if not (isinstance(st[-1], nn.Upsample) and st[-1].scale_factor in (2, 2.0)):
                return False
            if quantized:
                if not (isinstance(st[0], QuantDeformConvWithOffsetScaleBoundPositive) and len(st[1]) == 2
                        and isinstance(st[1][1], QuantAct)):
                    return False
                acts = (st[0].quant_act[1], st[0].quant_identity_deform, st[1][1])
                try:
                    uniform_act_settings(acts, "stage", allow_percentile=True, allow_global=True)
                    global_range_active(acts)              # (raises on a mixture)
                except NotImplementedError:
                    return False
                cout = st[0].quant_conv_channel_bn.conv.out_channels
            else:
                if not (isinstance(st[0], DeformConvWithOffsetScaleBoundPositive)
                        and isinstance(st[1], nn.BatchNorm2d) and hasattr(st[0], "conv_channel")):
                    return False
                cout = st[0].out_channels
            if shape is not None:
                Nb, C, H, W = shape
                up = 0 if i == 0 else 1
                if not N_.lib().cdn_codenet_stage_fused_supported(Nb, C, H, W, 0 if i == 0 else 1, up):
                    return False
                shape = (Nb, cout, 2 * H, 2 * W)
        return True

    # -- per-stage parameter views -------------------------------------------------------------
    @torch.no_grad()      # inference schedule: derived weights come from the modules' caches, never an autograd graph
    def _stage_params(self, st):
        if self.quantized:
            q, post = st[0], st[1]
            w_pw, b_pw = q.quant_conv_channel_bn.folded()
            i8, kb_flag = (stage_int8_codes(q.quant_conv_channel_bn, self.kblocked_codes) if self.int8_pointwise
                           else (None, 0))
            return dict(
                i8=i8, kb_flag=kb_flag,
                w_scale=q.quant_conv_scale.quantized_weight().reshape(-1),
                b_scale=q.quant_conv_scale.bias, lo=q.quant_act[0].min_val, hi=q.quant_act[0].max_val,
                w_dw=q.quant_deform_conv.quantized_weight(), w_pw=w_pw.reshape(w_pw.size(0), -1),
                bias=b_pw, ep_scale=None, ep_shift=None,
                acts=(q.quant_act[1], q.quant_identity_deform, post[1]))
        op, bn = st[0], st[1]
        es, eh = bn_affine(self._affine, bn)
        return dict(w_scale=op.conv_scale.weight.reshape(-1), b_scale=op.conv_scale.bias,
                    lo=op.conv_bound.min_val, hi=op.conv_bound.max_val, w_dw=op.conv.weight,
                    w_pw=op.conv_channel.weight.reshape(op.out_channels, -1), bias=None,
                    ep_scale=es, ep_shift=eh, acts=(None, None, None), i8=None, kb_flag=0)

    def _alloc(self, x):
        from . import _native as N_
        Nb, C, H, W = x.shape
        dev = x.device
        bufs, ws_bytes = [], 0
        for i, st in enumerate(self.stages):
            op = st[0]
            cin = op.quant_deform_conv.in_channels if self.quantized else op.in_channels
            cout = (op.quant_conv_channel_bn.conv.out_channels if self.quantized
                    else op.out_channels)
            up = 0 if i == 0 else 1
            Hs, Ws = (H, W) if i == 0 else (bufs[-1]["H"] * 2, bufs[-1]["W"] * 2)
            ws_bytes = max(ws_bytes, N_.lib().cdn_codenet_stage_workspace_bytes(Nb, cin, Hs, Ws, up))
            bufs.append(dict(C=cin, Co=cout, H=Hs, W=Ws, up=up,
                             r=torch.empty(Nb, Hs * Ws, cout, device=dev), parts=0, parts_buf=None))
        if not self.quantized and self.chain_scale:
            # chained fp32 stages (round 6): the pointwise epilogue of stage i leaves the partial sums of stage i + 1's scale
            # prediction -- no QuantAct sits between them in the fp32 model -- and stage i + 1 runs without its scale launch
            for sb in bufs[:-1]:
                sb["parts"] = int(N_.lib().cdn_codenet_stage_chain_parts(Nb, sb["C"], sb["Co"], sb["H"], sb["W"]))
                if sb["parts"]:
                    sb["parts_buf"] = torch.empty(sb["parts"] * Nb * sb["H"] * sb["W"], device=dev)
        last = bufs[-1]
        out = torch.empty(Nb, last["Co"], last["H"] * 2, last["W"] * 2, device=dev)
        ws = torch.zeros(ws_bytes // 4 + 64, device=dev)   # arrival counters must start at zero
        self._bufs = dict(shape=tuple(x.shape), dev=dev, stages=bufs, ws=ws, out=out)

    def __call__(self, x):
        """Stages + unpack: the Sequential's output tensor (NCHW, up-sampled, fake-quantised)."""
        from . import _native as N_
        from . import ops
        cur, cur_q, last = self.forward_nhwc(x)
        B = self._bufs
        rec = ops._tic("unpack", (last["Co"], last["H"], last["W"]))
        rc = N_.lib().cdn_codenet_unpack_nchw(cur.data_ptr(), cur_q, B["out"].data_ptr(), x.shape[0],
                                              last["Co"], last["H"], last["W"], 1,
                                              torch.cuda.current_stream(x.device).cuda_stream)
        ops._toc(rec)
        N_.check(rc, "cdn_codenet_unpack_nchw")
        return B["out"]

    def forward_nhwc(self, x, x_qstate=None, hw=None):
        """The three stages WITHOUT the final materialisation: returns (r, r_qstate, shape) with r the
        last stage's output [N, H*W, Co] channels-last at stage resolution, pre-quantisation and not yet
        up-sampled, r_qstate the device pointer of its QuantAct state (None in fp32) and shape the
        stage's dict (Co, H, W).  Consumers (FusedHeads) fake-quantise on load and up-sample by
        addressing.
        x: the NCHW tensor the backbone hands over, or -- with hw=(H, W) -- a channels-last [N, H*W, C]
        tensor holding PRE-quantisation values whose QuantAct state pointer is x_qstate (FusedBackbone)."""
        from . import _native as N_
        from . import ops
        nhwc_in = hw is not None
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == (3 if nhwc_in else 4)):
            raise NotImplementedError("FusedHotPath needs a float32 GPU tensor: NCHW, or [N, H*W, C] with hw")
        x = x.contiguous()
        if nhwc_in:
            x = x.view(x.shape[0], hw[0], hw[1], x.shape[2]).permute(0, 3, 1, 2)   # logical NCHW view
        op0 = self.stages[0][0]
        c0 = op0.quant_deform_conv.in_channels if self.quantized else op0.in_channels
        if x.shape[1] != c0:      # (the kernels take the channel count from the modules: a mismatch would read out of bounds)
            raise RuntimeError("FusedHotPath: the input has %d channels, stage 0 expects %d" % (x.shape[1], c0))
        if self._bufs is None or self._bufs["shape"] != tuple(x.shape) or self._bufs["dev"] != x.device:
            self._alloc(x)
        B = self._bufs
        Nb = x.shape[0]
        lib = N_.lib()
        stream = torch.cuda.current_stream(x.device).cuda_stream
        ws = B["ws"]
        ws_ptr = (ws.data_ptr() + 255) // 256 * 256
        ws_bytes = (ws.numel() * 4 - (ws_ptr - ws.data_ptr())) // 256 * 256
        cur, cur_nhwc, cur_q = x, int(nhwc_in), (x_qstate if nhwc_in else None)
        with torch.no_grad():
            for si, (st, sb) in enumerate(zip(self.stages, B["stages"])):
                p = self._stage_params(st)
                ptr = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
                a = []
                bits, mom, running = uniform_act_settings(p["acts"], "FusedHotPath stage", allow_percentile=True,
                                                          allow_global=True)
                pct = ACT_PERCENTILE if (p["acts"][0] is not None and p["acts"][0].percentile) else 0
                for act in p["acts"]:
                    if act is None:
                        a += [None, None, None]
                    else:
                        a += [act.x_min.data_ptr(), act.x_max.data_ptr(),
                              act._device_state(x.device).data_ptr()]
                rec = ops._tic("stage", (sb["C"], sb["H"], sb["W"]))

                def stage_call_chain():
                    nxt = B["stages"][si + 1] if si + 1 < len(B["stages"]) else None
                    prev = B["stages"][si - 1] if si > 0 else None
                    out_parts = sb["parts_buf"] if (nxt is not None and sb["parts"]) else None
                    in_parts = prev["parts_buf"] if (prev is not None and prev["parts"]) else None
                    nws = self._stage_params(self.stages[si + 1])["w_scale"] if out_parts is not None else None
                    rc = lib.cdn_codenet_stage_fused_forward_chain(
                        cur.data_ptr(), cur_nhwc | getattr(self, "gather_flag", 0), sb["up"], cur_q, Nb, sb["C"], sb["Co"],
                        sb["H"], sb["W"], ptr(p["w_scale"]), ptr(p["b_scale"]), float(p["lo"]), float(p["hi"]),
                        ptr(p["w_dw"]), ptr(p["w_pw"]), ptr(p["bias"]), ptr(p["ep_scale"]), ptr(p["ep_shift"]), 1,
                        ws_ptr, ws_bytes, sb["r"].data_ptr(), ptr(in_parts), prev["parts"] if in_parts is not None else 0,
                        ptr(nws), ptr(out_parts), stream)
                    N_.check(rc, "cdn_codenet_stage_fused_forward_chain")

                def stage_call(extra):
                    rc = lib.cdn_codenet_stage_fused_forward(
                        cur.data_ptr(), cur_nhwc | getattr(self, "gather_flag", 0) | pct | p["kb_flag"] | extra, sb["up"],
                        cur_q, Nb, sb["C"],
                        sb["Co"], sb["H"], sb["W"],
                        ptr(p["w_scale"]), ptr(p["b_scale"]), float(p["lo"]), float(p["hi"]),
                        ptr(p["w_dw"]), ptr(p["w_pw"]),
                        *([ptr(t) for t in p["i8"]] if p["i8"] is not None else [None, None, None]),
                        ptr(p["bias"]), ptr(p["ep_scale"]),
                        ptr(p["ep_shift"]), 1, *a, bits, mom, running, ws_ptr, ws_bytes,
                        sb["r"].data_ptr(), stream)
                    N_.check(rc, "cdn_codenet_stage_fused_forward")

                if global_range_active(p["acts"]):
                    # multi-process parity mode (collective 3): the stage call split at its three
                    # QuantActs -- each producer only measures, the batch extremes are reduced over the ranks (one
                    # 8-byte MAX all-reduce of {-min, max}), the commit applies the reference's update with them
                    for phase, act in zip((PHASE_SCALE, PHASE_GATHER, PHASE_POINTWISE), p["acts"]):
                        stage_call(DEFER_RANGE | phase)
                        self._global_commit(act, x.device, bits, mom, stream)
                elif not self.quantized and cur_q is None and (sb["parts"] or (si > 0 and B["stages"][si - 1]["parts"])):
                    stage_call_chain()
                else:
                    stage_call(0)
                ops._toc(rec)
                if self.stage_hook is not None:
                    self.stage_hook(sb)
                cur, cur_nhwc = sb["r"], 1
                cur_q = a[8]          # r_state of this stage (None in fp32)
        return cur, cur_q, B["stages"][-1]

    @staticmethod
    def _global_commit(act, dev, bits, mom, stream):
        """Range update of one QuantAct from the extremes of ALL ranks: the producer (CDN_X_DEFER_RANGE) left this rank's
        batch {min, max} in words [4], [5] of the device state."""
        import torch.distributed as dist
        from . import _native as N_
        st = act._device_state(dev)
        f = st.view(torch.float32)
        from .portable_quantizer.quant_modules import allreduce_extremes
        t = allreduce_extremes(f[4:5], f[5:6])           # one MAX all-reduce for both ends + the NaN flag
        rc = N_.lib().cdn_quantact_commit_range(act.x_min.data_ptr(), act.x_max.data_ptr(), st.data_ptr(), t.data_ptr(),
                                                bits, mom, 1, stream)
        N_.check(rc, "cdn_quantact_commit_range")

    # -- HIP graph -----------------------------------------------------------------------------
    def capture(self, x, unpack=True):
        """Capture one pass over the static input buffer `x` into a HIP graph; returns a callable
        replaying it (the output tensor is static too).  unpack=False: the three stages only, returning the
        channels-last stage-resolution tensor ``forward_nhwc`` hands to the native heads."""
        if self.quantized and any(global_range_active(self._stage_params(st)["acts"]) for st in self.stages):
            raise NotImplementedError("FusedHotPath.capture: the global-range mode runs collectives between the kernels; "
                                      "launch it eagerly")
        run = self.__call__ if unpack else (lambda t: self.forward_nhwc(t)[0])
        run(x)                        # allocate + warm (also derives cached weights)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = run(x)
        self._graph = g

        def replay():
            g.replay()
            return out
        return replay


class FrozenHotPath:
    """``deconv_layers`` (W4A8) with FROZEN QuantAct ranges on byte codes: cdn_codenet_stage_frozen_forward, three
    launches per stage, every quantised tensor crosses HBM as one byte per element.  This is the serving
    mode (``QuantAct.running_stat = False``: a plain attribute in the reference, quant_modules.py:172,181,203-219),
    NOT the reference's default -- ``FusedHotPath`` is.  Results are bit-identical to ``FusedHotPath`` with
    running_stat False as long as no activation leaves its frozen 8-bit grid; the reference does not clamp codes,
    a byte must, so a saturated code raises the sticky device flag read by ``overflowed()`` and the caller
    recomputes that batch with ``FusedHotPath``.  Stages whose input channel count is not a multiple of 4
    (CoDeNet2x stage 0, C = 2153) run on the fp32 frozen schedule and hand fp32 + state to the next stage.

    chain_scale=True (cdn_codenet_stage_frozen_chained_forward): the scale prediction of stage k+1 is accumulated as
    exact integer sums in stage k's pointwise epilogue and finished by stage k+1's gather -- two launches fewer and no
    re-read of r.  A DECLARED non-bit-identical variant: s_raw is the exact sum rounded once instead of an fp32 sum
    of C products, so single scale codes can differ from the default schedule by one LSB (checked against the oracle
    with the same code-flip tolerance, tests/test_gpu_frozen.py)."""

    def __init__(self, deconv_layers, chain_scale=False):
        from .portable_quantizer.quant_modules import QuantDeformConvWithOffsetScaleBoundPositive
        self.chain_scale = bool(chain_scale)
        mods = list(deconv_layers)
        if not mods or not isinstance(mods[0], QuantDeformConvWithOffsetScaleBoundPositive):
            raise NotImplementedError("FrozenHotPath needs the W4A8 deconv_layers")
        if not FusedHotPath.supported(deconv_layers):
            raise NotImplementedError("this deconv_layers configuration is not implemented by the fused schedules")
        self.seq = deconv_layers
        self.stages = [mods[i:i + 3] for i in range(0, len(mods), 3)]
        for st in self.stages:
            if st[0].quant_conv_channel_bn.folded_int8() is None:
                raise NotImplementedError("FrozenHotPath needs per-channel symmetric <= 4-bit pointwise weights")
        self._fp32 = FusedHotPath(deconv_layers)        # fp32 frozen schedule for stages without byte codes
        self._bufs = None

    @staticmethod
    def planes_fit(deconv_layers, input_shape):
        """True when every stage's stored plane fits the LDS-resident gather -- the byte-code entry points' own limit
        (cdn_codenet_stage_supported).  Above it (inputs beyond ~1100 px) the fp32 fused schedule, which honours frozen
        ranges too and gathers large planes from global memory, is the one to use: a byte-code stage cannot hand its
        codes to an fp32-schedule stage."""
        from . import _native as N_
        mods = list(deconv_layers)
        Nb, C, H, W = input_shape
        for i in range(0, len(mods), 3):
            up = 0 if i == 0 else 1
            if C % 4:      # no byte-code form (CoDeNet2x stage 0): that stage runs on the fp32 frozen schedule, NCHW input
                if up or not N_.lib().cdn_codenet_stage_fused_supported(Nb, C, H, W, 0, 0):
                    return False
            elif not N_.lib().cdn_codenet_stage_supported(Nb, C, H, W, 1, up):
                return False
            C, H, W = mods[i].quant_conv_channel_bn.conv.out_channels, 2 * H, 2 * W
        return True

    def _acts(self, st):
        return (st[0].quant_act[1], st[0].quant_identity_deform, st[1][1])

    def _alloc(self, shape, dev, nhwc_in, key):
        import ctypes
        from . import _native as N_
        Nb, C, H, W = shape
        lib = N_.lib()
        bufs, ws_bytes, ws32_bytes = [], 0, 0
        for i, st in enumerate(self.stages):
            cin = st[0].quant_deform_conv.in_channels
            cout = st[0].quant_conv_channel_bn.conv.out_channels
            up = 0 if i == 0 else 1
            Hs, Ws = (H, W) if i == 0 else (bufs[-1]["H"] * 2, bufs[-1]["W"] * 2)
            codes = cin % 4 == 0 and bool(lib.cdn_codenet_stage_supported(Nb, cin, Hs, Ws, 1 if (i or nhwc_in) else 0, up))
            if codes:
                ws_bytes = max(ws_bytes, lib.cdn_codenet_stage_frozen_workspace_bytes(Nb, cin, Hs, Ws, up))
            else:
                ws32_bytes = max(ws32_bytes, lib.cdn_codenet_stage_workspace_bytes(Nb, cin, Hs, Ws, up))
            bufs.append(dict(C=cin, Co=cout, H=Hs, W=Ws, up=up, codes=codes,
                             r8=torch.empty(Nb, Hs * Ws, cout, dtype=torch.int8, device=dev) if codes else None,
                             r=None if codes else torch.empty(Nb, Hs * Ws, cout, device=dev),
                             sums=None))
        sums_all = None
        if self.chain_scale:      # stage k's pointwise leaves the integer scale sums of stage k+1 (both on byte codes)
            take = [i for i in range(len(bufs) - 1)
                    if bufs[i]["codes"] and bufs[i + 1]["codes"]
                    and self.stages[i + 1][0].quant_conv_scale.int8_form() is not None]
            sizes = [(Nb * bufs[i]["H"] * bufs[i]["W"] + 3) // 4 * 4 for i in take]
            if take:      # ONE buffer, cleared by the step's first launch (cdn_quantact_frozen_params_clear)
                sums_all = torch.zeros(sum(sizes), dtype=torch.int32, device=dev)
                off = 0
                for i, sz in zip(take, sizes):
                    bufs[i]["sums"] = sums_all[off:off + Nb * bufs[i]["H"] * bufs[i]["W"]]
                    off += sz
        acts = [a for st in self.stages for a in self._acts(st)]
        n = len(acts)
        arr = ctypes.c_void_p * n
        last = bufs[-1]
        self._bufs = dict(
            key=key, stages=bufs, sums_all=sums_all,
            ws=torch.empty(ws_bytes + 512, dtype=torch.uint8, device=dev),
            ws32=torch.zeros(ws32_bytes // 4 + 64, device=dev) if ws32_bytes else None,
            overflow=OverflowFlags(len(bufs) + 16, dev),      # word i: stage i; words n .. n+15: the byte-code heads
            expanded=torch.empty(Nb, last["H"] * last["W"], last["Co"], device=dev),
            out=torch.empty(Nb, last["Co"], last["H"] * 2, last["W"] * 2, device=dev),
            n_acts=n, acts=acts,      # (kept alive: the pointer arrays below refer to their buffers)
            p_min=arr(*[a.x_min.data_ptr() for a in acts]), p_max=arr(*[a.x_max.data_ptr() for a in acts]),
            p_state=arr(*[a._device_state(dev).data_ptr() for a in acts]))
        for i, st in enumerate(self.stages):
            self._bufs["overflow"].name(i, self._acts(st))

    def head_flags(self):
        """the flag words the byte-code heads number from 0 (FusedHeads.forward_codes)"""
        return self._bufs["overflow"].slice(len(self.stages))

    def forward_codes(self, x, x_qstate=None, hw=None):
        """-> (r8 [N, H*W, Co] int8 codes of the last stage's output QuantAct (or the fp32 tensor when that stage
        had to run on the fp32 schedule), r_state pointer, the stage's shape dict).  Input as FusedHotPath.forward_nhwc."""
        from . import _native as N_
        nhwc_in = hw is not None
        codes_in = x.dtype == torch.int8       # byte codes of the QuantAct whose state is x_qstate (a frozen backbone)
        if not (x.is_cuda and (x.dtype == torch.float32 or (codes_in and nhwc_in and x_qstate is not None))
                and x.dim() == (3 if nhwc_in else 4)):
            raise NotImplementedError("FrozenHotPath needs a GPU tensor: float32 NCHW, float32 [N, H*W, C] with hw "
                                      "and its QuantAct state, or int8 codes [N, H*W, C] with hw and the state")
        x = x.contiguous()
        c0 = self.stages[0][0].quant_deform_conv.in_channels
        if codes_in and (c0 % 4 or x.shape[2] != c0):
            # CoDeNet2x (C = 2153, round 5): stage 0 runs on the fp32 frozen schedule (no byte-code form for C % 4 != 0), so
            # the backbone's codes -- rows padded to a multiple of 16 bytes -- are expanded to the values level / scale
            # first (re-quantising them with the same state returns the same codes)
            xf = self.__dict__.get("_xin")
            if xf is None or xf.shape != x.shape or xf.device != x.device:
                xf = self._xin = torch.empty(x.shape, dtype=torch.float32, device=x.device)
            N_.check(N_.lib().cdn_codenet_expand_codes(x.data_ptr(), x_qstate, xf.data_ptr(), x.numel(),
                                                        torch.cuda.current_stream(x.device).cuda_stream),
                     "cdn_codenet_expand_codes")
            # (an odd channel count has no channels-last form in the fused schedule either: NCHW, final values)
            x = xf.view(x.shape[0], hw[0], hw[1], x.shape[2])[..., :c0].permute(0, 3, 1, 2).contiguous()
            codes_in, nhwc_in, x_qstate, hw = False, False, None, None
        shape = (x.shape[0], x.shape[2], hw[0], hw[1]) if nhwc_in else tuple(x.shape)
        if shape[1] != c0:        # (the kernels take the channel count from the modules: a mismatch would read out of bounds)
            raise RuntimeError("FrozenHotPath: the input has %d channels, stage 0 expects %d" % (shape[1], c0))
        dev = x.device
        # the cached pointer arrays name the QuantActs' range buffers: a re-assigned buffer (load_state_dict(assign=
        # True), a .to() round trip) must rebuild them, so their addresses are part of the key
        key = (shape, dev, nhwc_in, codes_in) + tuple(p for st in self.stages for a in self._acts(st)
                                            for p in (a.x_min.data_ptr(), a.x_max.data_ptr()))
        if self._bufs is None or self._bufs["key"] != key:
            self._alloc(shape, dev, nhwc_in, key)
        B = self._bufs
        lib = N_.lib()
        stream = torch.cuda.current_stream(dev).cuda_stream
        Nb = shape[0]
        # ONE (bits, momentum, running_stat) over all nine QuantActs: cdn_quantact_frozen_params takes the bit
        # width once for every state it derives (stages with different activation_bit would silently get the
        # last stage's grid otherwise)
        bits, _, _ = uniform_act_settings(B["acts"], "FrozenHotPath (all stages)")
        # (scale, zero-point) of all nine frozen QuantActs from their range buffers: one launch per step
        sa = B["sums_all"]
        N_.check(lib.cdn_quantact_frozen_params_clear(B["n_acts"], B["p_min"], B["p_max"], B["p_state"], bits,
                                                      sa.data_ptr() if sa is not None else None,
                                                      sa.numel() * 4 if sa is not None else 0, stream),
                 "cdn_quantact_frozen_params_clear")
        ws_ptr = (B["ws"].data_ptr() + 255) // 256 * 256
        ws_bytes = B["ws"].numel() - (ws_ptr - B["ws"].data_ptr())
        ptr = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
        cur_ptr, cur_kind, cur_q = x.data_ptr(), (2 if codes_in else 1 if nhwc_in else 0), (x_qstate if nhwc_in else None)
        with torch.no_grad():
            for st, sb in zip(self.stages, B["stages"]):
                q = st[0]
                a_s, a_d, a_r = self._acts(st)
                sp = [a._device_state(dev).data_ptr() for a in (a_s, a_d, a_r)]
                if sb["codes"]:
                    (codes, scale, colsum), kb_flag = q.quant_conv_channel_bn.folded_int8(), 0
                else:
                    (codes, scale, colsum), kb_flag = stage_int8_codes(q.quant_conv_channel_bn)
                w_pw, b_pw = q.quant_conv_channel_bn.folded()
                w_sc = q.quant_conv_scale.quantized_weight().reshape(-1)
                w_dw = q.quant_deform_conv.quantized_weight()
                bound = q.quant_act[0]
                if sb["codes"]:
                    si = B["stages"].index(sb)
                    sums_in = B["stages"][si - 1]["sums"] if si > 0 and cur_kind == 2 else None
                    sw_ptr = nsc_ptr = None
                    if sums_in is not None:
                        sw_ptr = q.quant_conv_scale.int8_form()[1].data_ptr()
                    if sb["sums"] is not None:
                        nsc_ptr = self.stages[si + 1][0].quant_conv_scale.int8_form()[0].data_ptr()
                    rc = lib.cdn_codenet_stage_frozen_chained_forward(
                        cur_ptr, cur_kind | getattr(self, "gather_flag", 0), sb["up"], cur_q, Nb, sb["C"], sb["Co"],
                        sb["H"], sb["W"],
                        ptr(w_sc), ptr(q.quant_conv_scale.bias), float(bound.min_val), float(bound.max_val),
                        ptr(w_dw), ptr(codes), ptr(scale), ptr(colsum), ptr(b_pw), 1, sp[0], sp[1], sp[2],
                        ws_ptr, ws_bytes, sb["r8"].data_ptr(), B["overflow"].ptr(si),
                        ptr(sums_in), sw_ptr, nsc_ptr, ptr(sb["sums"]), stream)
                    N_.check(rc, "cdn_codenet_stage_frozen_chained_forward")
                    cur_ptr, cur_kind, cur_q = sb["r8"].data_ptr(), 2, sp[2]
                else:
                    if cur_kind == 2:
                        raise NotImplementedError("a byte-code stage cannot feed an fp32-schedule stage")
                    w32 = B["ws32"]
                    w32_ptr = (w32.data_ptr() + 255) // 256 * 256
                    w32_bytes = (w32.numel() * 4 - (w32_ptr - w32.data_ptr())) // 256 * 256
                    acts3 = []
                    for a in (a_s, a_d, a_r):
                        acts3 += [a.x_min.data_ptr(), a.x_max.data_ptr(), a._device_state(dev).data_ptr()]
                    rc = lib.cdn_codenet_stage_fused_forward(
                        cur_ptr, cur_kind | kb_flag, sb["up"], cur_q, Nb, sb["C"], sb["Co"], sb["H"], sb["W"],
                        ptr(w_sc), ptr(q.quant_conv_scale.bias), float(bound.min_val), float(bound.max_val),
                        ptr(w_dw), ptr(w_pw.reshape(w_pw.size(0), -1)), ptr(codes), ptr(scale), ptr(colsum), ptr(b_pw),
                        None, None, 1, *acts3, bits, float(a_r.momentum), 0, w32_ptr, w32_bytes,
                        sb["r"].data_ptr(), stream)
                    N_.check(rc, "cdn_codenet_stage_fused_forward")
                    cur_ptr, cur_kind, cur_q = sb["r"].data_ptr(), 1, sp[2]
        last = B["stages"][-1]
        return (last["r8"] if last["codes"] else last["r"]), cur_q, last

    def forward_nhwc(self, x, x_qstate=None, hw=None):
        """What FusedHotPath.forward_nhwc returns -- (fp32 [N, H*W, Co], QuantAct state pointer, shape) -- for the
        native heads: the byte codes expanded to the values level / scale (re-quantising them with the same state
        returns the same values)."""
        return self.expand(*self.forward_codes(x, x_qstate, hw))

    def expand(self, r, rq, last):
        """forward_codes' result -> what the fp32 heads take (byte codes expanded to level / scale; fp32 passes)."""
        from . import _native as N_
        if r.dtype != torch.int8:
            return r, rq, last
        B = self._bufs
        rc = N_.lib().cdn_codenet_expand_codes(r.data_ptr(), rq, B["expanded"].data_ptr(), r.numel(),
                                               torch.cuda.current_stream(r.device).cuda_stream)
        N_.check(rc, "cdn_codenet_expand_codes")
        return B["expanded"], rq, last

    def __call__(self, x):
        """The Sequential's output tensor (NCHW, up-sampled, fake-quantised), like FusedHotPath.__call__."""
        from . import _native as N_
        r, rq, last = self.forward_nhwc(x)
        B = self._bufs
        rc = N_.lib().cdn_codenet_unpack_nchw(r.data_ptr(), rq, B["out"].data_ptr(), x.shape[0], last["Co"], last["H"],
                                              last["W"], 1, torch.cuda.current_stream(x.device).cuda_stream)
        N_.check(rc, "cdn_codenet_unpack_nchw")
        return B["out"]

    def overflowed(self):
        """True when some code saturated since the last call of this method (synchronises; resets the flag): the
        batches computed in between must be recomputed with FusedHotPath (running_stat False)."""
        if self._bufs is None:
            return False
        return self._bufs["overflow"].any()

    def capture(self, x, codes_only=True, x_qstate=None, hw=None):
        """One pass over the static buffer `x` as a HIP graph; returns replay() -> the static output (byte codes
        of the last stage with codes_only, else the unpacked NCHW tensor).  x_qstate / hw: a channels-last input
        (fp32 pre-quantisation values or int8 codes) as in forward_codes."""
        if hw is not None:
            if not codes_only:
                raise NotImplementedError("a channels-last input is captured with codes_only")
            run = lambda t: self.forward_codes(t, x_qstate, hw)[0]      # noqa: E731
        else:
            run = (lambda t: self.forward_codes(t)[0]) if codes_only else self.__call__
        run(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = run(x)
        self._graph = g

        def replay():
            g.replay()
            return out
        return replay


class FusedHeads:
    """The detection heads on the stage kernels, fed by
    ``FusedHotPath.forward_nhwc``: per head
        1x1 conv (+BN) -> ReLU [-> QuantAct] -> depthwise 3x3 (+BN) -> ReLU [-> QuantAct] -> 1x1 conv + bias
    (fp32: shufflenetv2_dcn.py:244-271 nn.Sequential; W4A8: QuantDepthwiseNode, quant_modules.py:1013-1071).
    The first 1x1 conv runs at HALF resolution (a 1x1 conv commutes with the nearest up-sampling and the
    QuantAct extremes of a replicated tensor are those of the original), the depthwise kernel up-samples
    by addressing, so the up-sampled 64-channel tensor is never built.  Same parameters and QuantAct
    buffers (updated in place) as calling the head modules on the unpacked tensor."""

    def __init__(self, heads, int8_pointwise=True, small_tail=True, streams=True):
        self.heads = dict(heads)
        self.int8_pointwise = int8_pointwise
        # W4A8 heads with <= 4 output channels (wh, reg): range pass + depthwise -> quantise -> 1x1 conv as exact
        # integer dot products on the VALU (cdn_codenet_head_range_forward / _head_tail_small_forward)
        self.small_tail = small_tail and int8_pointwise
        self.streams = streams
        self._bufs = None
        self._affine = {}

    def _bn_affine(self, bn):
        return bn_affine(self._affine, bn)

    @staticmethod
    def supported(heads):
        """True when every head is a form the fused schedule implements with the QuantAct settings it
        implements (see act_fusable)."""
        from .portable_quantizer.quant_modules import QuantDepthwiseNode
        for mod in dict(heads).values():
            if isinstance(mod, QuantDepthwiseNode):
                if not (act_fusable(mod.quant_act1[1]) and act_fusable(mod.quant_act3[1])):
                    return False
            elif isinstance(mod, nn.Conv2d):
                if tuple(mod.kernel_size) != (1, 1):
                    return False
            elif not (isinstance(mod, nn.Sequential) and len(mod) == 7):
                return False
        return True

    def _params(self, mod):
        """-> list of layer dicts in execution order."""
        from .portable_quantizer.quant_modules import QuantDepthwiseNode
        if isinstance(mod, QuantDepthwiseNode):
            w1, b1 = mod.quant_convbn1.folded()
            w2, b2 = mod.quant_convbn2.folded()
            i8 = self.int8_pointwise
            return [
                dict(kind="pw", w=w1.reshape(w1.size(0), -1), bias=b1, ep=None, relu=1,
                     i8=mod.quant_convbn1.folded_int8() if i8 else None, act=mod.quant_act1[1]),
                dict(kind="dw", w=w2.reshape(w2.size(0), 9), bias=b2, ep=None, relu=1,
                     act=mod.quant_act3[1]),
                dict(kind="pw", w=mod.quant_conv.quantized_weight().reshape(mod.quant_conv.out_channels, -1),
                     bias=mod.quant_conv.bias, ep=None, relu=0,
                     i8=mod.quant_conv.int8_form() if i8 else None, act=None)]
        if isinstance(mod, nn.Conv2d):
            return [dict(kind="pw", w=mod.weight.reshape(mod.out_channels, -1), bias=mod.bias, ep=None,
                         relu=0, i8=None, act=None)]
        conv1, bn1, _, conv2, bn2, _, conv3 = list(mod)
        return [
            dict(kind="pw", w=conv1.weight.reshape(conv1.out_channels, -1), bias=conv1.bias,
                 ep=self._bn_affine(bn1), relu=1, i8=None, act=None),
            dict(kind="dw", w=conv2.weight.reshape(conv2.out_channels, 9), bias=conv2.bias,
                 ep=self._bn_affine(bn2), relu=1, act=None),
            dict(kind="pw", w=conv3.weight.reshape(conv3.out_channels, -1), bias=conv3.bias, ep=None,
                 relu=0, i8=None, act=None)]

    def _alloc(self, r, shape):
        from . import _native as N_
        Nb = r.shape[0]
        dev = r.device
        C, Hs, Ws = shape["Co"], shape["H"], shape["W"]
        M = Nb * Hs * Ws
        aux = N_.lib().cdn_codenet_aux_workspace_bytes()
        self._bufs = dict(
            key=(tuple(r.shape), dev), y1=torch.empty(M, C, device=dev),
            y2={}, o={},                                      # unfused tail only: allocated on first use, PER HEAD
            ws=torch.zeros(aux // 4 + 64, device=dev),        # arrival counters start at zero
            out={name: torch.empty(Nb, self._out_channels(m), 2 * Hs, 2 * Ws, device=dev)
                 for name, m in self.heads.items()})

    @staticmethod
    def _out_channels(mod):
        from .portable_quantizer.quant_modules import QuantDepthwiseNode
        if isinstance(mod, QuantDepthwiseNode):
            return mod.quant_conv.out_channels
        if isinstance(mod, nn.Conv2d):
            return mod.out_channels
        return list(mod)[-1].out_channels

    def __call__(self, r, r_qstate, shape):
        from . import _native as N_
        from . import ops
        if self._bufs is None or self._bufs["key"] != (tuple(r.shape), r.device):
            self._alloc(r, shape)
        B = self._bufs
        lib = N_.lib()
        Nb = r.shape[0]
        C, Hs, Ws = shape["Co"], shape["H"], shape["W"]
        M = Nb * Hs * Ws
        main = torch.cuda.current_stream(r.device)
        stream = main.cuda_stream
        ws_ptr = (B["ws"].data_ptr() + 255) // 256 * 256
        ws_bytes = (B["ws"].numel() * 4 - (ws_ptr - B["ws"].data_ptr())) // 256 * 256
        main_launch = (stream, ws_ptr, ws_bytes)
        # the heads are independent chains (1x1 -> range pass -> tail) of kernels that do not fill the chip
        # on their own: head i > 0 runs on its own stream with its own arrival counters and y1 buffer
        use_streams = self.streams and len(self.heads) > 1
        if use_streams and (B.get("side") is None or len(B["side"]) < len(self.heads) - 1):
            aux = N_.lib().cdn_codenet_aux_workspace_bytes()
            B["side"] = [torch.cuda.Stream(r.device) for _ in range(len(self.heads) - 1)]
            B["ws_side"] = [torch.zeros(aux // 4 + 64, device=r.device) for _ in range(len(self.heads) - 1)]
            B["y1_side"] = [torch.empty_like(B["y1"]) for _ in range(len(self.heads) - 1)]
        forked = []
        ptr = lambda t: t.data_ptr() if t is not None else None   # noqa: E731

        def act_args(act):
            if act is None:
                return [None, None, None, 8, 0.99, 0]
            if not act_fusable(act):
                raise NotImplementedError("FusedHeads: this QuantAct configuration (percentile / symmetric / "
                                          "full precision) is not implemented by the fused schedule")
            return [act.x_min.data_ptr(), act.x_max.data_ptr(), act._device_state(r.device).data_ptr(),
                    act.activation_bit, act.momentum, int(act.running_stat)]

        def pw(a, aq, m, layer, out):
            i8 = layer["i8"] if layer["i8"] is not None else (None, None, None)
            ep = layer["ep"] or (None, None)
            rec = ops._tic("head_pw", (layer["w"].shape[1], layer["w"].shape[0], m))
            rc = lib.cdn_codenet_pointwise_nhwc_forward(
                a.data_ptr(), aq, m, layer["w"].shape[1], layer["w"].shape[0], 0, 0, ptr(layer["w"]),
                ptr(i8[0]), ptr(i8[1]), ptr(i8[2]), ptr(layer["bias"]), ptr(ep[0]), ptr(ep[1]),
                layer["relu"], *act_args(layer["act"]), ws_ptr, ws_bytes, out.data_ptr(), stream)
            ops._toc(rec)
            N_.check(rc, "cdn_codenet_pointwise_nhwc_forward")

        outs = {}
        # every head's derived parameters (folded / fake-quantised weights, int8 codes: torch ops on the MAIN stream when
        # they are not cached yet) BEFORE the side streams fork: derived inside the loop they raced with the side-stream
        # kernels that read them -- the first call of a fresh FusedHeads returned garbage for a head about once in a
        # thousand runs (tools/stress_heads.py; both failures seen were first calls)
        with torch.no_grad():                    # (derived tensors are cached per weight version only without autograd)
            params = {name: self._params(mod) for name, mod in self.heads.items()}
        for layers_ in params.values():          # (and the QuantActs' device states: created by a fill on the main stream)
            for l_ in layers_:
                if l_["act"] is not None:
                    l_["act"]._device_state(r.device)
        if use_streams:
            # fork EVERY side stream before head 0 puts its kernels on the main stream: forked inside the loop, a side
            # stream waited for everything the main stream held by then -- head 0's whole chain -- and the heads ran as
            # "head 0, then the others" (seen in the kernel trace of round 4: 121 us of head 0 alone on the GPU)
            for sd in B["side"][:len(self.heads) - 1]:
                sd.wait_stream(main)
        with torch.no_grad():
            for hi, (name, mod) in enumerate(self.heads.items()):
                y1buf = B["y1"]
                if use_streams and hi > 0:
                    sd, wsb, y1buf = B["side"][hi - 1], B["ws_side"][hi - 1], B["y1_side"][hi - 1]
                    forked.append(sd)
                    stream = sd.cuda_stream
                    ws_ptr = (wsb.data_ptr() + 255) // 256 * 256
                    ws_bytes = (wsb.numel() * 4 - (ws_ptr - wsb.data_ptr())) // 256 * 256
                else:
                    stream, ws_ptr, ws_bytes = main_launch
                layers = params[name]
                small = (self.small_tail and len(layers) == 3 and layers[0]["act"] is not None
                         and layers[1]["act"] is not None and layers[1]["ep"] is None and layers[1]["relu"]
                         and layers[2]["i8"] is not None and C == 64
                         and (layers[2]["w"].shape[0] <= 4 or (layers[2]["w"].shape[0] <= 32 and Ws % 16 == 0))
                         and layers[2]["act"] is None and not layers[2]["relu"])
                if not small and name not in B["o"]:
                    B["o"][name] = torch.empty(4 * M, self._out_channels(mod), device=r.device)
                if not small and len(layers) == 3 and name not in B["y2"]:
                    # one scratch per head: the heads run concurrently on their own streams (a shared one was a race
                    # between them -- fp32 heads only: the W4A8 tails never store this tensor)
                    B["y2"][name] = torch.empty(4 * M, C, device=r.device)
                if len(layers) == 1:          # head_conv == 0: one 1x1 conv, up-sampled afterwards
                    o = B["o"][name][:M]
                    pw(r, r_qstate, M, layers[0], o)
                    rc = lib.cdn_codenet_unpack_nchw(o.data_ptr(), None, B["out"][name].data_ptr(), Nb,
                                                     o.shape[1], Hs, Ws, 1, stream)
                    N_.check(rc, "cdn_codenet_unpack_nchw")
                    outs[name] = B["out"][name]
                    continue
                l1, l2, l3 = layers
                pw(r, r_qstate, M, l1, y1buf)
                q1 = l1["act"]._device_state(r.device).data_ptr() if l1["act"] is not None else None
                ep = l2["ep"] or (None, None)
                if small:
                    # W4A8 heads: streaming range pass, then depthwise -> quantise -> 1x1 conv (<= 4 outputs: exact
                    # integer dot products on the VALU; up to 32: int8 matrix cores) -> NCHW; the 64-channel
                    # full-resolution tensor is never stored (bit-identical to the unfused schedule)
                    rec = ops._tic("head_range", (C, 2 * Hs, 2 * Ws))
                    rc = lib.cdn_codenet_head_range_forward(
                        y1buf.data_ptr(), q1, Nb, C, Hs, Ws, ptr(l2["w"]), ptr(l2["bias"]),
                        *act_args(l2["act"]), ws_ptr, ws_bytes, stream)
                    ops._toc(rec)
                    N_.check(rc, "cdn_codenet_head_range_forward")
                    q2 = l2["act"]._device_state(r.device).data_ptr()
                    i8 = l3["i8"]
                    rec = ops._tic("head_tail_small", (C, l3["w"].shape[0], 4 * M))
                    rc = lib.cdn_codenet_head_tail_small_forward(
                        y1buf.data_ptr(), q1, Nb, C, Hs, Ws, ptr(l2["w"]), ptr(l2["bias"]), q2, ptr(i8[0]),
                        ptr(i8[1]), ptr(i8[2]), ptr(l3["bias"]), l3["w"].shape[0], B["out"][name].data_ptr(),
                        stream)
                    ops._toc(rec)
                    N_.check(rc, "cdn_codenet_head_tail_small_forward")
                    outs[name] = B["out"][name]
                    continue
                rec = ops._tic("head_dw", (C, 2 * Hs, 2 * Ws))
                rc = lib.cdn_codenet_dw3x3_nhwc_forward(
                    y1buf.data_ptr(), q1, Nb, C, Hs, Ws, 1, 1, 0, 0, ptr(l2["w"]), ptr(l2["bias"]),
                    ptr(ep[0]), ptr(ep[1]), l2["relu"], *act_args(l2["act"]), ws_ptr, ws_bytes,
                    B["y2"][name].data_ptr(), stream)
                ops._toc(rec)
                N_.check(rc, "cdn_codenet_dw3x3_nhwc_forward")
                q2 = l2["act"]._device_state(r.device).data_ptr() if l2["act"] is not None else None
                o = B["o"][name]
                pw(B["y2"][name], q2, 4 * M, l3, o)
                rc = lib.cdn_codenet_unpack_nchw(o.data_ptr(), None, B["out"][name].data_ptr(), Nb,
                                                 o.shape[1], 2 * Hs, 2 * Ws, 0, stream)
                N_.check(rc, "cdn_codenet_unpack_nchw")
                outs[name] = B["out"][name]
        for sd in forked:
            main.wait_stream(sd)
        return outs


    # ---- frozen serving mode: the heads on the stages' byte codes --------------------------------------------
    def codes_supported(self, shape):
        """True when every head is a W4A8 QuantDepthwiseNode (64 channels, <= 32 outputs) with both QuantActs frozen:
        the form ``forward_codes`` implements."""
        from .portable_quantizer.quant_modules import QuantDepthwiseNode
        if shape["Co"] != 64 or not self.small_tail:
            return False
        for mod in self.heads.values():
            if not isinstance(mod, QuantDepthwiseNode):
                return False
            a1, a3 = mod.quant_act1[1], mod.quant_act3[1]
            if not (act_fusable(a1) and act_fusable(a3)) or a1.running_stat or a3.running_stat:
                return False
            nc = mod.quant_conv.out_channels
            if not (nc <= 4 or (nc <= 32 and shape["W"] % 16 == 0)) or mod.quant_conv.int8_form() is None:
                return False
        return True

    def forward_codes(self, r8, r_qstate, shape, overflow):
        """The heads on the BYTE CODES of the last deform stage (``FrozenHotPath.forward_codes``), every QuantAct
        frozen: per head the int8 pointwise kernel on codes (cdn_codenet_pointwise_q8_forward: exact integer sums,
        the codes of quant_act1 written as bytes) and the row-streaming tail reading those bytes
        (cdn_codenet_head_tail_small_q8_forward); no range passes, no fp32 copy of the stage output or of y1.
        Same values as ``__call__`` on the expanded codes with the same frozen states (the first 1x1 conv is the same
        integer sum; the tail decodes a code to the value its fp32 form fake-quantises to).  `overflow`: an OverflowFlags
        (word 2i: head i's y1 codes, word 2i + 1: its tail) or an int32 tensor (one word for everything)."""
        import ctypes
        from . import _native as N_
        dev = r8.device
        if self._bufs is None or self._bufs["key"] != (("codes",) + tuple(r8.shape), dev):
            Nb = r8.shape[0]
            M = Nb * shape["H"] * shape["W"]
            acts = [a for m in self.heads.values() for a in (m.quant_act1[1], m.quant_act3[1])]
            arr = ctypes.c_void_p * len(acts)
            self._bufs = dict(
                key=(("codes",) + tuple(r8.shape), dev), acts=acts,
                y8=[torch.empty(M, 64, dtype=torch.int8, device=dev) for _ in self.heads],
                side=[torch.cuda.Stream(dev) for _ in range(len(self.heads) - 1)] if self.streams else [],
                out={name: torch.empty(Nb, self._out_channels(m), 2 * shape["H"], 2 * shape["W"], device=dev)
                     for name, m in self.heads.items()})
        B = self._bufs
        acts = B["acts"]
        ptrs = tuple(a.x_min.data_ptr() for a in acts)
        if B.get("ptrs") != ptrs:             # (the arrays name the range buffers: rebuilt when they move)
            arr = ctypes.c_void_p * len(acts)
            B["p"] = (arr(*[a.x_min.data_ptr() for a in acts]), arr(*[a.x_max.data_ptr() for a in acts]),
                      arr(*[a._device_state(dev).data_ptr() for a in acts]))
            B["ptrs"] = ptrs
        lib = N_.lib()
        Nb, Hs, Ws = r8.shape[0], shape["H"], shape["W"]
        M = Nb * Hs * Ws
        main = torch.cuda.current_stream(dev)
        bits, _, _ = uniform_act_settings(acts, "FusedHeads.forward_codes")
        N_.check(lib.cdn_quantact_frozen_params(len(acts), *B["p"], bits, main.cuda_stream),
                 "cdn_quantact_frozen_params")
        ptr = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
        forked = []
        with torch.no_grad():
            params = {name: self._params(mod) for name, mod in self.heads.items()}   # (derived on the main stream: before the fork)
        for layers_ in params.values():
            for l_ in layers_:
                if l_["act"] is not None:
                    l_["act"]._device_state(dev)
        for sd in B["side"][:max(0, len(self.heads) - 1)]:      # (fork before head 0's kernels are on the main stream)
            sd.wait_stream(main)
        with torch.no_grad():
            for hi, (name, mod) in enumerate(self.heads.items()):
                st = main
                if B["side"] and hi > 0:
                    st = B["side"][hi - 1]
                    forked.append(st)
                l1, l2, l3 = params[name]
                q1 = l1["act"]._device_state(dev).data_ptr()
                q2 = l2["act"]._device_state(dev).data_ptr()
                c1, s1, k1 = l1["i8"]
                y8 = B["y8"][hi]
                if isinstance(overflow, OverflowFlags) and 2 * hi + 1 < overflow.count():
                    of1, of2 = overflow.ptr(2 * hi), overflow.ptr(2 * hi + 1)
                    overflow.name(2 * hi, [l1["act"]])
                    overflow.name(2 * hi + 1, [l2["act"]])
                else:
                    of1 = of2 = overflow.data_ptr()
                rc = lib.cdn_codenet_pointwise_q8_forward(
                    r8.data_ptr(), r_qstate, M, 64, 64, c1.data_ptr(), s1.data_ptr(), k1.data_ptr(), ptr(l1["bias"]),
                    1, q1, y8.data_ptr(), None, of1, st.cuda_stream)
                N_.check(rc, "cdn_codenet_pointwise_q8_forward")
                i8 = l3["i8"]
                rc = lib.cdn_codenet_head_tail_small_q8_forward(
                    y8.data_ptr(), q1, Nb, 64, Hs, Ws, ptr(l2["w"]), ptr(l2["bias"]), q2, ptr(i8[0]), ptr(i8[1]),
                    ptr(i8[2]), ptr(l3["bias"]), l3["w"].shape[0], B["out"][name].data_ptr(), of2,
                    st.cuda_stream)
                N_.check(rc, "cdn_codenet_head_tail_small_q8_forward")
        for sd in forked:
            main.wait_stream(sd)
        return B["out"]


class FusedBackbone:
    """layer0 .. layer4 of a ``PoseShuffleNetV2`` on the HIP kernels.
    W4A8: the reference's module tree after ``quantize_shufflenetv2_dcn`` (quantize_model.py:26-60) --
    layer0 = QuantBnConv2d(8) + ReLU + QuantAct, layers 1-3 = QuantBaseNode units sharing one block-output
    QuantAct per layer (quant_modules.py:809-907), layer4 = QuantBnConv2d + ReLU + QuantAct -- with the
    same parameters and QuantAct buffers (updated in place, in the reference's order).
    fp32: the un-quantised tree (shufflenetv2_dcn.py:57-114,205-240) in eval mode, BatchNorm folded into
    the convolutions, same kernels without quantisers.

    Activations are channels-last fp32.  A unit's three convolutions run on ONE half of the channels
    through row-strided views (no split copy); intermediates hold pre-quantisation values and are
    fake-quantised by their consumer while loading; concat + channel_shuffle is one interleave kernel
    that also applies the shared block-output QuantAct, so a unit's output tensor holds final values.
    Returns what ``FusedHotPath.forward_nhwc(x, x_qstate, hw)`` takes."""

    def __init__(self, model, int8_pointwise=True, shuffle_free=True, two_streams=True):
        self.model = model
        self.int8 = int8_pointwise
        self.shuffle_free = shuffle_free and int8_pointwise
        self.two_streams = two_streams
        self._bufs = None

    def _l4_weights(self, q4, logical, dev, gens=None):
        """layer4's 1x1 weights with the input columns in the physical order of the last layer."""
        key = (q4.conv.weight.data_ptr(), q4.conv.weight._version, q4.bn.weight._version,
               q4.bn.running_var._version, tuple(logical), tuple(gens) if gens is not None else None, dev)
        c = self.__dict__.get("_l4_cache")
        if c is None or c[0] != key:
            w, b = q4.folded()
            codes, scale, colsum = q4.folded_int8()
            cols = torch.tensor(logical, device=dev, dtype=torch.long)
            K, Co = len(logical), w.shape[0]
            cp = torch.zeros(Co, (K + 63) // 64 * 64, dtype=torch.int8, device=dev)
            cp[:, :K] = codes[:, cols]
            W4 = dict(w=w.reshape(Co, -1)[:, cols].contiguous(), codes=cp, scale=scale, colsum=colsum,
                      bias=b.contiguous(), Co=Co, K=K)
            self._l4_cache = (key, W4)
        return self._l4_cache[1]

    @staticmethod
    def supported(model):
        from .portable_quantizer.quant_modules import QuantAct, QuantBaseNode, QuantBnConv2d
        try:
            l0, l4 = model.layer0, model.layer4
            if isinstance(l0[0], nn.Conv2d):          # fp32 model: conv, bn, relu stem without max-pool
                c = l0[0]
                ok = ((len(l0) == 3 or (len(l0) == 4 and FusedBackbone._is_pool(l0[3])))
                      and c.out_channels == 24 and c.in_channels == 3 and c.bias is None
                      and tuple(c.kernel_size) == (3, 3) and tuple(c.padding) == (1, 1)
                      and isinstance(l4[0], nn.Conv2d))
                for name in ("layer1", "layer2", "layer3"):
                    for node in getattr(model, name):
                        ok = ok and hasattr(node, "b2") and len(node.b2) == 8
                return bool(ok)
            ok = (isinstance(l0[0], QuantBnConv2d) and len(l0[1]) in (2, 3) and isinstance(l0[1][1], QuantAct)
                  and (len(l0[1]) == 2 or FusedBackbone._is_pool(l0[1][2]))
                  and l0[0].conv.out_channels == 24 and l0[0].conv.in_channels == 3
                  and tuple(l0[0].conv.kernel_size) == (3, 3) and tuple(l0[0].conv.padding) == (1, 1)
                  and isinstance(l4[0], QuantBnConv2d) and isinstance(l4[1][1], QuantAct))
            for name in ("layer1", "layer2", "layer3"):
                for node in getattr(model, name):
                    ok = ok and isinstance(node, QuantBaseNode) and node.quant_act.quant_mode == "asymmetric" \
                        and node.quant_act2.quant_mode == "asymmetric"
            if ok:      # plain min/max, asymmetric, quantising QuantActs only (--act-percentile: module path)
                for part in (l0, model.layer1, model.layer2, model.layer3, l4):
                    ok = ok and all(act_fusable(a) for a in part.modules() if isinstance(a, QuantAct))
            return bool(ok)
        except (AttributeError, IndexError, TypeError):
            return False

    @staticmethod
    def _is_pool(m):
        def two(v):
            return (v, v) if isinstance(v, int) else tuple(v)
        return (isinstance(m, nn.MaxPool2d) and two(m.kernel_size) == (3, 3) and two(m.stride) == (2, 2)
                and two(m.padding) == (1, 1) and two(m.dilation) == (1, 1) and not m.ceil_mode)

    # -- low-level launches ----------------------------------------------------------------------
    def _act_args(self, act, dev):
        if act is None:
            return [None, None, None, 8, 0.99, 0]
        return [act.x_min.data_ptr(), act.x_max.data_ptr(), act._device_state(dev).data_ptr(),
                act.activation_bit, act.momentum, int(act.running_stat)]

    def _folded(self, convbn):
        """(weight, bias) of a QuantBnConv2d (fake-quantised, BN folded) or of an fp32 (conv, bn) pair with
        the BatchNorm folded into the convolution (eval mode; derived once)."""
        if not isinstance(convbn, tuple):
            return convbn.folded()
        conv, bn = convbn
        cache = self.__dict__.setdefault("_fp32_fold", {})
        key = (id(conv), conv.weight._version, bn.weight._version, bn.running_var._version)
        if key not in cache:
            with torch.no_grad():
                sf = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                w = (conv.weight * sf.reshape(-1, 1, 1, 1)).contiguous()
                b0 = conv.bias if conv.bias is not None else torch.zeros_like(bn.running_mean)
                cache[key] = (w, ((b0 - bn.running_mean) * sf + bn.bias).contiguous())
        return cache[key]

    def _pw(self, a_ptr, a_q, M, lda, convbn, relu, act, out, ldo):
        from . import _native as N_
        w, b = self._folded(convbn)
        Co, C = w.shape[0], w.shape[1]
        # 4-bit codes: integer MFMA when the input carries a QuantAct state, exact bf16 split otherwise
        i8 = convbn.folded_int8() if (self.int8 and not isinstance(convbn, tuple)) else None
        i8 = i8 if i8 is not None else (None, None, None)
        ptr = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
        w2 = w.reshape(Co, C)
        rc = N_.lib().cdn_codenet_pointwise_nhwc_forward(
            a_ptr, a_q, M, C, Co, lda, ldo, w2.data_ptr(), ptr(i8[0]), ptr(i8[1]), ptr(i8[2]), ptr(b),
            None, None, int(relu), *self._act_args(act, out.device), self._ws_ptr, self._ws_bytes,
            out.data_ptr(), self._stream)
        N_.check(rc, "cdn_codenet_pointwise_nhwc_forward")

    def _dw(self, a, a_q, N, C, H, W, stride, ld_in, convbn, act, out, ld_out):
        from . import _native as N_
        w, b = self._folded(convbn)
        rc = N_.lib().cdn_codenet_dw3x3_nhwc_forward(
            a.data_ptr(), a_q, N, C, H, W, 0, stride, ld_in, ld_out, w.reshape(C, 9).data_ptr(), b.data_ptr(),
            None, None, 0, *self._act_args(act, out.device), self._ws_ptr, self._ws_bytes, out.data_ptr(),
            self._stream)
        N_.check(rc, "cdn_codenet_dw3x3_nhwc_forward")

    def _il(self, srcA, ldA, qA, srcB, ldB, qB, M, h, dst, ld_dst):
        from . import _native as N_
        rc = N_.lib().cdn_codenet_interleave_forward(srcA, ldA, qA, srcB, ldB, qB, M, h, dst.data_ptr(),
                                                     ld_dst, self._stream)
        N_.check(rc, "cdn_codenet_interleave_forward")

    # -- buffers -----------------------------------------------------------------------------------
    @staticmethod
    def _unit(node):
        """Layers of a ShuffleNetV2 unit: W4A8 QuantBaseNode (QuantBnConv2d + QuantAct objects) or the fp32
        BaseNode (shufflenetv2_dcn.py:57-114: b2 = pw, bn, relu, dw, bn, pw, bn, relu; b1 = dw, bn, pw, bn,
        relu) as (conv, bn) pairs with no quantisers."""
        if hasattr(node, "quant_convbn1"):
            u = dict(c1=node.quant_convbn1, a1=node.quant_act1, c2=node.quant_convbn2, a2=node.quant_act2,
                     c3=node.quant_convbn3, sh=node.quant_act, h=node.quant_convbn3.conv.out_channels,
                     cin=node.quant_convbn1.conv.in_channels)
            if node.stride == 2:
                u.update(c4=node.quant_convbn4, a4=node.quant_act4, c5=node.quant_convbn5)
            return u
        b2 = node.b2
        u = dict(c1=(b2[0], b2[1]), a1=None, c2=(b2[3], b2[4]), a2=None, c3=(b2[5], b2[6]), sh=None,
                 h=b2[5].out_channels, cin=b2[0].in_channels)
        if node.stride == 2:
            b1 = node.b1
            u.update(c4=(b1[0], b1[1]), a4=None, c5=(b1[2], b1[3]))
        return u

    def _layer_bufs(self, nodes, h, cin, Nb, H, W, dev):
        """Scratch for one layer (a stride-2 unit followed by stride-1 units), cached per shape."""
        key = (id(nodes[0]), Nb, H, W, dev)
        cache = self.__dict__.setdefault("_layer_cache", {})
        if key not in cache:
            pad4 = lambda c: (c + 3) // 4 * 4   # noqa: E731
            oup = 2 * h
            s = nodes[0].stride
            Ho, Wo = ((H - 1) // 2 + 1, (W - 1) // 2 + 1) if s == 2 else (H, W)
            Mi, Mo = Nb * H * W, Nb * Ho * Wo
            z = lambda m, c: torch.zeros(m, c, device=dev)   # noqa: E731  (padding channels stay finite)
            cache[key] = dict(
                cin=cin, C=oup, h=h, ldh=pad4(h), Hin=H, Win=W, H=Ho, W=Wo,
                t4=z(Mo, pad4(cin)) if s == 2 else None, t5=z(Mo, pad4(h)) if s == 2 else None,
                t1s2=None,       # pw1 of the stride-2 unit at input resolution: allocated on demand (_t1s2; the
                                 # layer-1 unit recomputes it inside its depthwise instead, cdn_codenet_pwdw_s2_forward)
                Mi=Mi, dev=dev,
                t1=z(Mo, pad4(h)), t2=z(Mo, pad4(h)), t3=z(Mo, pad4(h)), ya=z(Mo, oup), yb=z(Mo, oup))
        return cache[key]

    @staticmethod
    def _t1s2(L):
        if L["t1s2"] is None:
            L["t1s2"] = torch.zeros(L["Mi"], L["ldh"], device=L["dev"])
        return L["t1s2"]

    def _prepare(self, dev):
        from . import _native as N_
        if self.__dict__.get("_ws") is None or self._ws.device != dev:
            aux = N_.lib().cdn_codenet_aux_workspace_bytes()
            self._ws = torch.zeros(aux // 4 + 64, device=dev)          # arrival counters start at zero
        self._ws_ptr = (self._ws.data_ptr() + 255) // 256 * 256
        self._ws_bytes = (self._ws.numel() * 4 - (self._ws_ptr - self._ws.data_ptr())) // 256 * 256
        self._stream = torch.cuda.current_stream(dev).cuda_stream
        self._dev = dev

    def run_units(self, nodes, x, x_ld, x_q, Nb, H, W):
        """A chain of QuantBaseNode units (first one may be stride 2) sharing their block-output QuantAct.
        x: channels-last [Nb*H*W, x_ld] holding PRE-quantisation values with QuantAct state pointer x_q,
        or final values (x_q None).  Returns (y [Nb*Ho*Wo, C] final values, C, Ho, Wo)."""
        dev = x.device
        self._prepare(dev)
        units = [self._unit(n) for n in nodes]
        for u_ in units:          # the QuantActs' device states exist before branch 1 forks to the side stream (a state is
            for k_ in ("a1", "a2", "a4", "sh"):     # created by a fill on the CURRENT stream: see FusedHeads.forward)
                if u_.get(k_) is not None:
                    u_[k_]._device_state(dev)
        cin = units[0]["cin"] if nodes[0].stride == 2 else x_ld
        L = self._layer_bufs(nodes, units[0]["h"], cin, Nb, H, W, dev)
        h, ldh, C = L["h"], L["ldh"], L["C"]
        Mi, Mo = Nb * L["Hin"] * L["Win"], Nb * L["H"] * L["W"]
        qptr = lambda act: act._device_state(dev).data_ptr() if act is not None else None   # noqa: E731
        y, y_other = L["ya"], L["yb"]
        with torch.no_grad():
            for node, u in zip(nodes, units):
                sh = u["sh"]                              # the layer's shared block-output QuantAct (W4A8)
                if node.stride == 2:
                    # branch 1 (reference order: first): dw s2 -> QuantAct -> pw -> ReLU -> shared QuantAct
                    self._dw(x, x_q, Nb, cin, L["Hin"], L["Win"], 2, x_ld, u["c4"], u["a4"], L["t4"],
                             L["t4"].shape[1])
                    self._pw(L["t4"].data_ptr(), qptr(u["a4"]), Mo, L["t4"].shape[1], u["c5"], True, sh,
                             L["t5"], ldh)
                    self._il(L["t5"].data_ptr(), ldh, qptr(sh), None, 0, None, Mo, h, y, C)
                    # branch 2: pw -> ReLU -> QuantAct -> dw s2 -> QuantAct -> pw -> ReLU -> shared QuantAct
                    self._pw(x.data_ptr(), x_q, Mi, x_ld, u["c1"], True, u["a1"], self._t1s2(L), ldh)
                    self._dw(L["t1s2"], qptr(u["a1"]), Nb, h, L["Hin"], L["Win"], 2, ldh, u["c2"], u["a2"],
                             L["t2"], ldh)
                    self._pw(L["t2"].data_ptr(), qptr(u["a2"]), Mo, ldh, u["c3"], True, sh, L["t3"], ldh)
                    self._il(None, 0, None, L["t3"].data_ptr(), ldh, qptr(sh), Mo, h, y, C)
                else:
                    # x holds FINAL values; x1 = x[:, :h] passes through, x2 = x[:, h:] is a strided view
                    self._pw(x.data_ptr() + 4 * h, None, Mo, C, u["c1"], True, u["a1"], L["t1"], ldh)
                    self._dw(L["t1"], qptr(u["a1"]), Nb, h, L["H"], L["W"], 1, ldh, u["c2"], u["a2"], L["t2"],
                             ldh)
                    self._pw(L["t2"].data_ptr(), qptr(u["a2"]), Mo, ldh, u["c3"], True, sh, L["t3"], ldh)
                    self._il(x.data_ptr(), C, None, L["t3"].data_ptr(), ldh, qptr(sh), Mo, h, y, C)
                x, x_ld, x_q = y, C, None
                y, y_other = y_other, y
        return x, C, L["H"], L["W"]

    # -- layers without a physical channel shuffle -----------------------------------------------------
    # concat + channel_shuffle(2) only renames channels, so a layer keeps ONE activation tensor whose
    # physical channel slots never move: the pass-through half of a stride-1 unit stays where it is, the
    # unit's branch writes its (pre-quantisation) output into the slots its input half occupied, and the
    # renaming is folded into the weights on the host (columns permuted to the physical order, zero columns
    # for the pass-through half; `out_map` names the slot of each output channel).  The layer's shared
    # block-output QuantAct moves at every call, so each call writes its state to its own slot of a state
    # array and slot p remembers which call ("generation") produced it: consumers fake-quantise channel p
    # with state gen[p] while loading -- the values every reader sees are the ones the reference
    # materialised at that call.  No interleave kernel, no copy of the pass-through half.
    _MIXED_MAX_C = 512

    def mixed_supported(self, nodes, max_c=None):
        """max_c: the widest layer (2h, cin) the caller's kernels take -- the running-range schedule's per-channel state
        table in LDS (kMixedMaxC = 512) by default; the byte-code schedule (FrozenBackbone: one frozen grid per layer, no
        table) passes its own."""
        max_c = self._MIXED_MAX_C if max_c is None else max_c
        if not self.int8 or not all(hasattr(n, "quant_convbn1") for n in nodes) or nodes[0].stride != 2:
            return False
        if any(n.stride != 1 for n in nodes[1:]):
            return False
        h = nodes[0].quant_convbn3.conv.out_channels
        cin = nodes[0].quant_convbn1.conv.in_channels
        if (2 * h) % 4 or 2 * h > max_c or cin > max_c or len(nodes) + 1 > 250:
            return False
        convs = [n.quant_convbn1 for n in nodes] + [n.quant_convbn3 for n in nodes] + [nodes[0].quant_convbn5]
        return all(c.folded_int8() is not None for c in convs)

    @staticmethod
    def _death(L, h):
        """Units until logical channel L sits in the consumed half (logical index >= h)."""
        d = 1
        while L < h:
            if L == 0:
                return 1 << 20
            L, d = 2 * L, d + 1
        return d

    def _mixed_plan(self, nodes, in_logical, dev, in_gens=None):
        """Host bookkeeping of one layer: slot assignment, generations, permuted weights (cached until a
        weight changes).  in_logical: logical index of every physical input channel (None: identity)."""
        units = [self._unit(n) for n in nodes]
        for u_ in units:          # the QuantActs' device states exist before branch 1 forks to the side stream (a state is
            for k_ in ("a1", "a2", "a4", "sh"):     # created by a fill on the CURRENT stream: see FusedHeads.forward)
                if u_.get(k_) is not None:
                    u_[k_]._device_state(dev)
        convs = []
        for u in units:
            convs += [u[k] for k in ("c1", "c2", "c3", "c4", "c5") if k in u]
        key = (tuple((c.conv.weight.data_ptr(), c.conv.weight._version, c.bn.weight._version,
                      c.bn.running_var._version, c.bn.running_mean._version, c.bn.bias._version) for c in convs),
               tuple(in_logical) if in_logical is not None else None,
               tuple(in_gens) if in_gens is not None else None, dev)
        cache = self.__dict__.setdefault("_mixed_cache", {})
        ck = id(nodes[0])
        if ck in cache and cache[ck]["key"] == key:
            return cache[ck]
        h, cin = units[0]["h"], units[0]["cin"]
        C = 2 * h
        lin = list(in_logical) if in_logical is not None else list(range(cin))
        lin_t = torch.tensor(lin, device=dev, dtype=torch.long)
        i32 = lambda v: torch.tensor(v, device=dev, dtype=torch.int32)   # noqa: E731
        u8 = lambda v: torch.tensor(v, device=dev, dtype=torch.uint8)    # noqa: E731

        def pw_weights(convbn, cols, K, gens=None):
            """1x1 weights with input columns re-ordered: column p of the result is logical column cols[p]
            (-1: zero column).  Returns dict(w fp32 [Co,K], codes int8 [Co,Kpad], scale, colsum, bias); with the
            generation of every column (gens) also the (k-tile, generation) segments of the int8 kernel."""
            w, b = convbn.folded()
            codes, scale, colsum = convbn.folded_int8()
            Co = w.shape[0]
            cols_t = torch.tensor(cols, device=dev, dtype=torch.long)
            live = cols_t >= 0
            src = cols_t.clamp(min=0)
            w2 = w.reshape(Co, -1)[:, src] * live.to(w.dtype)
            kpad = (K + 63) // 64 * 64
            cp = torch.zeros(Co, kpad, dtype=torch.int8, device=dev)
            cp[:, :K] = codes[:, src] * live.to(torch.int8)
            out = dict(w=w2.contiguous(), codes=cp.contiguous(), scale=scale, colsum=colsum, bias=b.contiguous(),
                       Co=Co, K=K)
            return out

        plan = dict(key=key, h=h, cin=cin, C=C, units=[])
        logical, gen = [0] * C, [0] * C
        ngen = 0
        for k, (node, u) in enumerate(zip(nodes, units)):
            P = {}
            if k == 0:
                order = sorted(range(C), key=lambda L: (self._death(L, h), L))
                slot_of = {L: s_ for s_, L in enumerate(order)}
                w4, b4 = u["c4"].folded()
                P["w4"] = w4.reshape(cin, 9)[lin_t].contiguous()
                P["b4"] = b4[lin_t].contiguous()
                P["c5"] = pw_weights(u["c5"], lin, cin)
                P["c1"] = pw_weights(u["c1"], lin, cin, in_gens)
                P["c3"] = pw_weights(u["c3"], list(range(h)), h)
                P["omapA"] = i32([slot_of[2 * i] for i in range(h)])
                P["omapB"] = i32([slot_of[2 * i + 1] for i in range(h)])
                P["genA"], P["genB"] = ngen, ngen + 1
                for L, s_ in slot_of.items():
                    logical[s_] = L
                    gen[s_] = ngen + (L & 1)
                ngen += 2
            else:
                P2 = [p_ for p_ in range(C) if logical[p_] >= h]
                # generation 255 = "this physical column meets only zero weight codes in this unit's first 1x1 conv" (the
                # pass-through half): pwd3_kernel skips the 32-channel windows that hold nothing else (round 4)
                P["gen_in"] = u8([gen[p_] if logical[p_] >= h else 255 for p_ in range(C)])
                P["c1"] = pw_weights(u["c1"], [logical[p_] - h if logical[p_] >= h else -1 for p_ in range(C)], C,
                                     list(gen))
                P["c3"] = pw_weights(u["c3"], list(range(h)), h)
                fresh = sorted(range(h), key=lambda i: (self._death(2 * i + 1, h), i))
                omap = [0] * h
                for p_ in range(C):
                    if logical[p_] < h:
                        logical[p_] *= 2
                for i, p_ in zip(fresh, P2):
                    omap[i] = p_
                    logical[p_] = 2 * i + 1
                    gen[p_] = ngen
                P["omapB"] = i32(omap)
                P["genB"] = ngen
                ngen += 1
            w2, b2 = u["c2"].folded()
            P["w2"], P["b2"] = w2.reshape(h, 9).contiguous(), b2.contiguous()
            plan["units"].append(P)
        assert sorted(logical) == list(range(C))
        inv = [0] * C                  # physical slot of every logical channel (materialize; built once: a
        for p_, L_ in enumerate(logical):   # host-to-device copy is not allowed while a graph is captured)
            inv[L_] = p_
        plan.update(logical=list(logical), gen=u8(gen), gen_list=list(gen), ngen=ngen,
                    inv=torch.tensor(inv, device=dev, dtype=torch.long), gen_long=u8(gen).long(),
                    states=torch.zeros(ngen * 8, dtype=torch.int32, device=dev))
        cache[ck] = plan
        return plan

    def _fork_side(self, dev):
        """Route the following launches to the side stream (forked from the current stream), with the
        second set of arrival counters."""
        from . import _native as N_
        if self.__dict__.get("_side") is None or self._side.device != dev:
            self._side = torch.cuda.Stream(dev)
            aux = N_.lib().cdn_codenet_aux_workspace_bytes()
            self._ws2 = torch.zeros(aux // 4 + 64, device=dev)
        self._side.wait_stream(torch.cuda.current_stream(dev))
        self._main_launch = (self._stream, self._ws_ptr, self._ws_bytes)
        p2 = (self._ws2.data_ptr() + 255) // 256 * 256
        self._stream, self._ws_ptr = self._side.cuda_stream, p2
        self._ws_bytes = (self._ws2.numel() * 4 - (p2 - self._ws2.data_ptr())) // 256 * 256
        return True

    def _leave_side(self, dev):
        """Back to the main stream; returns the event that marks the end of the side-stream work."""
        ev = torch.cuda.Event()
        ev.record(self._side)
        self._stream, self._ws_ptr, self._ws_bytes = self._main_launch
        return ev

    def _pw_raw(self, a_ptr, a_q, a_gen, M, lda, Wt, relu, act, state_ptr, out_map, out_ptr, ldo):
        from . import _native as N_
        aa = self._act_args(act, self._dev)
        if state_ptr is not None:
            aa[2] = state_ptr
        rc = N_.lib().cdn_codenet_pointwise_mixed_forward(
            a_ptr, a_q, a_gen, M, Wt["K"], Wt["Co"], lda, ldo, Wt["w"].data_ptr(), Wt["codes"].data_ptr(),
            Wt["scale"].data_ptr(), Wt["colsum"].data_ptr(), Wt["bias"].data_ptr(), None, None, int(relu),
            out_map, *aa, self._ws_ptr, self._ws_bytes, out_ptr, self._stream)
        N_.check(rc, "cdn_codenet_pointwise_mixed_forward")

    recompute_pw1 = True      # A/B switch (tools/e2e_native_bench.py --no-recompute)
    range_first = True        # A/B switch: the recomputed conv's range pass before branch 1 is forked

    @staticmethod
    def _lib():
        from . import _native as N_
        return N_.lib()

    def _pwdw_raw(self, x_ptr, x_q, N, cin, H, W, ld_x, Wt, act_mid, C, w, b, act_out, out, ld_out, apply_only=False):
        """1x1 conv (range-only pass -> act_mid) recomputed inside the stride-2 depthwise (-> out, range of act_out).
        apply_only: the range-only pass has been issued by the caller (`_pw_raw` with out_ptr None)."""
        from . import _native as N_
        am, ao = self._act_args(act_mid, self._dev), self._act_args(act_out, out.device)
        fn = N_.lib().cdn_codenet_pwdw_s2_apply if apply_only else N_.lib().cdn_codenet_pwdw_s2_forward
        rc = fn(
            x_ptr, x_q, N, cin, H, W, ld_x, Wt["w"].data_ptr(), Wt["codes"].data_ptr(), Wt["scale"].data_ptr(),
            Wt["colsum"].data_ptr(), Wt["bias"].data_ptr(), am[0], am[1], am[2], C, w.data_ptr(), b.data_ptr(), ld_out,
            ao[0], ao[1], ao[2], ao[3], ao[4], ao[5], self._ws_ptr, self._ws_bytes, out.data_ptr(), self._stream)
        N_.check(rc, "cdn_codenet_pwdw_s2_forward")

    def _dw_raw(self, a_ptr, a_q, a_gen, N, C, H, W, stride, ld_in, w, b, act, out, ld_out):
        from . import _native as N_
        rc = N_.lib().cdn_codenet_dw3x3_mixed_forward(
            a_ptr, a_q, a_gen, N, C, H, W, 0, stride, ld_in, ld_out, w.data_ptr(), b.data_ptr(), None, None, 0,
            *self._act_args(act, out.device), self._ws_ptr, self._ws_bytes, out.data_ptr(), self._stream)
        N_.check(rc, "cdn_codenet_dw3x3_mixed_forward")

    def run_units_mixed(self, nodes, x, x_ld, x_in, Nb, H, W):
        """A layer (stride-2 unit + stride-1 units) without a physical shuffle.  x: channels-last
        [Nb*H*W, x_ld]; x_in: None (final values), an int (pointer of the ONE QuantAct state its
        pre-quantisation values are loaded with) or the layout dict of the previous layer.  Returns the
        layout dict(t=[M, C] pre-quantisation values, logical, gen (device uint8), states, C, H, W)."""
        dev = x.device
        self._prepare(dev)
        mixed_in = isinstance(x_in, dict)
        plan = self._mixed_plan(nodes, x_in["logical"] if mixed_in else None, dev,
                                x_in.get("gen_list") if mixed_in else None)
        units = [self._unit(n) for n in nodes]
        for u_ in units:          # the QuantActs' device states exist before branch 1 forks to the side stream (a state is
            for k_ in ("a1", "a2", "a4", "sh"):     # created by a fill on the CURRENT stream: see FusedHeads.forward)
                if u_.get(k_) is not None:
                    u_[k_]._device_state(dev)
        h, cin, C = plan["h"], plan["cin"], plan["C"]
        L = self._layer_bufs(nodes, h, cin, Nb, H, W, dev)
        ldh = L["ldh"]
        Mi, Mo = Nb * L["Hin"] * L["Win"], Nb * L["H"] * L["W"]
        qptr = lambda act: act._device_state(dev).data_ptr() if act is not None else None   # noqa: E731
        Y, S = L["ya"], plan["states"]
        sp = lambda g: S.data_ptr() + 32 * g   # noqa: E731
        a_q = (x_in["states"].data_ptr() if mixed_in else x_in)
        a_gen = x_in["gen"].data_ptr() if mixed_in else None
        with torch.no_grad():
            for k, (u, P) in enumerate(zip(units, plan["units"])):
                sh = u["sh"]
                if k == 0:
                    # the two branches of the stride-2 unit are independent until the shared QuantAct: branch 1
                    # (reference order: first) runs on a side stream with its own arrival counters, and branch
                    # 2's last conv -- the second call of the shared QuantAct -- waits for it
                    t4 = L["t4"]
                    recompute = (self.recompute_pw1 and not mixed_in and a_q is not None and u["a1"] is not None
                                 and self._lib().cdn_codenet_pwdw_s2_supported(Nb, cin, h, L["Hin"], L["Win"]))
                    if recompute and self.two_streams and self.range_first:
                        # the range-only pass of the recomputed conv BEFORE the fork: beside branch 1 it took 106 us of
                        # the critical path (both stream the stem's output), alone 55; branch 1 (memory-bound) then runs
                        # beside the VALU-bound recomputing kernel, which used to have the GPU to itself
                        self._pw_raw(x.data_ptr(), a_q, None, Mi, x_ld, P["c1"], True, u["a1"], None, None, None, 0)
                    ev = self._fork_side(dev) if self.two_streams else None
                    self._dw_raw(x.data_ptr(), a_q, a_gen, Nb, cin, L["Hin"], L["Win"], 2, x_ld, P["w4"], P["b4"],
                                 u["a4"], t4, t4.shape[1])
                    self._pw_raw(t4.data_ptr(), qptr(u["a4"]), None, Mo, t4.shape[1], P["c5"], True, sh,
                                 sp(P["genA"]), P["omapA"].data_ptr(), Y.data_ptr(), C)
                    if ev is not None:
                        ev = self._leave_side(dev)
                    if recompute:
                        # layer 1: the 1x1 conv (K = 24) recomputed inside the stride-2 depthwise -- its 58-channel
                        # fp32 output at input resolution (243 MB at batch 64, 512 x 512) is never stored
                        self._pwdw_raw(x.data_ptr(), a_q, Nb, cin, L["Hin"], L["Win"], x_ld, P["c1"], u["a1"], h,
                                       P["w2"], P["b2"], u["a2"], L["t2"], ldh,
                                       apply_only=self.two_streams and self.range_first)
                    else:
                        self._pw_raw(x.data_ptr(), a_q, a_gen, Mi, x_ld, P["c1"], True, u["a1"], None, None,
                                     self._t1s2(L).data_ptr(), ldh)
                        self._dw_raw(L["t1s2"].data_ptr(), qptr(u["a1"]), None, Nb, h, L["Hin"], L["Win"], 2, ldh,
                                     P["w2"], P["b2"], u["a2"], L["t2"], ldh)
                    if ev is not None:
                        torch.cuda.current_stream(dev).wait_event(ev)
                else:
                    self._pw_raw(Y.data_ptr(), S.data_ptr(), P["gen_in"].data_ptr(), Mo, C, P["c1"], True,
                                 u["a1"], None, None, L["t1"].data_ptr(), ldh)
                    self._dw_raw(L["t1"].data_ptr(), qptr(u["a1"]), None, Nb, h, L["H"], L["W"], 1, ldh,
                                 P["w2"], P["b2"], u["a2"], L["t2"], ldh)
                self._pw_raw(L["t2"].data_ptr(), qptr(u["a2"]), None, Mo, ldh, P["c3"], True, sh, sp(P["genB"]),
                             P["omapB"].data_ptr(), Y.data_ptr(), C)
        return dict(t=Y, logical=plan["logical"], gen=plan["gen"], gen_list=plan["gen_list"], states=S, C=C,
                    H=L["H"], W=L["W"], inv=plan["inv"], gen_long=plan["gen_long"])

    @staticmethod
    def materialize(layout):
        """The layer output in the reference's (logical) channel order with every channel fake-quantised by
        its generation's state -- what the module path holds; plain torch, for tests and hand-overs."""
        t, S = layout["t"], layout["states"].view(torch.float32).view(-1, 8)
        g = layout["gen_long"] if "gen_long" in layout else layout["gen"].long()
        scale, zp = S[g, 2], S[g, 3]
        q = (torch.round(scale * t - zp) + zp) / scale
        inv = layout.get("inv")
        if inv is None:
            inv = torch.empty(len(layout["logical"]), dtype=torch.long, device=t.device)
            inv[torch.tensor(layout["logical"], device=t.device)] = torch.arange(len(layout["logical"]),
                                                                                 device=t.device)
        return q[:, inv]

    def __call__(self, images):
        from . import _native as N_
        if not (images.is_cuda and images.dtype == torch.float32 and images.dim() == 4
                and images.shape[1] == 3):
            raise NotImplementedError("FusedBackbone needs a [N,3,H,W] float32 GPU tensor")
        images = images.contiguous()
        m, dev = self.model, images.device
        self._prepare(dev)
        Nb, _, R, R2 = images.shape
        qptr = lambda act: act._device_state(dev).data_ptr() if act is not None else None   # noqa: E731
        quant = hasattr(m.layer0[0], "folded")
        if quant:
            q0, act0 = m.layer0[0], m.layer0[1][1]
            conv0 = q0.conv
            q4, act4 = m.layer4[0], m.layer4[1][1]
            c4 = q4.conv.out_channels
        else:                                            # fp32: Sequential(conv, bn, relu)
            q0, act0, conv0 = (m.layer0[0], m.layer0[1]), None, m.layer0[0]
            q4, act4 = (m.layer4[0], m.layer4[1]), None
            c4 = m.layer4[0].out_channels
        s0 = conv0.stride[0]
        H, W = (R + 2 - 3) // s0 + 1, (R2 + 2 - 3) // s0 + 1
        key = (tuple(images.shape), dev)
        if self._bufs is None or self._bufs["key"] != key:
            self._bufs = dict(key=key, t0=torch.empty(Nb, H * W, 24, device=dev))
        B = self._bufs
        with torch.no_grad():
            # ---- layer0: dense 3x3 conv + folded BN + ReLU, range of its QuantAct ------------------
            w0, b0 = self._folded(q0)
            rc = N_.lib().cdn_codenet_stem_forward(
                images.data_ptr(), Nb, R, R2, 24, s0, w0.reshape(24, 27).data_ptr(), b0.data_ptr(), 1,
                *self._act_args(act0, dev), self._ws_ptr, self._ws_bytes, B["t0"].data_ptr(), self._stream)
            N_.check(rc, "cdn_codenet_stem_forward")
            x, x_ld, x_q = B["t0"], 24, qptr(act0)          # pre-quantisation values + state
            pooled = (len(m.layer0[1]) == 3) if quant else (len(m.layer0) == 4)
            if pooled:                                      # "S2 + MaxPool" stems (configs b, e)
                Hp, Wp = (H - 1) // 2 + 1, (W - 1) // 2 + 1
                if B.get("tp") is None or B["tp"].shape != (Nb, Hp * Wp, 24):
                    B["tp"] = torch.empty(Nb, Hp * Wp, 24, device=dev)
                rc = N_.lib().cdn_codenet_maxpool3x3s2_nhwc_forward(x.data_ptr(), x_q, Nb, 24, H, W,
                                                                    B["tp"].data_ptr(), self._stream)
                N_.check(rc, "cdn_codenet_maxpool3x3s2_nhwc_forward")
                x, x_q, H, W = B["tp"], None, Hp, Wp         # final values from here on
            lay = None                                       # layout dict while the layers run unshuffled
            for name in ("layer1", "layer2", "layer3"):
                nodes = list(getattr(m, name))
                if quant and self.shuffle_free and self.mixed_supported(nodes):
                    lay = self.run_units_mixed(nodes, x, x_ld, lay if lay is not None else x_q, Nb, H, W)
                    x, x_ld, H, W = lay["t"], lay["C"], lay["H"], lay["W"]
                else:
                    if lay is not None:                      # hand-over into the interleaving path
                        x, lay = self.materialize(lay).contiguous(), None
                        x_q = None
                    x, x_ld, H, W = self.run_units(nodes, x, x_ld, x_q, Nb, H, W)
                x_q = None
            # ---- layer4: 1x1 conv + folded BN + ReLU, range of its QuantAct ---------------------------
            if B.get("out") is None or B["out"].shape != (Nb, H * W, c4):
                B["out"] = torch.empty(Nb, H * W, c4, device=dev)
            if lay is not None and q4.folded_int8() is not None and x_ld <= self._MIXED_MAX_C:
                W4 = self._l4_weights(q4, lay["logical"], dev, lay.get("gen_list"))
                self._pw_raw(x.data_ptr(), lay["states"].data_ptr(), lay["gen"].data_ptr(), Nb * H * W, x_ld, W4,
                             True, act4, None, None, B["out"].data_ptr(), c4)
            else:
                if lay is not None:
                    x = self.materialize(lay).contiguous()
                self._pw(x.data_ptr(), None, Nb * H * W, x_ld, q4, True, act4, B["out"], 0)
            if c4 % 4:
                # the channels-last hand-over into stage 0 needs C % 4 == 0; CoDeNet2x's 2153 channels are
                # materialised as the NCHW tensor the stage takes from a PyTorch backbone (fake-quantised)
                if B.get("out_nchw") is None or B["out_nchw"].shape != (Nb, c4, H, W):
                    B["out_nchw"] = torch.empty(Nb, c4, H, W, device=dev)
                rc = N_.lib().cdn_codenet_unpack_nchw(B["out"].data_ptr(), qptr(act4), B["out_nchw"].data_ptr(),
                                                      Nb, c4, H, W, 0, self._stream)
                N_.check(rc, "cdn_codenet_unpack_nchw")
                return B["out_nchw"], None, None
        return B["out"], qptr(act4), (H, W)


class FrozenBackbone:
    """layer0 .. layer4 of a W4A8 ``PoseShuffleNetV2`` with every QuantAct FROZEN (``running_stat = False``: the
    serving mode, quant_modules.py:172,181,203-219 skipped) on BYTE CODES: every activation crosses HBM as one byte
    per element (the code of its QuantAct), every 1x1 conv is the int8-MFMA kernel on the codes as they lie in memory
    (cdn_codenet_pointwise_q8_strided_forward), the depthwise convs and the stem write codes directly
    (cdn_codenet_dw3x3_q8_forward, cdn_codenet_stem_q8_forward); no range epilogues, no arrival counters.  With frozen
    ranges the layer's shared block-output QuantAct is ONE fixed grid, so the "generations" of the running-range
    schedule (FusedBackbone) collapse and a layer is one int8 tensor whose channel slots never move (same slot
    assignment and permuted weight codes as FusedBackbone._mixed_plan).  Returns (codes [N, H*W, 1024] int8, state
    pointer of layer4's QuantAct, (H, W)) -- what ``FrozenHotPath.forward_codes`` takes.

    Arithmetic: the depthwise chains and the stem are those of the fp32 kernels on the values (q + zp) / scale
    (bit-identical); a unit's first 1x1 conv is an EXACT integer sum here where the running-range schedule
    accumulates exact products in fp32 (pwd3_kernel), so results agree with FusedBackbone at running_stat False up
    to single code flips (tests/test_gpu_backbone.py).  A saturated code sets the overflow flag (``overflowed()``)."""

    _MAX_C = 1024      # widest layer of the byte-code schedule (no per-channel table: int8 rows of up to 1024 codes)

    def __init__(self, model, fuse_dwpw=True, two_streams=True):
        self.model = model
        self._fb = FusedBackbone(model)
        self._bufs = None
        self.fuse_dwpw = fuse_dwpw          # a unit's depthwise inside its second 1x1 conv (cdn_codenet_dwpw_q8_forward)
        self.two_streams = two_streams      # the two branches of a stride-2 unit on two streams

    @staticmethod
    def supported(model):
        from .portable_quantizer.quant_modules import QuantAct
        if not FusedBackbone.supported(model) or not hasattr(model.layer0[0], "folded"):
            return False
        fb = FusedBackbone(model)
        if not (len(model.layer0[1]) == 2 or (len(model.layer0[1]) == 3 and FusedBackbone._is_pool(model.layer0[1][2]))):
            return False                                     # (stem = conv, [ReLU, QuantAct] or [ReLU, QuantAct, MaxPool])
        for name in ("layer1", "layer2", "layer3"):      # (CoDeNet2x: layer3 is 976 channels wide; round 5)
            if not fb.mixed_supported(list(getattr(model, name)), max_c=FrozenBackbone._MAX_C):
                return False
        q4 = model.layer4[0]
        if q4.folded_int8() is None:
            return False
        acts = [a for n in ("layer0", "layer1", "layer2", "layer3", "layer4") for a in getattr(model, n).modules()
                if isinstance(a, QuantAct)]
        return bool(acts) and len(acts) <= 48 and all(act_fusable(a) and not a.running_stat for a in acts)

    def still_frozen(self):
        """The per-forward check (supported() walks the module tree once, when the object is built)."""
        from .portable_quantizer.quant_modules import QuantAct
        acts = self.__dict__.get("_acts")
        if acts is None:
            acts = self._acts = [a for n in ("layer0", "layer1", "layer2", "layer3", "layer4")
                                 for a in getattr(self.model, n).modules() if isinstance(a, QuantAct)]
        return not any(a.running_stat for a in acts)

    def overflowed(self):
        if self._bufs is None:
            return False
        return self._bufs["overflow"].any()

    def _alloc(self, images, key):
        import ctypes
        from .portable_quantizer.quant_modules import QuantAct
        m, dev = self.model, images.device
        acts = []
        for n in ("layer0", "layer1", "layer2", "layer3", "layer4"):
            for a in getattr(m, n).modules():
                if isinstance(a, QuantAct) and all(a is not b for b in acts):
                    acts.append(a)
        arr = ctypes.c_void_p * len(acts)
        self._bufs = dict(
            key=key, layers={}, overflow=OverflowFlags(len(acts), dev), acts=acts, n_acts=len(acts),
            p_min=arr(*[a.x_min.data_ptr() for a in acts]), p_max=arr(*[a.x_max.data_ptr() for a in acts]),
            p_state=arr(*[a._device_state(dev).data_ptr() for a in acts]))
        self._bufs["act_index"] = {id(a): i for i, a in enumerate(acts)}
        for i, a in enumerate(acts):            # word i: the launches that write act i's codes
            self._bufs["overflow"].name(i, [a])

    def _of(self, *acts):
        """the flag word of the launch writing `acts[0]`'s codes (a fused launch writing two QuantActs' codes is
        attributed to both)"""
        B = self._bufs
        i = B["act_index"][id(acts[0])]                 # (the acts list keeps the modules alive: ids are stable)
        for extra in acts[1:]:                          # (fused depthwise -> pointwise launches; recorded once)
            if (i, id(extra)) not in B.setdefault("attributed", set()):
                B["attributed"].add((i, id(extra)))
                B["overflow"].who[i].append(extra)      # (the backbone's flags are never sliced: word i == who[i])
        return B["overflow"].ptr(i)

    @staticmethod
    def _ld(c):
        return (c + 15) // 16 * 16          # 16-byte aligned byte rows (the int8 pointwise loads 16 bytes per lane)

    def _layer(self, name, nodes, x8, x_ld, x_state, in_logical, Nb, H, W):
        from . import _native as N_
        fb, lib, B = self._fb, N_.lib(), self._bufs
        dev = x8.device
        plan = fb._mixed_plan(nodes, in_logical, dev, None)
        units = [fb._unit(n) for n in nodes]
        h, cin, C = plan["h"], plan["cin"], plan["C"]
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        Mi, Mo = Nb * H * W, Nb * Ho * Wo
        ldh, ldc, ldi = self._ld(h), self._ld(C), self._ld(cin)
        L = B["layers"].get(name)
        if L is None:
            z = lambda m_, c_: torch.zeros(m_, c_, dtype=torch.int8, device=dev)   # noqa: E731
            L = B["layers"][name] = dict(Y=z(Mo, ldc), t4=z(Mo, ldi), t1s2=z(Mi, ldh), t1=z(Mo, ldh), t2=z(Mo, ldh))
        main = torch.cuda.current_stream(dev)
        st = main.cuda_stream
        qp = lambda act: act._device_state(dev).data_ptr()   # noqa: E731

        def pw(a, a_state, M, K, lda, Wt, act, out, ldo, omap, st=st):
            rc = lib.cdn_codenet_pointwise_q8_strided_forward(
                a.data_ptr(), a_state, M, K, Wt["Co"], lda, ldo, Wt["codes"].data_ptr(), Wt["scale"].data_ptr(),
                Wt["colsum"].data_ptr(), Wt["bias"].data_ptr(), 1, omap, qp(act), out.data_ptr(), None, self._of(act), st)
            N_.check(rc, "cdn_codenet_pointwise_q8_strided_forward")

        def dw(a, a_state, Cc, Hs, Ws, stride, ld_in, w, b, act, out, ld_out, st=st):
            rc = lib.cdn_codenet_dw3x3_q8_forward(a.data_ptr(), a_state, Nb, Cc, Hs, Ws, stride, ld_in, ld_out,
                                                  w.data_ptr(), b.data_ptr(), 0, qp(act), out.data_ptr(), self._of(act), st)
            N_.check(rc, "cdn_codenet_dw3x3_q8_forward")

        def dwpw(a, a_state, Cc, Hs, Ws, stride, ld_in, w, b, act, Wt, out_act, omap, tmp=None, st=st):
            """depthwise (output codes of `act`) -> 1x1 conv -> ReLU -> codes of out_act into Y's slots: one launch
            where cdn_codenet_dwpw_q8_supported, else the two kernels through the scratch tensor `tmp`"""
            if self.fuse_dwpw and lib.cdn_codenet_dwpw_q8_supported(Cc, Hs, Ws, stride, Wt["Co"]):
                rc = lib.cdn_codenet_dwpw_q8_forward(
                    a.data_ptr(), a_state, Nb, Cc, Hs, Ws, stride, ld_in, w.data_ptr(), b.data_ptr(), 0, qp(act),
                    Wt["Co"], Wt["codes"].data_ptr(), Wt["scale"].data_ptr(), Wt["colsum"].data_ptr(),
                    Wt["bias"].data_ptr(), 1, ldc, omap, qp(out_act), Y.data_ptr(), self._of(out_act, act), st)
                N_.check(rc, "cdn_codenet_dwpw_q8_forward")
                return
            tmp = L["t2"] if tmp is None else tmp
            ldt = tmp.shape[1]
            dw(a, a_state, Cc, Hs, Ws, stride, ld_in, w, b, act, tmp, ldt, st=st)
            pw(tmp, qp(act), Mo, Cc, ldt, Wt, out_act, Y, ldc, omap, st=st)

        Y = L["Y"]
        with torch.no_grad():
            for k, (u, P) in enumerate(zip(units, plan["units"])):
                sh = u["sh"]
                if k == 0:
                    # the two branches of a stride-2 unit read the same input and write disjoint slots of Y: branch 1
                    # on a side stream (these launches leave most of the chip idle on their own)
                    side = None
                    if self.two_streams:
                        side = B.get("side")
                        if side is None:
                            side = B["side"] = torch.cuda.Stream(dev)
                        side.wait_stream(main)
                    # branch 1: dw s2 -> QuantAct -> pw -> ReLU -> shared QuantAct (slots omapA)
                    dwpw(x8, x_state, cin, H, W, 2, x_ld, P["w4"], P["b4"], u["a4"], P["c5"], sh, P["omapA"].data_ptr(),
                         tmp=L["t4"], st=side.cuda_stream if side is not None else st)
                    # branch 2: pw -> ReLU -> QuantAct -> dw s2 -> QuantAct -> pw -> ReLU -> shared QuantAct (omapB)
                    pw(x8, x_state, Mi, cin, x_ld, P["c1"], u["a1"], L["t1s2"], ldh, None)
                    dwpw(L["t1s2"], qp(u["a1"]), h, H, W, 2, ldh, P["w2"], P["b2"], u["a2"], P["c3"], sh,
                         P["omapB"].data_ptr())
                    if side is not None:
                        main.wait_stream(side)
                else:
                    pw(Y, qp(sh), Mo, C, ldc, P["c1"], u["a1"], L["t1"], ldh, None)
                    dwpw(L["t1"], qp(u["a1"]), h, Ho, Wo, 1, ldh, P["w2"], P["b2"], u["a2"], P["c3"], sh,
                         P["omapB"].data_ptr())
        return Y, ldc, qp(units[0]["sh"]), plan["logical"], Ho, Wo

    def __call__(self, images):
        from . import _native as N_
        if not (images.is_cuda and images.dtype == torch.float32 and images.dim() == 4 and images.shape[1] == 3):
            raise NotImplementedError("FrozenBackbone needs a [N,3,H,W] float32 GPU tensor")
        images = images.contiguous()
        m, dev, lib = self.model, images.device, N_.lib()
        Nb, _, R, R2 = images.shape
        # (the cached pointer arrays name the QuantActs' range buffers: their addresses are part of the key)
        stale = (self._bufs is None or self._bufs["key"][:2] != (tuple(images.shape), dev)
                 or self._bufs["key"][2:] != tuple(a.x_min.data_ptr() for a in self._bufs["acts"]))
        if stale:
            self._alloc(images, None)
            self._bufs["key"] = (tuple(images.shape), dev) + tuple(a.x_min.data_ptr() for a in self._bufs["acts"])
        B = self._bufs
        st = torch.cuda.current_stream(dev).cuda_stream
        bits, _, _ = uniform_act_settings(B["acts"], "FrozenBackbone")
        N_.check(lib.cdn_quantact_frozen_params(B["n_acts"], B["p_min"], B["p_max"], B["p_state"], bits, st),
                 "cdn_quantact_frozen_params")
        q0, act0 = m.layer0[0], m.layer0[1][1]
        q4, act4 = m.layer4[0], m.layer4[1][1]
        s0 = q0.conv.stride[0]
        H, W = (R + 2 - 3) // s0 + 1, (R2 + 2 - 3) // s0 + 1
        with torch.no_grad():
            w0, b0 = self._fb._folded(q0)
            if B.get("x0") is None:
                B["x0"] = torch.zeros(Nb, H * W, 32, dtype=torch.int8, device=dev)
            rc = lib.cdn_codenet_stem_q8_forward(images.data_ptr(), Nb, R, R2, 24, s0, w0.reshape(24, 27).data_ptr(),
                                                 b0.data_ptr(), 1, act0._device_state(dev).data_ptr(),
                                                 B["x0"].data_ptr(), 32, self._of(act0), st)
            N_.check(rc, "cdn_codenet_stem_q8_forward")
            x8, x_ld, x_state, logical = B["x0"], 32, act0._device_state(dev).data_ptr(), None
            if len(m.layer0[1]) == 3:                        # "S2 + MaxPool" stems (README configs b, e): pool the codes
                Hp, Wp = (H - 1) // 2 + 1, (W - 1) // 2 + 1
                if B.get("x0p") is None:
                    B["x0p"] = torch.zeros(Nb, Hp * Wp, 32, dtype=torch.int8, device=dev)
                rc = lib.cdn_codenet_maxpool3x3s2_q8_forward(x8.data_ptr(), Nb, 24, H, W, 32, 32, B["x0p"].data_ptr(), st)
                N_.check(rc, "cdn_codenet_maxpool3x3s2_q8_forward")
                x8, H, W = B["x0p"], Hp, Wp
            for name in ("layer1", "layer2", "layer3"):
                x8, x_ld, x_state, logical, H, W = self._layer(name, list(getattr(m, name)), x8, x_ld, x_state, logical,
                                                               Nb, H, W)
            c4 = q4.conv.out_channels
            ld4 = c4 if c4 % 4 == 0 else self._ld(c4)      # (CoDeNet2x: 2153 codes in rows of 2160 bytes, the pad stays 0)
            if B.get("out") is None:
                B["out"] = torch.zeros(Nb, H * W, ld4, dtype=torch.int8, device=dev)
            W4 = self._fb._l4_weights(q4, logical, dev, None)
            rc = lib.cdn_codenet_pointwise_q8_strided_forward(
                x8.data_ptr(), x_state, Nb * H * W, W4["K"], c4, x_ld, ld4, W4["codes"].data_ptr(),
                W4["scale"].data_ptr(), W4["colsum"].data_ptr(), W4["bias"].data_ptr(), 1, None,
                act4._device_state(dev).data_ptr(), B["out"].data_ptr(), None, self._of(act4), st)
            N_.check(rc, "cdn_codenet_pointwise_q8_strided_forward (layer4)")
        return B["out"], act4._device_state(dev).data_ptr(), (H, W)
