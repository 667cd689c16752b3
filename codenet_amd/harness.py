"""The caller side of the hot path: a ``PoseShuffleNetV2`` with the
reference's module tree / state-dict keys (lib/models/networks/shufflenetv2_dcn.py:57-114,189-330) whose
``deconv_layers`` are this package's deform modules, ``ctdet_decode`` (lib/models/decode.py:10-16,110-127,474-505; a torch
restatement for CPU tensors and ``ctdet_decode_native`` on the HIP kernels) and the body of ``CtdetDetector.process``
(lib/detectors/ctdet.py:29-46).  Module by module every operator of the three deform stages runs on the hand-written
kernels; ``enable_fused()`` runs the WHOLE network on them -- backbone (pipeline.FusedBackbone), stages
(pipeline.FusedHotPath), heads (pipeline.FusedHeads) -- and ``enable_fused(frozen_codes=True)`` the byte-code serving
schedule (pipeline.FrozenBackbone / FrozenHotPath / FusedHeads.forward_codes); ``capture_process`` records network +
decode as one HIP graph.
"""
import hashlib

import torch
import torch.nn as nn
import torch.nn.functional as F

from .modules import dcn_deform_conv
from .portable_quantizer.quant_modules import channel_shuffle

BN_MOMENTUM = 0.1


class BaseNode(nn.Module):
    """ShuffleNetV2 unit: stride 1 = split / transform one half / concat / shuffle; stride 2 = two
    transformed branches (reference :57-114)."""

    def __init__(self, inp, oup, stride, batch_norm=nn.BatchNorm2d, conv_kernel=nn.Conv2d):
        super().__init__()
        self.stride = stride
        half = oup // 2

        def pw(i, o):
            return [nn.Conv2d(i, o, 1, 1, 0, bias=False), batch_norm(o, momentum=BN_MOMENTUM)]

        def dw(c, s):
            return [conv_kernel(c, c, 3, s, 1, groups=c, bias=False), batch_norm(c, momentum=BN_MOMENTUM)]

        if stride == 1:
            self.b2 = nn.Sequential(*pw(half, half), nn.ReLU(inplace=True), *dw(half, 1),
                                    *pw(half, half), nn.ReLU(inplace=True))
        elif stride == 2:
            self.b1 = nn.Sequential(*dw(inp, 2), *pw(inp, half), nn.ReLU(inplace=True))
            self.b2 = nn.Sequential(*pw(inp, half), nn.ReLU(inplace=True), *dw(half, 2),
                                    *pw(half, half), nn.ReLU(inplace=True))

    def forward(self, x):
        if self.stride == 1:
            half = x.shape[1] // 2
            x1, x2 = x[:, :half], self.b2(x[:, half:])
        else:
            x1, x2 = self.b1(x), self.b2(x)
        return channel_shuffle(torch.cat((x1, x2), 1), 2)


class PoseShuffleNetV2(nn.Module):
    """Backbone (stride 32) -> three deform up-sampling stages -> heads (reference :189-330).
    forward returns ``[ {head: tensor} ]`` like the reference."""

    def __init__(self, heads, head_conv, w2=None, deform=False, maxpool=False):
        super().__init__()
        self.w2 = w2
        self.deform_backbone = deform
        self.heads = heads
        self.deconv_with_bias = False
        self.channels = [24, 244, 488, 976, 2153] if w2 else [24, 116, 232, 464, 1024]
        c = self.channels
        stem = [nn.Conv2d(3, c[0], 3, 2 if maxpool else 4, 1, bias=False),
                nn.BatchNorm2d(c[0], momentum=BN_MOMENTUM), nn.ReLU(inplace=True)]
        if maxpool:
            stem.append(nn.MaxPool2d(kernel_size=3, stride=2, padding=1))
        self.layer0 = nn.Sequential(*stem)
        # deform=True (reference :216-230; its factory never passes it): the units' 3x3 convs are CoDeNet operators
        kern = dcn_deform_conv.DeformConvWithOffsetScaleBoundPositive if deform else nn.Conv2d
        for idx, reps in enumerate([3, 7, 3]):
            nodes = [BaseNode(c[idx], c[idx + 1], 2, nn.BatchNorm2d, kern)]
            nodes += [BaseNode(c[idx], c[idx + 1], 1, nn.BatchNorm2d, kern) for _ in range(reps)]
            setattr(self, "layer%d" % (idx + 1), nn.Sequential(*nodes))
        self.layer4 = nn.Sequential(nn.Conv2d(c[3], c[4], 1, 1, 0, bias=False),
                                    nn.BatchNorm2d(c[4], momentum=BN_MOMENTUM), nn.ReLU(inplace=True))
        layers = []
        planes_in = [c[4], 256, 128]
        for cin, cout in zip(planes_in, [256, 128, 64]):
            layers += [dcn_deform_conv.DeformConvWithOffsetScaleBoundPositive(
                           cin, cout, 3, 1, 1, groups=cout, bias=False, hidden_state=128,
                           BN_MOMENTUM=BN_MOMENTUM),
                       nn.BatchNorm2d(cout, momentum=BN_MOMENTUM), nn.ReLU(inplace=True),
                       nn.Upsample(scale_factor=2, mode="nearest")]
        self.deconv_layers = nn.Sequential(*layers)
        for head, classes in heads.items():
            if head_conv > 0:
                fc = nn.Sequential(
                    nn.Conv2d(64, head_conv, 1, 1, 0, bias=False),
                    nn.BatchNorm2d(head_conv, momentum=BN_MOMENTUM), nn.ReLU(inplace=True),
                    nn.Conv2d(head_conv, head_conv, 3, 1, 1, groups=head_conv, bias=False),
                    nn.BatchNorm2d(head_conv, momentum=BN_MOMENTUM), nn.ReLU(inplace=True),
                    nn.Conv2d(head_conv, classes, kernel_size=1, stride=1, padding=0, bias=True))
            else:
                fc = nn.Conv2d(64, classes, kernel_size=1, stride=1, padding=0, bias=True)
            setattr(self, head, fc)

    def enable_fused(self, flag=True, backbone=True, frozen_codes=False, frozen_backbone=True):
        """Inference on GPU tensors: run deconv_layers AND the heads on the fused HIP schedules
        (pipeline.FusedHotPath.forward_nhwc -> pipeline.FusedHeads; nothing is materialised between
        them) and, for a W4A8 model (backbone=True), layer0..layer4 on pipeline.FusedBackbone.  The
        returned tensors are static buffers, overwritten by the next call.

        frozen_codes (serving mode; needs every QuantAct of deconv_layers at running_stat False, e.g. after
        pipeline.set_running_stat(model, False)): the three deform stages run on the byte-code schedule
        (pipeline.FrozenHotPath, chained scale sums), and -- frozen_backbone, when the QuantActs of layer0..layer4
        are frozen too and pipeline.FrozenBackbone.supported(model) -- the backbone on byte codes as well, so that
        every activation between the image and the heads crosses HBM as one byte.  The reference does not clamp
        activation codes, a byte must: check ``frozen_overflowed()`` after a batch and recompute it with frozen_codes
        off if it says True (pipeline.cover_frozen_ranges widens EMA ranges over calibration batches beforehand)."""
        from .portable_quantizer.quant_modules import QuantAct
        self._fused = bool(flag)
        self._fused_backbone = bool(backbone)
        self._frozen_codes = bool(frozen_codes)
        self._frozen_backbone = bool(frozen_backbone)
        self._fpath = self._fheads = self._fbackbone = self._ffrozen = self._fzbackbone = self._fzheads = None
        # the QuantActs whose settings decide _fused_ok(): collected once (the module tree is fixed after
        # quantize_shufflenetv2_dcn), so a forward reads 5 attributes of ~70 modules instead of walking the tree
        self.__dict__["_fused_acts"] = [a for a in self.modules() if isinstance(a, QuantAct)]
        self.__dict__["_fused_ok_cache"] = {}
        self.__dict__.pop("_stage_acts", None)           # re-collected after a re-quantisation
        self.__dict__["_fzbackbone_tried"] = None
        return self

    def _stage_acts_frozen(self):
        from .portable_quantizer.quant_modules import QuantAct
        acts = self.__dict__.get("_stage_acts")
        if acts is None:
            acts = self.__dict__["_stage_acts"] = [a for a in self.deconv_layers.modules() if isinstance(a, QuantAct)]
        return bool(acts) and not any(a.running_stat for a in acts)

    def frozen_overflowed(self):
        """True when a code of the byte-code stage schedule saturated since the last call (synchronises, resets)."""
        hit = False
        for f in (getattr(self, "_ffrozen", None), getattr(self, "_fzbackbone", None)):
            if f is not None and f.overflowed():
                hit = True
        return hit

    def _fused_ok(self, x):
        """The fused schedules implement the reference's default QuantAct settings (stored planes beyond the
        LDS-resident gather -- inputs above ~1100 px -- are gathered from global memory); anything else
        (--act-percentile in backbone / heads, symmetric activations) keeps the module-by-module path -- decided
        BEFORE any kernel runs, so no QuantAct state is half-updated.  Cached per (input shape, QuantAct
        configuration)."""
        from . import pipeline
        cfg = tuple((a.percentile, a.quant_mode, a.full_precision_flag, a.activation_bit, a.running_stat,
                     getattr(a, "global_range", False)) for a in self.__dict__["_fused_acts"])
        key = (tuple(x.shape), cfg)
        cache = self.__dict__.setdefault("_fused_ok_cache", {})
        if key not in cache:
            Nb, _, R, R2 = x.shape
            stem = self.layer0[0].conv if hasattr(self.layer0[0], "conv") else self.layer0[0]
            down = stem.stride[0] * (2 if any(isinstance(m, nn.MaxPool2d) for m in self.layer0.modules()) else 1)
            h, w = R // (down * 8), R2 // (down * 8)
            ok = (R % (down * 8) == 0 and R2 % (down * 8) == 0 and h > 0 and w > 0
                  and pipeline.FusedHotPath.supported(self.deconv_layers, (Nb, self.channels[4], h, w))
                  and pipeline.FusedHeads.supported({hd: getattr(self, hd) for hd in self.heads}))
            if not ok:
                import warnings
                warnings.warn("codenet_amd: enable_fused() does not cover this model configuration / input "
                              "shape %s; running the module-by-module path (the deform stages alone stay on the "
                              "fused schedule where it implements them, e.g. --act-percentile)" % (tuple(x.shape),))
            cache[key] = ok
        return cache[key]

    def forward(self, x):
        if getattr(self, "_fused", False) and x.is_cuda and not torch.is_grad_enabled() \
                and self._fused_ok(x):
            from . import pipeline
            if self._fpath is None:
                self._fpath = pipeline.FusedHotPath(self.deconv_layers)
                self._fheads = pipeline.FusedHeads({h: getattr(self, h) for h in self.heads})
                self._fbackbone = (pipeline.FusedBackbone(self)
                                   if self._fused_backbone and pipeline.FusedBackbone.supported(self) else None)
            stages = self._fpath
            if getattr(self, "_frozen_codes", False) and self._stage_acts_frozen() and self._frozen_planes_fit(x):
                if self._ffrozen is None:
                    self._ffrozen = pipeline.FrozenHotPath(self.deconv_layers, chain_scale=True)
                if self._fzbackbone is None and self._fbackbone is not None and self._frozen_backbone:
                    # decided per QuantAct configuration, not once: the backbone may be frozen after the stages
                    cfg = tuple(a.running_stat for a in self.__dict__["_fused_acts"])
                    if self.__dict__.get("_fzbackbone_tried") != cfg:
                        self.__dict__["_fzbackbone_tried"] = cfg
                        if pipeline.FrozenBackbone.supported(self):
                            self._fzbackbone = pipeline.FrozenBackbone(self)
                stages = self._ffrozen
                # every QuantAct of the backbone frozen too: byte codes end to end
                if self._fzbackbone is not None and self._fzbackbone.still_frozen():
                    # the QuantAct parameters of the stages and the heads in the backbone's first launch (once their
                    # buffers exist: from the second call on) -- two launches fewer in the chain
                    also, sb = None, stages._bufs
                    hb = self._fzheads._bufs if self._fzheads is not None else None
                    if (sb is not None and hb is not None and hb.get("acts") is not None and sb.get("acts") is not None
                            and getattr(self, "merge_frozen_params", True)):      # (A/B switch: tools/e2e_frozen_bench.py)
                        also = (list(sb["acts"]) + list(hb["acts"]), sb.get("sums_all"))
                    feat8, fq, hw = self._fzbackbone(x, also=also)
                    cov = self._fzbackbone.covered
                    if cov is not None:
                        stages.params_covered = cov
                        self._fzheads.params_covered = cov
                    r8, rq, last = stages.forward_codes(feat8, fq, hw)
                    if r8.dtype == torch.int8:
                        if self._fzheads is None:
                            self._fzheads = pipeline.FusedHeads({h: getattr(self, h) for h in self.heads})
                        if self._fzheads.codes_supported(last):          # the heads on byte codes as well
                            return [self._fzheads.forward_codes(r8, rq, last, stages.head_flags())]
                    return [self._fheads(*stages.expand(r8, rq, last))]
            if self._fbackbone is not None:       # W4A8: the whole network on the HIP kernels
                feat, fq, hw = self._fbackbone(x)       # hw None: an NCHW tensor (odd channel count)
                return [self._fheads(*stages.forward_nhwc(feat, fq, hw))]
            x = self.layer4(self.layer3(self.layer2(self.layer1(self.layer0(x)))))
            return [self._fheads(*self._fpath.forward_nhwc(x))]
        x = self.layer4(self.layer3(self.layer2(self.layer1(self.layer0(x)))))
        if getattr(self, "_fused", False) and x.is_cuda and not torch.is_grad_enabled() and self._stages_fused_ok(x):
            # enable_fused() on a configuration the fused heads / backbone do not implement (--act-percentile): the three
            # deform stages still run on the fused schedule (one C call per stage + unpack), the rest module by module
            from . import pipeline
            if self._fpath is None:
                self._fpath = pipeline.FusedHotPath(self.deconv_layers)
            x = self._fpath(x)
            return [{head: getattr(self, head)(x) for head in self.heads}]
        from .functions.codenet_stage import forward_stage_blocks
        x = forward_stage_blocks(self.deconv_layers, x)      # == self.deconv_layers(x); fused blocks in the QAT step
        return [{head: getattr(self, head)(x) for head in self.heads}]

    def _frozen_planes_fit(self, x):
        """The byte-code schedules need every stage's stored plane in LDS; larger inputs keep the fp32 fused schedule
        (frozen ranges included).  Cached per input shape."""
        from . import pipeline
        key = ("fzfit", tuple(x.shape))
        cache = self.__dict__.setdefault("_fused_ok_cache", {})
        if key not in cache:
            Nb, _, R, R2 = x.shape
            stem = self.layer0[0].conv if hasattr(self.layer0[0], "conv") else self.layer0[0]
            down = stem.stride[0] * (2 if any(isinstance(m, nn.MaxPool2d) for m in self.layer0.modules()) else 1) * 8
            cache[key] = bool(pipeline.FrozenHotPath.planes_fit(self.deconv_layers,
                                                                (Nb, self.channels[4], R // down, R2 // down)))
        return cache[key]

    def _stages_fused_ok(self, feat):
        from . import pipeline
        cfg = tuple((a.percentile, a.quant_mode, a.full_precision_flag, a.activation_bit, a.running_stat,
                     getattr(a, "global_range", False)) for a in self.__dict__.get("_fused_acts", ()))
        key = ("stages", tuple(feat.shape), cfg)
        cache = self.__dict__.setdefault("_fused_ok_cache", {})
        if key not in cache:
            cache[key] = bool(pipeline.FusedHotPath.supported(self.deconv_layers, tuple(feat.shape)))
        return cache[key]


def fill_state_dict_(model, seed=317):
    """Deterministic, construction-order independent synthetic weights: every tensor is drawn from a
    generator seeded by (seed, state-dict key), so two models with the same keys (this package's
    and the reference's) get identical values.  Scales keep activations O(1) through the net and
    make conv_scale non-degenerate."""
    sd = model.state_dict()
    with torch.no_grad():
        for key in sorted(sd):
            t = sd[key]
            if not t.dtype.is_floating_point:
                continue
            h = int(hashlib.sha256(("%d:%s" % (seed, key)).encode()).hexdigest()[:12], 16)
            g = torch.Generator().manual_seed(h)
            shape = tuple(t.shape)
            if key.endswith("running_var"):
                v = torch.rand(shape, generator=g) + 0.5
            elif key.endswith("running_mean"):
                v = torch.randn(shape, generator=g) * 0.1
            elif key.endswith(("x_min", "x_max")):
                v = torch.zeros(shape)
            elif "conv_scale" in key and key.endswith("weight"):
                v = torch.randn(shape, generator=g) * (3.0 / max(1, t[0].numel()) ** 0.5)
            elif "conv_scale" in key and key.endswith("bias"):
                v = torch.ones(shape)
            elif t.dim() == 4:                       # conv weight: variance preserving
                fan_in = t.shape[1] * t.shape[2] * t.shape[3]
                v = torch.randn(shape, generator=g) * (1.2 / fan_in) ** 0.5
            elif key.endswith("weight"):             # BN gamma
                v = torch.rand(shape, generator=g) + 0.5
            elif key in ("hm.6.bias", "hm.quant_conv.bias", "hm.bias"):   # CenterNet focal-loss prior
                v = torch.full(shape, -2.19)
            else:                                    # BN beta / conv bias
                v = torch.randn(shape, generator=g) * 0.1
            t.copy_(v.to(t.device))
    return model


# ---- decode (lib/models/decode.py, lib/models/utils.py) ---------------------------------------

def _nms(heat, kernel=3):
    hmax = F.max_pool2d(heat, (kernel, kernel), stride=1, padding=(kernel - 1) // 2)
    return heat * (hmax == heat).float()


def _gather_feat(feat, ind):
    return feat.gather(1, ind.unsqueeze(2).expand(ind.size(0), ind.size(1), feat.size(2)))


def _transpose_and_gather_feat(feat, ind):
    feat = feat.permute(0, 2, 3, 1).contiguous()
    return _gather_feat(feat.view(feat.size(0), -1, feat.size(3)), ind)


def _topk(scores, K=40):
    batch, cat, height, width = scores.size()
    topk_scores, topk_inds = torch.topk(scores.view(batch, cat, -1), K)
    topk_inds = topk_inds % (height * width)
    topk_ys = torch.div(topk_inds, width, rounding_mode="floor").float()
    topk_xs = (topk_inds % width).float()
    topk_score, topk_ind = torch.topk(topk_scores.view(batch, -1), K)
    topk_clses = torch.div(topk_ind, K, rounding_mode="floor").int()
    topk_inds = _gather_feat(topk_inds.view(batch, -1, 1), topk_ind).view(batch, K)
    topk_ys = _gather_feat(topk_ys.view(batch, -1, 1), topk_ind).view(batch, K)
    topk_xs = _gather_feat(topk_xs.view(batch, -1, 1), topk_ind).view(batch, K)
    return topk_score, topk_inds, topk_clses, topk_ys, topk_xs


def ctdet_decode(heat, wh, reg=None, cat_spec_wh=False, K=100):
    """3x3 max-pool peak filter -> two-level top-K -> boxes [x1,y1,x2,y2,score,class]."""
    batch, cat, height, width = heat.size()
    heat = _nms(heat)
    scores, inds, clses, ys, xs = _topk(heat, K=K)
    if reg is not None:
        reg = _transpose_and_gather_feat(reg, inds).view(batch, K, 2)
        xs = xs.view(batch, K, 1) + reg[:, :, 0:1]
        ys = ys.view(batch, K, 1) + reg[:, :, 1:2]
    else:
        xs = xs.view(batch, K, 1) + 0.5
        ys = ys.view(batch, K, 1) + 0.5
    wh = _transpose_and_gather_feat(wh, inds)
    if cat_spec_wh:
        wh = wh.view(batch, K, cat, 2)
        wh = wh.gather(2, clses.view(batch, K, 1, 1).expand(batch, K, 1, 2).long()).view(batch, K, 2)
    else:
        wh = wh.view(batch, K, 2)
    clses = clses.view(batch, K, 1).float()
    scores = scores.view(batch, K, 1)
    bboxes = torch.cat([xs - wh[..., 0:1] / 2, ys - wh[..., 1:2] / 2,
                        xs + wh[..., 0:1] / 2, ys + wh[..., 1:2] / 2], dim=2)
    return torch.cat([bboxes, scores, clses], dim=2)


class ProcessBuffers:
    """Static buffers of the native post-processing -- the sigmoid copy of `hm` and the decode workspace -- keyed by
    (device, shape / size, stream).  OWNED: one instance lives on the model (`process`) or in the replay closure of a
    captured graph (`capture_process`, which runs on its own side stream), so the buffers die with their owner and
    two models / graphs of the same shape never share an `hm` a caller still holds."""

    def __init__(self):
        self.sigmoid = {}
        self.decode_ws = {}

    def sigmoid_buffer(self, hm, tag="hm"):
        # `tag` keeps the merged hm and wh of a flip test apart: a 2-class model gives both the shape [1,2,H,W]
        key = (tag, hm.device, tuple(hm.shape), torch.cuda.current_stream(hm.device).cuda_stream)
        if key not in self.sigmoid:
            self.sigmoid[key] = torch.empty_like(hm)
        return self.sigmoid[key]

    def workspace(self, device, need, stream):
        # its histograms must be zero at entry and are left zero by every call, so concurrent calls on
        # different streams must not share one
        key = (device, need, stream.cuda_stream)
        if key not in self.decode_ws:
            self.decode_ws[key] = torch.zeros(need // 4 + 64, dtype=torch.int32, device=device)
        return self.decode_ws[key]


class _BoundedBuffers(ProcessBuffers):
    """Owner-less calls of ctdet_decode_native: at most `cap` workspaces are kept (oldest dropped first).  A captured
    graph keeps the raw workspace pointer, so an owner-less call under stream capture is refused: capture with an
    owner (`bufs=ProcessBuffers()` kept alive next to the graph; capture_process does that)."""
    cap = 4

    def workspace(self, device, need, stream):
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("ctdet_decode_native under graph capture needs bufs=ProcessBuffers() owned by the "
                               "capture (the graph keeps the workspace pointer; the shared default cache evicts)")
        ws = super().workspace(device, need, stream)
        while len(self.decode_ws) > self.cap:
            self.decode_ws.pop(next(iter(self.decode_ws)))
        return ws


_default_bufs = _BoundedBuffers()


def ctdet_decode_native(heat, wh, reg=None, cat_spec_wh=False, K=100, apply_sigmoid=False,
                        heat_out=None, bufs=None):
    """ctdet_decode on the HIP kernels (codenet_decode.hip, cdn_ctdet_decode): same arguments and result
    as ``ctdet_decode``; apply_sigmoid=True takes logits (heat_out, if given, receives the sigmoid).
    Equal scores are ordered by ascending flat index.  GPU float32 tensors only."""
    from . import _native as N_
    if not (heat.is_cuda and heat.dtype == torch.float32):
        raise NotImplementedError("ctdet_decode_native needs float32 GPU tensors")
    heat, wh = heat.contiguous(), wh.contiguous()
    reg = reg.contiguous() if reg is not None else None
    B, cat, H, W = heat.shape
    lib = N_.lib()
    need = lib.cdn_ctdet_decode_workspace_bytes(B, cat, H, W)
    # one workspace per (owner, device, size, stream); `bufs` = the owner's ProcessBuffers (a captured graph keeps
    # the raw pointer, so its owner must outlive the graph: capture_process keeps it in the replay closure)
    stream = torch.cuda.current_stream(heat.device)
    ws = (bufs if bufs is not None else _default_bufs).workspace(heat.device, need, stream)
    ws_ptr = (ws.data_ptr() + 255) // 256 * 256
    dets = torch.empty(B, K, 6, device=heat.device)
    rc = lib.cdn_ctdet_decode(heat.data_ptr(), wh.data_ptr(), reg.data_ptr() if reg is not None else None,
                              B, cat, H, W, int(bool(cat_spec_wh)), K, int(bool(apply_sigmoid)),
                              heat_out.data_ptr() if heat_out is not None else None, dets.data_ptr(),
                              ws_ptr, need, stream.cuda_stream)
    N_.check(rc, "cdn_ctdet_decode")
    return dets


def process(model, images, flip_test=True, reg_offset=True, cat_spec_wh=False, K=100, native_decode=None,
            bufs=None):
    """CtdetDetector.process (lib/detectors/ctdet.py:29-46): images [2,3,R,R] = image + its W-flip
    when flip_test.  Returns (output dict, dets [B,K,6]).  native_decode (default: on GPU tensors) runs
    the peak filter / top-K / box assembly on the HIP kernels, without flip_test fused with the sigmoid."""
    if bufs is None and isinstance(model, nn.Module):
        bufs = model.__dict__.setdefault("_process_bufs", ProcessBuffers())      # owned by (and freed with) the model
    with torch.no_grad():
        output = model(images)[-1]
        if native_decode is None:
            native_decode = output["hm"].is_cuda
        reg = output["reg"] if reg_offset else None
        if native_decode and not flip_test:
            hm = output["hm"]
            # the reference's in-place hm.sigmoid_() (ctdet.py:32): the sigmoid goes to a second static buffer
            # that replaces output["hm"] (in place it would cost a second kernel, see cdn_ctdet_decode)
            # (a model on the fused path returns static buffers anyway; the module-by-module path returns fresh
            # tensors per call, so its sigmoid is a fresh tensor too)
            static = bufs is not None and getattr(model, "_fused", False)
            sig = bufs.sigmoid_buffer(hm) if static else torch.empty_like(hm)
            dets = ctdet_decode_native(hm, output["wh"], reg=reg, cat_spec_wh=cat_spec_wh, K=K,
                                       apply_sigmoid=True, heat_out=sig, bufs=bufs)
            output = dict(output)
            output["hm"] = sig
            return output, dets
        if native_decode and flip_test and output["hm"].shape[0] == 2 and not cat_spec_wh:
            # the README's test commands (--flip_test): sigmoid + mirror merge in one launch into static buffers, then
            # the native decode (capturable as a graph: capture_process(flip_test=True))
            from . import _native as N_
            hm2, wh2 = output["hm"].contiguous(), output["wh"].contiguous()
            static = bufs is not None and getattr(model, "_fused", False)
            hm = bufs.sigmoid_buffer(hm2[0:1], "hm") if static else torch.empty_like(hm2[0:1])
            wh = bufs.sigmoid_buffer(wh2[0:1], "wh") if static else torch.empty_like(wh2[0:1])
            rc = N_.lib().cdn_ctdet_flip_merge(hm2.data_ptr(), wh2.data_ptr(), 1, hm2.shape[1], wh2.shape[1], hm2.shape[2],
                                               hm2.shape[3], hm.data_ptr(), wh.data_ptr(),
                                               torch.cuda.current_stream(hm2.device).cuda_stream)
            N_.check(rc, "cdn_ctdet_flip_merge")
            dets = ctdet_decode_native(hm, wh, reg=reg[0:1] if reg is not None else None, cat_spec_wh=False, K=K,
                                       bufs=bufs)
            if hm2.data_ptr() != output["hm"].data_ptr():      # (contiguous() copied: hand the sigmoid back)
                output = dict(output)
                output["hm"] = hm2
            return output, dets         # output["hm"] holds the sigmoid of both images, as after the reference's sigmoid_()
        hm = output["hm"].sigmoid_()
        wh = output["wh"]
        if flip_test:
            hm = (hm[0:1] + torch.flip(hm[1:2], [3])) / 2
            wh = (wh[0:1] + torch.flip(wh[1:2], [3])) / 2
            reg = reg[0:1] if reg is not None else None
        if native_decode:
            dets = ctdet_decode_native(hm, wh, reg=reg, cat_spec_wh=cat_spec_wh, K=K, bufs=bufs)
        else:
            dets = ctdet_decode(hm, wh, reg=reg, cat_spec_wh=cat_spec_wh, K=K)
    return output, dets


def capture_process(model, images, reg_offset=True, cat_spec_wh=False, K=100, flip_test=False):
    """Capture ``process(model, images, flip_test)`` -- the whole fused network + native decode (flip_test: images =
    [image, its W-mirror] and the native sigmoid + mirror merge in between, the README's test procedure) -- over the
    static `images` buffer into one HIP graph (about 110 kernel launches; at small batches the Python /
    launch overhead of issuing them one by one dominates).  Returns replay() -> (output dict, dets); copy
    new images into `images` before each replay.  Needs model.enable_fused() and a GPU tensor."""
    assert getattr(model, "_fused", False) and images.is_cuda
    bufs = ProcessBuffers()          # owned by this capture: kept alive by replay(), released with it
    kw = dict(flip_test=bool(flip_test), reg_offset=reg_offset, cat_spec_wh=cat_spec_wh, K=K, bufs=bufs)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                 # warm-up: buffers, cached weights, kernel attributes
        for _ in range(3):
            process(model, images, **kw)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    # capture ON the warm-up stream: the static buffers are keyed by stream, so a capture stream of its own allocated
    # them again INSIDE the capture -- and the zero fill of the 84-MB decode workspace became a graph node, replayed
    # with every batch (12 us + a boundary in the kernel trace of round 4)
    with torch.cuda.graph(graph, stream=side):
        result = process(model, images, **kw)

    def replay():
        graph.replay()
        return result
    replay.buffers = bufs
    return replay


def create_model(heads=None, head_conv=64, w2=False, maxpool=False, quantize=False, seed=317,
                 w_bit=4, a_bit=8, wt_percentile=False, act_percentile=False):
    """PoseShuffleNetV2 with synthetic weights, optionally rewritten to W4A8 like
    lib/detectors/base_detector.py:28-34 does."""
    from .portable_quantizer import quantize_shufflenetv2_dcn
    heads = heads or {"hm": 20, "wh": 2, "reg": 2}
    model = PoseShuffleNetV2(heads, head_conv, w2=w2, maxpool=maxpool)
    fill_state_dict_(model, seed)
    if quantize:
        quantize_shufflenetv2_dcn(model, w_bit, None, a_bit, "symmetric", "asymmetric", True,
                                  wt_percentile, act_percentile, False, w2=w2, maxpool=maxpool)
    return model.eval()
