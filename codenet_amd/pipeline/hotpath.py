"""FusedHotPath: the three deform stages as one C call per stage (running-range default; fp32 chained form).

Part of codenet_amd.pipeline (split by concern in round 6; `from codenet_amd import pipeline` exposes every name as
before)."""
import os

import torch
import torch.nn as nn

from ..modules.dcn_deform_conv import DeformConvWithOffsetScaleBoundPositive
from .common import (ACT_PERCENTILE, DEFER_RANGE, PHASE_GATHER, PHASE_POINTWISE, PHASE_SCALE, bn_affine, global_range_active, stage_int8_codes, uniform_act_settings)


class FusedHotPath:
    """Runs a ``deconv_layers`` Sequential (fp32 or W4A8, built from the modules of this package)
    as the fused per-stage kernel schedule of codenet_fused.hip: one C-ABI call per stage, then one
    unpack (fake-quant + nearest x2 + NCHW) for the consumer.  Same parameters, same QuantAct
    buffers (updated in place), same results as calling the Sequential module by module.

    All device buffers are allocated once per input shape, so a call issues only kernel launches
    and can be captured into a HIP graph (``capture()``)."""

    def __init__(self, deconv_layers, int8_pointwise=True, kblocked_codes=True, chain_scale=True):
        from ..portable_quantizer.quant_modules import QuantDeformConvWithOffsetScaleBoundPositive
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
        from .. import _native as N_
        from ..portable_quantizer.quant_modules import QuantAct, QuantDeformConvWithOffsetScaleBoundPositive
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
        from .. import _native as N_
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
        from .. import _native as N_
        from .. import ops
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
        from .. import _native as N_
        from .. import ops
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
        from .. import _native as N_
        st = act._device_state(dev)
        f = st.view(torch.float32)
        from ..portable_quantizer.quant_modules import allreduce_extremes
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
