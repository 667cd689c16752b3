"""FusedHeads: the three detection heads on the stage kernels at half resolution (running ranges and byte codes).

Part of codenet_amd.pipeline (split by concern in round 6; `from codenet_amd import pipeline` exposes every name as
before)."""
import os

import torch
import torch.nn as nn

from .common import (OverflowFlags, act_fusable, bn_affine, uniform_act_settings)


class FusedHeads:
    """The detection heads on the stage kernels, fed by
    ``FusedHotPath.forward_nhwc``: per head
        1x1 conv (+BN) -> ReLU [-> QuantAct] -> depthwise 3x3 (+BN) -> ReLU [-> QuantAct] -> 1x1 conv + bias
    (fp32: shufflenetv2_dcn.py:244-271 nn.Sequential; W4A8: QuantDepthwiseNode, quant_modules.py:1013-1071).
    The first 1x1 conv runs at HALF resolution (a 1x1 conv commutes with the nearest up-sampling and the
    QuantAct extremes of a replicated tensor are those of the original), the depthwise kernel up-samples
    by addressing, so the up-sampled 64-channel tensor is never built.  Same parameters and QuantAct
    buffers (updated in place) as calling the head modules on the unpacked tensor."""

    def __init__(self, heads, int8_pointwise=True, small_tail=True, streams=True, fuse_first=None):
        self.heads = dict(heads)
        self.int8_pointwise = int8_pointwise
        # round 6: the heads' first 1x1 convs (64 -> 64 each, one shared input) as ONE launch (cdn_codenet_heads_pointwise_
        # forward): the input is read from HBM once instead of once per head; bit-identical to the per-head launches.
        # NOT the default: measured on one box, three interleaved pairs -- the heads alone 0.360 -> 0.337 ms (-6 %), but
        # the whole network 2.548 -> 2.570 ms (+0.9 %): inside the network the stage output the heads read was written a
        # moment ago and still sits in the 256-MB Infinity Cache, and three launches on three streams overlap each head's
        # range pass / tail with the other heads' first convs.  CDN_HEADS_FUSE_FIRST=1 (or fuse_first=True) turns it on.
        if fuse_first is None:
            fuse_first = os.environ.get("CDN_HEADS_FUSE_FIRST", "0") == "1"
        self.fuse_first = bool(fuse_first)
        self._first = None            # (cache key, concatenated int8 forms of the first convs)
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
        from ..portable_quantizer.quant_modules import QuantDepthwiseNode
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
        from ..portable_quantizer.quant_modules import QuantDepthwiseNode
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

    def _first_convs(self, params, dev):
        """The heads' first convs concatenated along the output rows (cached per weight version): dict of tensors, or None
        when the heads do not have the common form (64 -> 64 int8 pointwise + ReLU + plain QuantAct each)."""
        firsts = [layers[0] for layers in params.values()]
        if not (2 <= len(firsts) <= 4) or any(len(layers) != 3 for layers in params.values()):
            return None
        C = firsts[0]["w"].shape[1]
        for l_ in firsts:
            if (l_["i8"] is None or l_["ep"] is not None or not l_["relu"] or l_["act"] is None
                    or tuple(l_["w"].shape) != (64, C) or not act_fusable(l_["act"]) or l_["i8"][0].shape[0] != 64):
                return None
        key = tuple((t.data_ptr(), t._version) for l_ in firsts for t in (l_["w"], l_["i8"][0]))
        if self._first is None or self._first[0] != key:
            nh = len(firsts)
            cat = dict(
                w=torch.cat([l_["w"] for l_ in firsts], 0).contiguous(),
                codes=torch.cat([l_["i8"][0] for l_ in firsts], 0).contiguous(),
                scale=torch.cat([l_["i8"][1].reshape(-1) for l_ in firsts], 0).contiguous(),
                colsum=torch.cat([l_["i8"][2].reshape(-1) for l_ in firsts], 0).contiguous(),
                bias=(torch.cat([l_["bias"].reshape(-1) for l_ in firsts], 0).contiguous()
                      if all(l_["bias"] is not None for l_ in firsts) else None),
                omap=(torch.arange(64 * nh, dtype=torch.int32, device=dev) % 64).contiguous())
            if cat["bias"] is None and any(l_["bias"] is not None for l_ in firsts):
                return None
            self._first = (key, cat)
        return self._first[1]

    def _alloc(self, r, shape):
        from .. import _native as N_
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
        from ..portable_quantizer.quant_modules import QuantDepthwiseNode
        if isinstance(mod, QuantDepthwiseNode):
            return mod.quant_conv.out_channels
        if isinstance(mod, nn.Conv2d):
            return mod.out_channels
        return list(mod)[-1].out_channels

    def __call__(self, r, r_qstate, shape):
        from .. import _native as N_
        from .. import ops
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
        # round 6: the three first convs as one launch on the main stream, in front of the fork (the side streams then wait
        # for it: each head's range pass + tail follow on its own stream)
        fused_first = False
        first = (self._first_convs(params, r.device)
                 if (self.fuse_first and use_streams and B.get("y1_side") is not None
                     and len(B["y1_side"]) >= len(self.heads) - 1) else None)
        if first is not None:
            acts = [layers_[0]["act"] for layers_ in params.values()]
            try:
                bits, mom, running = uniform_act_settings(acts, "FusedHeads first convs")
            except NotImplementedError:
                first = None
        if first is not None and lib.cdn_codenet_heads_pointwise_supported(M, C, len(self.heads)):
            import ctypes
            nh = len(self.heads)
            if B.get("y1_all") is None:
                B["y1_all"] = torch.empty(nh, M, 64, device=r.device)
            wss = [B["ws"]] + list(B["ws_side"][:nh - 1])
            wptr = [(w_.data_ptr() + 255) // 256 * 256 for w_ in wss]
            wbytes = min((w_.numel() * 4 - (p_ - w_.data_ptr())) // 256 * 256 for w_, p_ in zip(wss, wptr))
            P = ctypes.c_void_p * nh
            rec = ops._tic("head_pw", (C, 64 * nh, M))
            rc = lib.cdn_codenet_heads_pointwise_forward(
                r.data_ptr(), r_qstate, M, C, nh, ptr(first["w"]), ptr(first["codes"]), ptr(first["scale"]),
                ptr(first["colsum"]), ptr(first["bias"]), 1, P(*[a_.x_min.data_ptr() for a_ in acts]),
                P(*[a_.x_max.data_ptr() for a_ in acts]), P(*[a_._device_state(r.device).data_ptr() for a_ in acts]),
                bits, mom, running, P(*wptr), wbytes, ptr(first["omap"]), B["y1_all"].data_ptr(), M * 64, stream)
            ops._toc(rec)
            N_.check(rc, "cdn_codenet_heads_pointwise_forward")
            fused_first = True
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
                if fused_first:
                    y1buf = B["y1_all"][hi]              # (written by the one launch in front of the fork)
                else:
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
        from ..portable_quantizer.quant_modules import QuantDepthwiseNode
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
        from .. import _native as N_
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
        cov, self.params_covered = getattr(self, "params_covered", None), None
        if not (cov is not None and set(id(a) for a in acts) <= set(cov[0])):      # (else: FrozenBackbone's first launch)
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
