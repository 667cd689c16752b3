"""FusedBackbone: stem, ShuffleNetV2 units and layer4 on the HIP kernels (running ranges, shuffle-free slots).

Part of codenet_amd.pipeline (split by concern in round 6; `from codenet_amd import pipeline` exposes every name as
before)."""
import os

import torch
import torch.nn as nn

from .common import (act_fusable)


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
        from ..portable_quantizer.quant_modules import QuantAct, QuantBaseNode, QuantBnConv2d
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
        from .. import _native as N_
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
        from .. import _native as N_
        w, b = self._folded(convbn)
        rc = N_.lib().cdn_codenet_dw3x3_nhwc_forward(
            a.data_ptr(), a_q, N, C, H, W, 0, stride, ld_in, ld_out, w.reshape(C, 9).data_ptr(), b.data_ptr(),
            None, None, 0, *self._act_args(act, out.device), self._ws_ptr, self._ws_bytes, out.data_ptr(),
            self._stream)
        N_.check(rc, "cdn_codenet_dw3x3_nhwc_forward")

    def _il(self, srcA, ldA, qA, srcB, ldB, qB, M, h, dst, ld_dst):
        from .. import _native as N_
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
        from .. import _native as N_
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
        from .. import _native as N_
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

    def _pw_raw(self, a_ptr, a_q, a_gen, M, lda, Wt, relu, act, state_ptr, out_map, out_ptr, ldo, n_gens=0):
        """n_gens: the number of QuantAct states behind a_q when a_gen names them (0: unknown) -- the streaming kernel then
        loads them all in the round trip of the generation bytes (cdn_codenet_pointwise_mixed_forward_n)."""
        from .. import _native as N_
        aa = self._act_args(act, self._dev)
        if state_ptr is not None:
            aa[2] = state_ptr
        rc = N_.lib().cdn_codenet_pointwise_mixed_forward_n(
            a_ptr, a_q, a_gen, n_gens if (a_gen and self.preload_states) else 0, M, Wt["K"], Wt["Co"], lda, ldo, Wt["w"].data_ptr(), Wt["codes"].data_ptr(),
            Wt["scale"].data_ptr(), Wt["colsum"].data_ptr(), Wt["bias"].data_ptr(), None, None, int(relu),
            out_map, *aa, self._ws_ptr, self._ws_bytes, out_ptr, self._stream)
        N_.check(rc, "cdn_codenet_pointwise_mixed_forward")

    recompute_pw1 = True      # A/B switch (tools/e2e_native_bench.py --no-recompute)
    preload_states = True     # A/B switch (--no-preload-states): n_gens handed to the mixed-generation pointwise
    range_first = True        # A/B switch: the recomputed conv's range pass before branch 1 is forked

    @staticmethod
    def _lib():
        from .. import _native as N_
        return N_.lib()

    def _pwdw_raw(self, x_ptr, x_q, N, cin, H, W, ld_x, Wt, act_mid, C, w, b, act_out, out, ld_out, apply_only=False):
        """1x1 conv (range-only pass -> act_mid) recomputed inside the stride-2 depthwise (-> out, range of act_out).
        apply_only: the range-only pass has been issued by the caller (`_pw_raw` with out_ptr None)."""
        from .. import _native as N_
        am, ao = self._act_args(act_mid, self._dev), self._act_args(act_out, out.device)
        fn = N_.lib().cdn_codenet_pwdw_s2_apply if apply_only else N_.lib().cdn_codenet_pwdw_s2_forward
        rc = fn(
            x_ptr, x_q, N, cin, H, W, ld_x, Wt["w"].data_ptr(), Wt["codes"].data_ptr(), Wt["scale"].data_ptr(),
            Wt["colsum"].data_ptr(), Wt["bias"].data_ptr(), am[0], am[1], am[2], C, w.data_ptr(), b.data_ptr(), ld_out,
            ao[0], ao[1], ao[2], ao[3], ao[4], ao[5], self._ws_ptr, self._ws_bytes, out.data_ptr(), self._stream)
        N_.check(rc, "cdn_codenet_pwdw_s2_forward")

    def _dw_raw(self, a_ptr, a_q, a_gen, N, C, H, W, stride, ld_in, w, b, act, out, ld_out):
        from .. import _native as N_
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
                                     self._t1s2(L).data_ptr(), ldh, n_gens=x_in["states"].numel() // 8 if mixed_in else 0)
                        self._dw_raw(L["t1s2"].data_ptr(), qptr(u["a1"]), None, Nb, h, L["Hin"], L["Win"], 2, ldh,
                                     P["w2"], P["b2"], u["a2"], L["t2"], ldh)
                    if ev is not None:
                        torch.cuda.current_stream(dev).wait_event(ev)
                else:
                    self._pw_raw(Y.data_ptr(), S.data_ptr(), P["gen_in"].data_ptr(), Mo, C, P["c1"], True,
                                 u["a1"], None, None, L["t1"].data_ptr(), ldh, n_gens=plan["ngen"])
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
        from .. import _native as N_
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
                             True, act4, None, None, B["out"].data_ptr(), c4, n_gens=lay["states"].numel() // 8)
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
