"""The CoDeNet stage as ONE autograd function on the HIP kernels (training path, config e).

The reference composes ``DeformConvWithOffsetScaleBoundPositive`` / its W4A8 counterpart from framework ops
under autograd (modules/dcn_deform_conv.py:323-330; quant_modules.py:668-671): conv_scale -> Hardtanh
[-> QuantAct] -> offsets -> deform_conv (native forward / backward, dcn_deform_conv_cuda.cpp:151-484)
[-> QuantAct] -> conv_channel.  Here forward and backward of the whole chain are native:

    forward   cdn_codenet_scale_forward, cdn_quantact_forward, cdn_codenet_dw_forward, cdn_quantact_forward,
              cdn_codenet_pointwise_forward
    backward  cdn_codenet_pointwise_forward on the transposed weights (data gradient), cdn_codenet_pointwise_wgrad
              (weight + bias gradient, f32 MFMA), cdn_codenet_dw_backward (grad_x, grad_s, grad_w_dw),
              cdn_codenet_scale_backward (grad_x +=, grad_w_scale)

Straight-through estimators exactly as the reference's quantisers define them (quant_utils.py:202-204,
227-229: the QuantAct backward is the identity); the Hardtanh passes gradients where lo < s_raw < hi.
The WEIGHT transformations (per-channel fake-quantisation, BN fold) are tiny tensors and stay torch ops in the
calling module, so their gradients (straight-through, chain rule through the fold) come from autograd.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _native as N_
from .. import ops


def _p(t):
    return t.data_ptr() if t is not None else None


def _workspace(nbytes, device):
    """Scratch of one native call (split-K partials, per-workgroup partial sums), owned by that call: a fresh tensor from
    the caching allocator -- under stream capture from the capturing graph's private pool, so a graph owns every address
    it recorded.  (Rounds 4-5 kept one buffer per (device, stream) in a module-global dict, grown by re-allocation: entries
    leaked with every side stream, a recycled stream handle found a stale buffer, and two captures on the shared capture
    stream shared one buffer that lived in the first graph's pool -- ADVICE r5.)  An allocation from the cache costs
    about a microsecond of host time; stream-ordered reuse by the allocator gives the same memory back every step."""
    return torch.empty(nbytes // 4 + 64, dtype=torch.float32, device=device)


# A/B switch (tools/train_step_bench.py --no-bf16-wgrad): exact bf16 products on the integer form of d
# (pw_wgrad_q3_kernel) where the tiles are whole, or the f32-MFMA kernel everywhere
WGRAD_BF16X3 = True


def pointwise_wgrad(grad_y, d, want_bias, d_state=None):
    """(grad_w [Co, C], grad_b [Co] or None) of y = conv1x1(d, w) + b; grad_y [N,Co,H,W], d [N,C,H,W].  d_state: d holds
    pre-quantisation values, fake-quantised on load with that (snapshot of a) QuantAct state."""
    Nb, Co, H, W = grad_y.shape
    C = d.shape[1]
    lib = N_.lib()
    need = lib.cdn_codenet_pointwise_wgrad_workspace_bytes(Nb, C, Co, H * W)
    ws = _workspace(need, grad_y.device)
    gw = torch.empty(Co, C, device=grad_y.device)
    gb = torch.empty(Co, device=grad_y.device) if want_bias else None
    if d_state is not None:
        fn = lib.cdn_codenet_pointwise_wgrad_q if WGRAD_BF16X3 else lib.cdn_codenet_pointwise_wgrad_q_f32
        rc = fn(_p(grad_y), _p(d), _p(d_state), _p(gw), _p(gb), Nb, C, Co, H * W, _p(ws), ws.numel() * 4,
                ops._stream(grad_y))
        N_.check(rc, "cdn_codenet_pointwise_wgrad_q")
        return gw, gb
    rc = lib.cdn_codenet_pointwise_wgrad(_p(grad_y), _p(d), _p(gw), _p(gb), Nb, C, Co, H * W, _p(ws),
                                         ws.numel() * 4, ops._stream(grad_y))
    N_.check(rc, "cdn_codenet_pointwise_wgrad")
    return gw, gb


# Reproducible gather backward (round 5; tools/train_step_bench.py --atomic-dw-bwd for the A/B): grad_s / grad_w_dw as
# per-workgroup partials reduced in a fixed order (cdn_codenet_dw_backward_r) instead of float atomics -- two identical
# runs of a QAT step then give bit-identical parameters.
REPRODUCIBLE_DW_BWD = True

# A/B switch (tools/train_step_bench.py --no-fuse-dq): d fake-quantised by its consumers while loading (True) or stored
# fake-quantised by a pass of its own (False).  Same values either way.
FUSE_DQ_ON_LOAD = True


class CodenetStageFunction(Function):
    """y = conv1x1(QA_d(dw_deform(x, QA_s(hardtanh(conv1x1(x, w_scale) + b_scale)), w_dw)), w_pw) + b_pw.
    act_s / act_d: QuantAct modules (asymmetric, plain min/max) or None; their buffers are updated in place
    by the forward exactly as in eval mode (the reference's QuantAct does not distinguish, quant_modules.py:
    203-219)."""

    @staticmethod
    def forward(ctx, x, w_scale, b_scale, w_dw, w_pw, b_pw, lo, hi, act_s, act_d, want_range=False, x_up=False,
                pw_int8=False):
        """x_up (round 4): x is the STORED tensor [N,C,H/2,W/2] whose nearest x2 up-sampling is the stage's input (stages
        1-2: shufflenetv2_dcn.py:303-308).  The scale is predicted at stored resolution (a 1x1 conv of a replicated tensor
        is the replicated conv; min / max and hence the QuantAct range are those of the replicated plane), the gather reads
        the stored planes through the up-sampling (same values, bit-identical d), and the backward accumulates the
        gradient with respect to the stored tensor directly (cdn_codenet_dw_up2_backward: the four pixels of a block
        share their bilinear cells) -- the up-sampled tensor and its gradient are never written."""
        ops._gpu_f32(x, w_scale, b_scale, w_dw, w_pw, b_pw)
        x = x.contiguous()
        ctx.x_up = bool(x_up)
        ctx.set_materialize_grads(False)      # (no zero tensor for the non-differentiable partials output: a launch per stage)
        # every producer leaves the {min, max} pairs of its output for the QuantAct behind it: no range passes
        post_act = want_range if (want_range is not None and not isinstance(want_range, bool)) else None
        want_range = bool(want_range) or post_act is not None
        if act_s is not None and act_s.running_stat and ops.FUSE_RANGE_UPDATE:
            # round 6: the producer's last workgroup updates the QuantAct (no update launch); one apply launch for the plane
            s_c = ops.codenet_scale_update(x, w_scale, b_scale, lo, hi, act_s)
            s = ops.quantact_apply(s_c, act_s)
        elif act_s is not None and act_s.running_stat:
            s_c, sp = ops.codenet_scale(x, w_scale, b_scale, lo, hi, want_range=True)   # clamped, pre-quantisation
            s = ops.quantact_forward_partials(s_c, act_s, sp)
        else:
            s_c = ops.codenet_scale(x, w_scale, b_scale, lo, hi)
            s = _native_quantact(act_s, s_c) if act_s is not None else s_c
        have_pw = w_pw is not None
        d_snap = None          # snapshot of act_d's state when d stays un-quantised in memory (quantised by its consumers)
        with torch.no_grad():
            if (act_d is not None and act_d.running_stat and have_pw and FUSE_DQ_ON_LOAD
                    and ops.codenet_dw_update_supported(x, x_up)):
                # round 6: the gather's last workgroup updates the QuantAct and leaves the state snapshot itself
                d, d_snap = ops.codenet_dw_update(x, s, w_dw, act_d, up2=x_up)
                d_q = d
            elif act_d is not None and act_d.running_stat and have_pw and FUSE_DQ_ON_LOAD:
                # the gather leaves its {min, max} pairs, the QuantAct only updates, the pointwise kernel (and, in the
                # backward, the weight-gradient kernel) fake-quantise d while loading it: d_q is never stored
                d, dp = ops.codenet_dw_up2(x, s, w_dw) if x_up else ops.codenet_dw_range(x, s, w_dw)
                d_snap = ops.quantact_forward_partials(d, act_d, dp, want_out=False, want_state_copy=True)
                d_q = d
            elif act_d is not None and act_d.running_stat:
                d, dp = ops.codenet_dw_up2(x, s, w_dw) if x_up else ops.codenet_dw_range(x, s, w_dw)
                d_q = ops.quantact_forward_partials(d, act_d, dp)
            else:
                d = ops.codenet_dw_up2(x, s, w_dw, want_range=False) if x_up else ops.codenet_dw(x, s, w_dw.contiguous())
                d_q = _native_quantact(act_d, d) if act_d is not None else d
        yp = None
        keep = {}
        if have_pw and want_range:
            upd = ((post_act, True) if (post_act is not None and post_act.running_stat and native_act_ok(post_act)
                                        and not getattr(post_act, "global_range", False)) else None)
            y, yp = ops.codenet_pointwise(d_q, w_pw, b_pw, want_range=True, d_state=d_snap, int8_weights=pw_int8, keep=keep,
                                          update_act=upd)
            if keep.get("range_committed"):
                post_act._range_committed = True      # (ReluQuant / ReluQuantUpsample behind the stage: apply only)
        else:
            y = ops.codenet_pointwise(d_q, w_pw, b_pw, d_state=d_snap, int8_weights=pw_int8, keep=keep) if have_pw else d_q
        ctx.pw_fwd_ws = keep.get("fwd_ws")      # (the int8 forward's weight scales: the data gradient reads them)
        ctx.lo, ctx.hi, ctx.have_pw = float(lo), float(hi), have_pw
        ctx.has_b_scale, ctx.has_b_pw = b_scale is not None, b_pw is not None
        ctx.save_for_backward(x, s_c, s, w_scale, w_dw, d_q if have_pw else None, w_pw, d_snap)
        if want_range:
            if yp is None:
                yp = y.new_zeros(0, 2)
            ctx.mark_non_differentiable(yp)
            return y, yp
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, *unused):
        x, s_c, s, w_scale, w_dw, d_q, w_pw, d_snap = ctx.saved_tensors
        need = ctx.needs_input_grad
        if gy is None:                   # (set_materialize_grads(False): y took no part in the loss)
            return (None,) * 13
        gy = gy.contiguous()
        Nb, C, H, W = x.shape            # (x_up: the STORED resolution; the stage runs at 2H x 2W)
        lib = N_.lib()
        g_wpw = g_bpw = None
        if ctx.have_pw:
            Co = w_pw.shape[0]
            if need[4] or (need[5] and ctx.has_b_pw):
                gw2, g_bpw = pointwise_wgrad(gy, d_q, need[5] and ctx.has_b_pw, d_state=d_snap)
                g_wpw = gw2.view_as(w_pw) if need[4] else None
            # data gradient: exact products on bf16 MFMA where the forward ran on the weight codes, else the same
            # contraction as the forward with the transposed weights on f32 MFMA
            gd = ops.codenet_pointwise_dgrad_q4(gy, w_pw, ctx.pw_fwd_ws) if ctx.pw_fwd_ws is not None else None
            if gd is None:
                gd = ops.codenet_pointwise(gy, w_pw.reshape(Co, C).t().contiguous().view(C, Co, 1, 1))
        else:
            gd = gy
        # gather backward (QuantAct on d: straight-through)
        want_x, want_s, want_wdw = need[0], (need[1] or need[2] or need[0]), need[3]
        def gs_and_gw():
            # grad_s and grad_w_dw in one buffer: the library's fill of grad_s covers a grad_w that lies right behind it
            if want_s and want_wdw:
                buf = torch.empty(s.numel() + w_dw.numel(), device=x.device)
                return buf[:s.numel()].view_as(s), buf[s.numel():].view_as(w_dw)
            return (torch.empty_like(s) if want_s else None), (torch.zeros_like(w_dw) if want_wdw else None)

        def run(up2, Hf, Wf):
            gx = torch.empty_like(x) if want_x else None
            gs, g_wdw = gs_and_gw()
            args = (_p(x), _p(s), _p(w_dw.contiguous()), _p(gd), _p(gx), _p(gs), _p(g_wdw), Nb, C, Hf, Wf)
            nws = lib.cdn_codenet_dw_backward_workspace_bytes(Nb, C, Hf, Wf, int(up2)) if REPRODUCIBLE_DW_BWD else 0
            if nws:        # (0: no reproducible form for a plane beyond LDS -- the atomic one below)
                ws = _workspace(nws, x.device)
                fn = lib.cdn_codenet_dw_up2_backward_r if up2 else lib.cdn_codenet_dw_backward_r
                N_.check(fn(*args, _p(ws), ops._stream(x)), "cdn_codenet_dw_backward_r")
            else:
                fn = lib.cdn_codenet_dw_up2_backward if up2 else lib.cdn_codenet_dw_backward
                N_.check(fn(*args, ops._stream(x)), "cdn_codenet_dw_backward")
            return gx, gs, g_wdw

        if ctx.x_up:
            gx, gs, g_wdw = run(True, 2 * H, 2 * W)
        elif not lib.cdn_codenet_dw_backward_supported(H, W):
            gx, gs, g_wdw = ops._dw_backward_generic(x, s, w_dw, gd, (want_x, want_s, want_wdw))
            gx = gx.contiguous() if gx is not None else None
        else:
            gx, gs, g_wdw = run(False, H, W)
        g_wscale = g_bscale = None
        if want_s:
            # QuantAct on s: straight-through; Hardtanh: gradient where lo < s_raw < hi (s_c is the clamped
            # value: s_c == lo or hi <=> s_raw outside or on the bound, where torch's hardtanh_backward is 0 too)
            # -- folded into the kernel (cdn_codenet_scale_backward_masked); column C of the partials is the bias share
            want_w, want_b = need[1], need[2] and ctx.has_b_scale
            part = torch.empty(Nb, C + 1, device=x.device) if (want_w or want_b) else None
            rc = lib.cdn_codenet_scale_backward_masked(_p(x), _p(gs), _p(s_c), ctx.lo, ctx.hi,
                                                       _p(w_scale.contiguous().view(-1)), _p(gx), _p(part), Nb, C, H, W,
                                                       ops._stream(x))
            N_.check(rc, "cdn_codenet_scale_backward_masked")
            if part is not None:
                tot = part.sum(0)                               # over the images, one reduction for both
                if want_w:
                    g_wscale = tot[:C].view_as(w_scale)
                if want_b:
                    g_bscale = tot[C:].reshape(1)
        return gx, g_wscale, g_bscale, g_wdw, g_wpw, g_bpw, None, None, None, None, None, None, None


def _native_quantact(act, t):
    """QuantAct.forward on the device kernel: range tracking in place + fake-quantisation (no autograd)."""
    out, _ = ops.quantact_forward(t, act.x_min, act.x_max, act._device_state(t.device), bits=act.activation_bit,
                                  momentum=act.momentum, running=act.running_stat)
    return out


def codenet_stage(x, w_scale, b_scale, w_dw, w_pw, b_pw, lo, hi, act_s=None, act_d=None, want_range=False, x_up=False,
                  pw_int8=False):
    """want_range: returns (y, partials) -- the per-workgroup {min, max} pairs of y ([n, 2], empty without a pointwise
    conv) for a QuantAct behind the stage (ReluQuantUpsample).  x_up: see CodenetStageFunction.forward.  pw_int8: w_pw is
    a per-channel symmetric <= 4-bit fake-quantised weight (the forward conv_channel may sum integer codes)."""
    return CodenetStageFunction.apply(x, w_scale, b_scale, w_dw, w_pw, b_pw, lo, hi, act_s, act_d, want_range, x_up,
                                      pw_int8)


class QuantActSTE(Function):
    """QuantAct under autograd on the device kernel: forward = range tracking + fake-quantisation, backward =
    identity (AsymmetricQuantFunction.backward, quant_utils.py:202-204)."""

    @staticmethod
    def forward(ctx, x, act):
        return _native_quantact(act, x.contiguous())

    @staticmethod
    def backward(ctx, g):
        return g, None


class FakeQuantWeight(Function):
    """Per-output-channel symmetric weight fake-quantisation in one launch (cdn_codenet_weight_prep), straight-through
    backward (SymmetricQuantFunction.backward returns grad_output.clone(), quant_utils.py:227-229).  Bit-identical to
    the torch composition in portable_quantizer (min / max per channel, n / clamp(mag), round, clamp, true division)."""

    @staticmethod
    def forward(ctx, w, bits, percentile=False):
        w = w.contiguous()
        out = torch.empty_like(w)
        co = w.shape[0]
        k_lo, k_hi, shrink = weight_range_ranks(w.numel() // co, percentile)
        rc = N_.lib().cdn_codenet_weight_prep_ranked(_p(w), co, w.numel() // co, None, None, None, None, int(bits),
                                                     k_lo, k_hi, shrink, _p(out), None, ops._stream(w))
        N_.check(rc, "cdn_codenet_weight_prep_ranked")
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None, None    # (the reference clones; an identity backward needs no copy)


class MultiFakeQuantWeight(Function):
    """FakeQuantWeight for several weight tensors in ONE launch (cdn_codenet_weight_prep_multi): meta is a tuple of
    (bits, percentile) per tensor; returns the fake-quantised tensors (bit-identical to FakeQuantWeight on each);
    straight-through backward."""

    @staticmethod
    def forward(ctx, meta, *ws):
        import ctypes
        n = len(ws)
        ws = [w.contiguous() for w in ws]
        outs = [torch.empty_like(w) for w in ws]
        co = [w.shape[0] for w in ws]
        kk = [w.numel() // w.shape[0] for w in ws]
        ranks = [weight_range_ranks(k, bool(m[1])) for k, m in zip(kk, meta)]
        P, I64, I, F = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int * n, ctypes.c_float * n
        rc = N_.lib().cdn_codenet_weight_prep_multi(
            n, P(*[w.data_ptr() for w in ws]), I64(*co), I64(*kk), None, None, None, None, I(*[int(m[0]) for m in meta]),
            I(*[r[0] for r in ranks]), I(*[r[1] for r in ranks]), F(*[r[2] for r in ranks]),
            P(*[o.data_ptr() for o in outs]), None, ops._stream(ws[0]))
        N_.check(rc, "cdn_codenet_weight_prep_multi")
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        return (None,) + tuple(gs)


class MultiFoldFakeQuantWeight(Function):
    """FoldFakeQuantWeight for several QuantBnConv2d weights in ONE launch: flat = (w, conv_bias or None, gamma, beta,
    running_mean, running_var) per tensor, meta = (bits, percentile, eps) per tensor; returns (w_q_0, b_0, w_q_1, b_1, ...).
    The per-channel factors come from the framework's own add / sqrt / div (their foreach forms: one launch each for all
    tensors); backward per tensor as FoldFakeQuantWeight's."""

    @staticmethod
    def forward(ctx, meta, *flat):
        import ctypes
        n = len(meta)
        ws = [flat[6 * t].contiguous() for t in range(n)]
        cbs = [flat[6 * t + 1] for t in range(n)]
        gam = [flat[6 * t + 2] for t in range(n)]
        bet = [flat[6 * t + 3].contiguous() for t in range(n)]
        mea = [flat[6 * t + 4].contiguous() for t in range(n)]
        var = [flat[6 * t + 5] for t in range(n)]
        stds = torch._foreach_sqrt(torch._foreach_add(var, [float(m[2]) for m in meta]))
        sfs = [t_.contiguous() for t_ in torch._foreach_div(gam, stds)]
        outs = [torch.empty_like(w) for w in ws]
        bs = [torch.empty(w.shape[0], device=w.device) for w in ws]
        co = [w.shape[0] for w in ws]
        kk = [w.numel() // w.shape[0] for w in ws]
        ranks = [weight_range_ranks(k, bool(m[1])) for k, m in zip(kk, meta)]
        P, I64, I, F = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int * n, ctypes.c_float * n
        cbp = [c.contiguous() if c is not None else None for c in cbs]
        rc = N_.lib().cdn_codenet_weight_prep_multi(
            n, P(*[w.data_ptr() for w in ws]), I64(*co), I64(*kk), P(*[t_.data_ptr() for t_ in sfs]),
            P(*[t_.data_ptr() for t_ in bet]), P(*[t_.data_ptr() for t_ in mea]),
            P(*[c.data_ptr() if c is not None else None for c in cbp]), I(*[int(m[0]) for m in meta]),
            I(*[r[0] for r in ranks]), I(*[r[1] for r in ranks]), F(*[r[2] for r in ranks]),
            P(*[o.data_ptr() for o in outs]), P(*[b.data_ptr() for b in bs]), ops._stream(ws[0]))
        N_.check(rc, "cdn_codenet_weight_prep_multi")
        ctx.n = n
        ctx.has_cb = [c is not None for c in cbs]
        saved = []
        for t in range(n):
            saved += [ws[t], sfs[t], stds[t], mea[t]] + ([cbp[t]] if cbp[t] is not None else [])
        ctx.save_for_backward(*saved)
        res = []
        for t in range(n):
            res += [outs[t], bs[t]]
        return tuple(res)

    @staticmethod
    @once_differentiable
    def backward(ctx, *gs):
        saved = list(ctx.saved_tensors)
        need = ctx.needs_input_grad
        grads = [None]
        for t in range(ctx.n):
            w, sf, std, mean = saved[:4]
            saved = saved[4:]
            cb = None
            if ctx.has_cb[t]:
                cb, saved = saved[0], saved[1:]
            g_wq, g_b = gs[2 * t], gs[2 * t + 1]
            nd = need[1 + 6 * t: 7 + 6 * t]
            co = w.shape[0]
            if g_wq is None and g_b is None:
                grads += [None] * 6
                continue
            if g_wq is None:
                g_wq = torch.zeros_like(w)
            new = lambda *sh: torch.empty(*sh, device=w.device)      # noqa: E731
            g_w = torch.empty_like(w) if nd[0] else None
            g_gamma = new(co) if nd[2] else None
            g_beta = new(co) if nd[3] else None
            g_cb = new(co) if (cb is not None and nd[1]) else None
            rc = N_.lib().cdn_codenet_weight_prep_backward(
                _p(g_wq.contiguous()), _p(g_b.contiguous()) if g_b is not None else None, _p(w), _p(sf), _p(std), _p(mean),
                _p(cb), co, w.numel() // co, _p(g_w), _p(g_gamma), _p(g_beta), _p(g_cb), ops._stream(w))
            N_.check(rc, "cdn_codenet_weight_prep_backward")
            grads += [g_w, g_cb, g_gamma, g_beta, None, None]
        return tuple(grads)


def folded_weights_of_stages(stages):
    """(w_q, folded bias) of every stage's conv_channel + BN from one launch (plus three foreach launches for the
    per-channel factors), or None where a layer is not on the device weight-prep path."""
    cbs = [q.quant_conv_channel_bn for q in stages]
    if not MULTI_WEIGHT_PREP or not cbs or len(cbs) > 8:
        return None
    for cb in cbs:
        if cb.full_precision_flag or not native_weight_prep_ok(cb.conv.weight, cb):
            return None
    flat = []
    for cb in cbs:
        flat += [cb.conv.weight, cb.conv.bias, cb.bn.weight, cb.bn.bias, cb.bn.running_mean, cb.bn.running_var]
    outs = MultiFoldFakeQuantWeight.apply(tuple((cb.weight_bit, cb.weight_percentile, cb.bn.eps) for cb in cbs), *flat)
    return [(outs[2 * i], outs[2 * i + 1]) for i in range(len(cbs))]


# A/B switch (tools/train_step_bench.py --no-multi-prep)
MULTI_WEIGHT_PREP = True


def small_weights_of_stages(stages):
    """The fake-quantised conv_scale and depthwise weights of all `stages` from one launch: a list of (w_scale_q, w_dw_q)
    per stage, or None where a quantiser is not on the device weight-prep path (the modules then prepare their own)."""
    qs = []
    for q in stages:
        for m in (q.quant_conv_scale, q.quant_deform_conv):
            if m.full_precision_flag or not native_weight_prep_ok(m.weight, m):
                return None
            qs.append(m)
    if not MULTI_WEIGHT_PREP or not qs or len(qs) > 8:
        return None
    outs = MultiFakeQuantWeight.apply(tuple((m.weight_bit, m.weight_percentile) for m in qs), *[m.weight for m in qs])
    return [(outs[2 * i], outs[2 * i + 1]) for i in range(len(stages))]


class FoldFakeQuantWeight(Function):
    """QuantBnConv2d's weight path in one launch: BN fold from the running statistics, then per-channel symmetric
    fake-quantisation; returns (w_q, folded fp32 bias).  Backward = autograd of the reference's composition
    (quant_modules.py:365-372 under a straight-through quantiser): with sf = gamma / sqrt(var + eps),
    grad_w = g * sf, grad_gamma = (sum_k g * w + grad_b * (conv_bias - mean)) / sqrt(var + eps), grad_beta = grad_b,
    grad_conv_bias = grad_b * sf."""

    @staticmethod
    def forward(ctx, w, conv_bias, gamma, beta, mean, var, eps, bits, percentile=False):
        w = w.contiguous()
        co = w.shape[0]
        wq = torch.empty_like(w)
        b = torch.empty(co, device=w.device)
        # the per-channel factor with the framework's own sqrt / division (PyTorch-ROCm's sqrt is not the correctly
        # rounded one; the factor must be the framework's bit for bit): two tiny ops, everything else is the kernel
        std = torch.sqrt(var + eps)
        sf = (gamma / std).contiguous()
        k_lo, k_hi, shrink = weight_range_ranks(w.numel() // co, percentile)
        rc = N_.lib().cdn_codenet_weight_prep_ranked(
            _p(w), co, w.numel() // co, _p(sf), _p(beta.contiguous()), _p(mean.contiguous()),
            _p(conv_bias.contiguous()) if conv_bias is not None else None, int(bits), k_lo, k_hi, shrink, _p(wq), _p(b),
            ops._stream(w))
        N_.check(rc, "cdn_codenet_weight_prep_ranked")
        ctx.save_for_backward(w, sf, std, mean, conv_bias)
        ctx.has_cb = conv_bias is not None
        return wq, b

    @staticmethod
    @once_differentiable
    def backward(ctx, g_wq, g_b):
        w, sf, std, mean, conv_bias = ctx.saved_tensors
        need = ctx.needs_input_grad
        co = w.shape[0]
        new = lambda *sh: torch.empty(*sh, device=w.device)      # noqa: E731
        g_w = torch.empty_like(w) if need[0] else None
        g_gamma = new(co) if need[2] else None
        g_beta = new(co) if need[3] else None
        g_cb = new(co) if (ctx.has_cb and need[1]) else None
        rc = N_.lib().cdn_codenet_weight_prep_backward(
            _p(g_wq.contiguous()), _p(g_b.contiguous()) if g_b is not None else None, _p(w), _p(sf), _p(std),
            _p(mean.contiguous()), _p(conv_bias.contiguous()) if ctx.has_cb else None, co, w.numel() // co, _p(g_w),
            _p(g_gamma), _p(g_beta), _p(g_cb), ops._stream(w))
        N_.check(rc, "cdn_codenet_weight_prep_backward")
        return g_w, g_cb, g_gamma, g_beta, None, None, None, None, None


def weight_range_ranks(length, percentile):
    """(k_low, k_high, shrink) of cdn_codenet_weight_prep_ranked for a channel of `length` weights: plain min / max, or
    the reference's --wt-percentile rule (quant_modules.py:287-300; the expressions of portable_quantizer._channel_range)."""
    import math
    if not percentile:
        return 1, 1, 1.0
    if length < 10:
        return 1, 1, 0.95
    lo = math.ceil(length * 0.1 * 0.01)
    hi = math.ceil(length * 99.9 * 0.01)
    return lo, length - hi + 1, 1.0


def native_weight_prep_ok(w, quantizer):
    """Training-time weight transformation on the device kernel: per-channel symmetric fake-quantisation with plain
    min / max ranges or the --wt-percentile ranges of the README's QAT command (ranks up to 4: channels of up to 4000
    weights)."""
    if not (w.is_cuda and w.dtype == torch.float32 and torch.is_grad_enabled() and quantizer.per_channel
            and quantizer.quant_mode == "symmetric" and not quantizer.full_precision_flag
            and not quantizer.quantize_bias):
        return False
    k_lo, k_hi, _ = weight_range_ranks(w.numel() // w.shape[0], quantizer.weight_percentile)
    return k_lo <= 4 and k_hi <= 4


class ReluQuantUpsample(Function):
    """The block after every deform stage -- ReLU(inplace) -> QuantAct -> Upsample(x2, nearest)
    (lib/models/networks/shufflenetv2_dcn.py:303-308 after quantize_model.py:79-81) -- as one autograd function on two
    kernels: forward reads the pre-ReLU tensor once (range tracking on max(y, 0), then the fake-quantised values
    written straight to their 2x2 replicas); backward sums the replicas' gradients where y > 0 (straight-through
    QuantAct).  Same values as the three modules (tests/test_train_step.py)."""

    @staticmethod
    def forward(ctx, y, act, partials=None):
        ops._gpu_f32(y)
        y = y.contiguous()
        Nb, C, H, W = y.shape
        out = torch.empty(Nb, C, 2 * H, 2 * W, device=y.device)
        if getattr(act, "_range_committed", False):
            # round 6: the stage's pointwise kernel has already updated this QuantAct (its last workgroup): apply only
            act._range_committed = False
            rc = N_.lib().cdn_quantact_relu_apply(_p(y), _p(out), Nb * C, H, W, 1, _p(act._device_state(y.device)),
                                                  ops._stream(y))
            N_.check(rc, "cdn_quantact_relu_apply")
        elif partials is not None and partials.shape[0] > 0 and act.running_stat:
            # the stage's pointwise kernel left the {min, max} pairs of y: no range pass over y
            rc = N_.lib().cdn_quantact_relu_up2_forward_partials(
                _p(y), _p(out), Nb * C, H, W, _p(act.x_min), _p(act.x_max), _p(act._device_state(y.device)),
                _p(partials), partials.shape[0], int(act.activation_bit), float(act.momentum), 1, ops._stream(y))
            N_.check(rc, "cdn_quantact_relu_up2_forward_partials")
        else:
            rc = N_.lib().cdn_quantact_relu_up2_forward(_p(y), _p(out), Nb * C, H, W, _p(act.x_min), _p(act.x_max),
                                                        _p(act._device_state(y.device)), int(act.activation_bit),
                                                        float(act.momentum), int(bool(act.running_stat)),
                                                        ops._stream(y))
            N_.check(rc, "cdn_quantact_relu_up2_forward")
        ctx.save_for_backward(y)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = g.contiguous()
        Nb, C, H, W = y.shape
        gy = torch.empty_like(y)
        rc = N_.lib().cdn_up2_relu_backward(_p(g), _p(y), _p(gy), Nb * C, H, W, ops._stream(y))
        N_.check(rc, "cdn_up2_relu_backward")
        return gy, None, None


class ReluQuant(Function):
    """ReLU(inplace) -> QuantAct of the block behind a stage WITHOUT materialising the Upsample that follows (round 4):
    the next stage reads this stored tensor through the up-sampling (CodenetStageFunction x_up).  Same range tracking and
    fake-quantised values as ReluQuantUpsample; backward = grad where y > 0 (the 2x2 sum of the up-sampling backward is
    part of the next stage's gather backward)."""

    @staticmethod
    def forward(ctx, y, act, partials=None):
        ops._gpu_f32(y)
        y = y.contiguous()
        out = torch.empty_like(y)
        if getattr(act, "_range_committed", False):      # (see ReluQuantUpsample.forward)
            act._range_committed = False
            Nb, C, H, W = y.shape
            rc = N_.lib().cdn_quantact_relu_apply(_p(y), _p(out), Nb * C, H, W, 0, _p(act._device_state(y.device)),
                                                  ops._stream(y))
            N_.check(rc, "cdn_quantact_relu_apply")
            ctx.save_for_backward(y)
            return out
        use_p = partials is not None and partials.shape[0] > 0 and act.running_stat
        rc = N_.lib().cdn_quantact_relu_forward(
            _p(y), _p(out), y.numel(), _p(act.x_min), _p(act.x_max), _p(act._device_state(y.device)),
            _p(partials) if use_p else None, partials.shape[0] if use_p else 0, int(act.activation_bit),
            float(act.momentum), int(bool(act.running_stat)), ops._stream(y))
        N_.check(rc, "cdn_quantact_relu_forward")
        ctx.save_for_backward(y)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = g.contiguous()
        gy = torch.empty_like(y)
        rc = N_.lib().cdn_relu_backward(_p(g), _p(y), _p(gy), y.numel(), ops._stream(y))
        N_.check(rc, "cdn_relu_backward")
        return gy, None, None


# A/B switch (tools/train_step_bench.py --no-stored-res): stages 1-2 of the QAT step on the stored tensors (True) or on
# the materialised up-sampled ones (False: round 3's path).
STORED_RES_STAGES = True


def forward_stage_blocks(seq, x):
    """``seq(x)`` for a quantised ``deconv_layers`` Sequential; in the QAT step on the GPU the block
    [ReLU, QuantAct] + Upsample(x2, nearest) behind every stage runs as ReluQuantUpsample (one forward and one
    backward kernel instead of relu / fake-quant / upsample and their three backward kernels)."""
    import torch.nn as nn
    from ..portable_quantizer.quant_modules import QuantAct, QuantDeformConvWithOffsetScaleBoundPositive
    mods = list(seq)

    def block_ok(q, post, up):
        return (isinstance(q, QuantDeformConvWithOffsetScaleBoundPositive) and isinstance(post, nn.Sequential)
                and len(post) == 2 and isinstance(post[0], nn.ReLU) and isinstance(post[1], QuantAct)
                and native_act_ok(post[1]) and not getattr(post[1], "global_range", False)
                and isinstance(up, nn.Upsample) and up.mode == "nearest" and up.size is None
                and up.scale_factor in (2, 2.0, (2, 2), (2.0, 2.0)))

    if not (torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and len(mods) % 3 == 0 and mods
            and all(block_ok(*mods[i:i + 3]) for i in range(0, len(mods), 3))):
        return seq(x)
    # hooks on the inner modules must fire: the fused blocks bypass their __call__
    for i in range(0, len(mods), 3):
        for m in (mods[i + 1], mods[i + 1][0], mods[i + 1][1], mods[i + 2]):
            if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks:
                return seq(x)
    lib = N_.lib()
    x_up = False
    pre = pre_pw = None
    if all(mods[i]._train_path_ok(x) for i in range(0, len(mods), 3)):
        stages = [mods[i] for i in range(0, len(mods), 3)]
        pre, pre_pw = small_weights_of_stages(stages), folded_weights_of_stages(stages)
    for i in range(0, len(mods), 3):
        # (y, {min, max} pairs of y) on the native training path; the weights of all stages from two launches
        pw = (pre[i // 3] if pre is not None else (None, None)) + (pre_pw[i // 3] if pre_pw is not None else (None, None))
        y = mods[i](x, want_range=mods[i + 1][1], x_up=x_up, pre_w=pw)      # (the QuantAct behind the stage: see forward)
        y, part = y if isinstance(y, tuple) else (y, None)
        nxt = mods[i + 3] if i + 3 < len(mods) else None
        # the next stage reads its input through the up-sampling (stored tensor, never materialised) where its gather
        # kernels support the shape
        # (a stage with nothing to differentiate takes the inference kernels, which want the materialised tensor)
        x_up = bool(STORED_RES_STAGES and nxt is not None and nxt._train_path_ok(y) and not nxt._fast_path_ok(y)
                    and lib.cdn_codenet_dw_up2_supported(y.shape[0], y.shape[1], 2 * y.shape[2], 2 * y.shape[3]))
        if x_up:
            x = ReluQuant.apply(y, mods[i + 1][1], part)
        else:
            x = ReluQuantUpsample.apply(y, mods[i + 1][1], part)
    return x


def native_act_ok(act):
    """The device QuantAct implements the reference's default: asymmetric, plain batch min/max, quantising."""
    return act.quant_mode == "asymmetric" and not act.percentile and not act.full_precision_flag
