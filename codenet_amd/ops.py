"""CoDeNet fast-path operators (f32) over the C ABI: the three steps of
``DeformConvWithOffsetScaleBoundPositive.forward`` (modules/dcn_deform_conv.py:323-330 of the
reference) without the 18-channel offset tensor, plus the on-device QuantAct.  GPU tensors only."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _native as N_


class KernelTimer:
    """Optional HIP-event timing of named fast-path launches on the stream they are issued on
    (bench.py's live roofline measurement).  Usage: ``with KernelTimer({"dw"}) as kt: ...`` then
    ``kt.durations_ms()`` after a device synchronise."""
    active = None

    def __init__(self, names):
        self.names = set(names)
        self.records = []          # (name, tag, start_event, end_event)

    def __enter__(self):
        KernelTimer.active = self
        return self

    def __exit__(self, *exc):
        KernelTimer.active = None

    def durations_ms(self):
        out = {}
        for name, tag, e0, e1 in self.records:
            out.setdefault((name, tag), []).append(e0.elapsed_time(e1))
        return out


def _tic(name, tag):
    kt = KernelTimer.active
    if kt is None or name not in kt.names:
        return None
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    return (kt, name, tag, e0)


def _toc(rec):
    if rec is not None:
        kt, name, tag, e0 = rec
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        kt.records.append((name, tag, e0, e1))


def _gpu_f32(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise NotImplementedError("codenet_amd runs on the GPU only (got a %s tensor)" % t.device)
        if t.dtype != torch.float32:
            raise RuntimeError("codenet fast paths are float32, got %s" % t.dtype)


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _p(t):
    return t.data_ptr() if t is not None else None


def _partials(n, like):
    return torch.empty(int(n), 2, device=like.device, dtype=torch.float32)


def codenet_scale(x, w_scale, b_scale, lo, hi, want_range=False):
    """s = Hardtanh(lo,hi)(conv1x1(x; C->1) + b): [N,C,H,W] -> [N,1,H,W].  want_range: also the per-workgroup
    {min, max} pairs of s for the QuantAct behind it (cdn_quantact_forward_partials)."""
    _gpu_f32(x, w_scale, b_scale)
    x = x.contiguous()
    Nb, C, H, W = x.shape
    w = w_scale.contiguous().view(-1)
    if w.numel() != C:
        raise RuntimeError("conv_scale weight must have %d elements, got %d" % (C, w.numel()))
    b = b_scale.contiguous().view(-1) if b_scale is not None else None
    s = x.new_empty(Nb, 1, H, W)
    rec = _tic("scale", (C, H, W))
    if want_range:
        part = _partials(N_.lib().cdn_codenet_scale_range_partials(Nb, H, W), x)
        rc = N_.lib().cdn_codenet_scale_forward_range(_p(x), _p(w), _p(b), _p(s), Nb, C, H, W, float(lo), float(hi),
                                                      _p(part), _stream(x))
        _toc(rec)
        N_.check(rc, "cdn_codenet_scale_forward_range")
        return s, part
    rc = N_.lib().cdn_codenet_scale_forward(_p(x), _p(w), _p(b), _p(s), Nb, C, H, W, float(lo),
                                            float(hi), _stream(x))
    _toc(rec)
    N_.check(rc, "cdn_codenet_scale_forward")
    return s


# A/B switch (tools/train_step_bench.py --no-fused-update): the QAT step's producers update the QuantAct behind them in their
# last workgroup (round 6) instead of leaving {min, max} partials for an update launch
FUSE_RANGE_UPDATE = True


def _arrive(act, device):
    """The QuantAct's arrival counters for the in-kernel range update (zero between launches)."""
    buf = getattr(act, "_arrive_buf", None)
    if buf is None or buf.device != device:
        buf = act._arrive_buf = torch.zeros(N_.lib().cdn_quantact_arrive_words(), dtype=torch.int32, device=device)
    return buf


def _update_args(act, device, snap=None):
    return (_p(act.x_min), _p(act.x_max), _p(act._device_state(device)), _p(_arrive(act, device)),
            int(act.activation_bit), float(act.momentum)) + ((_p(snap),) if snap is not False else ())


def codenet_scale_update(x, w_scale, b_scale, lo, hi, act):
    """codenet_scale + the range update of the QuantAct behind it by the launch's last workgroup
    (cdn_codenet_scale_forward_update): returns the clamped, PRE-quantisation s; act's range / state are updated."""
    _gpu_f32(x, w_scale, b_scale)
    x = x.contiguous()
    Nb, C, H, W = x.shape
    w = w_scale.contiguous().view(-1)
    if w.numel() != C:
        raise RuntimeError("conv_scale weight must have %d elements, got %d" % (C, w.numel()))
    b = b_scale.contiguous().view(-1) if b_scale is not None else None
    s = x.new_empty(Nb, 1, H, W)
    rec = _tic("scale", (C, H, W))
    rc = N_.lib().cdn_codenet_scale_forward_update(_p(x), _p(w), _p(b), _p(s), Nb, C, H, W, float(lo), float(hi),
                                                   *_update_args(act, x.device, None), _stream(x))
    _toc(rec)
    N_.check(rc, "cdn_codenet_scale_forward_update")
    return s


def quantact_apply(x, act):
    """Fake-quantisation of x with the QuantAct's state as its producer left it (cdn_quantact_apply)."""
    x = x.contiguous()
    out = torch.empty_like(x)
    rc = N_.lib().cdn_quantact_apply(_p(x), _p(out), x.numel(), _p(act._device_state(x.device)), _stream(x))
    N_.check(rc, "cdn_quantact_apply")
    return out


def codenet_dw_update_supported(x, up2):
    Nb, C, H, W = x.shape
    return bool(FUSE_RANGE_UPDATE and N_.lib().cdn_codenet_dw_forward_update_supported(
        Nb, C, (2 * H) if up2 else H, (2 * W) if up2 else W, int(bool(up2))))


def codenet_dw_update(x, s, w_dw, act, up2=False):
    """The gather forward + the update of the QuantAct behind d by the launch's last workgroup: (d, state snapshot).
    up2: x, s are the stored tensors (codenet_dw_up2's convention)."""
    _gpu_f32(x, s, w_dw)
    x, s, w_dw = x.contiguous(), s.contiguous(), w_dw.contiguous()
    Nb, C, Hs, Ws = x.shape
    H, W = (2 * Hs, 2 * Ws) if up2 else (Hs, Ws)
    d = x.new_empty(Nb, C, H, W)
    snap = torch.empty(8, dtype=torch.int32, device=x.device)
    rc = N_.lib().cdn_codenet_dw_forward_update(_p(x), _p(s), _p(w_dw), _p(d), Nb, C, H, W, int(bool(up2)),
                                                *_update_args(act, x.device, snap), _stream(x))
    N_.check(rc, "cdn_codenet_dw_forward_update")
    return d, snap


def codenet_dw_range(x, s, w_dw):
    """The gather / depthwise forward (no autograd) + the {min, max} pairs of its output."""
    _gpu_f32(x, s, w_dw)
    x, s, w_dw = x.contiguous(), s.contiguous(), w_dw.contiguous()
    Nb, C, H, W = x.shape
    d = torch.empty_like(x)
    part = _partials(N_.lib().cdn_codenet_dw_range_partials(Nb, C, H, W), x)
    rc = N_.lib().cdn_codenet_dw_forward_range(_p(x), _p(s), _p(w_dw), _p(d), Nb, C, H, W, _p(part), _stream(x))
    N_.check(rc, "cdn_codenet_dw_forward_range")
    return d, part


def codenet_dw_up2(xs, ss, w_dw, want_range=True):
    """The gather / depthwise forward (no autograd) of a stage whose input is the nearest x2 up-sampling of the stored
    tensor xs [N,C,H/2,W/2] with the stored scale plane ss [N,1,H/2,W/2]: d [N,C,H,W] (+ its {min, max} pairs),
    bit-identical to codenet_dw_range on the materialised up-sampled tensors (cdn_codenet_dw_up2_forward)."""
    _gpu_f32(xs, ss, w_dw)
    xs, ss, w_dw = xs.contiguous(), ss.contiguous(), w_dw.contiguous()
    Nb, C, Hs, Ws = xs.shape
    H, W = 2 * Hs, 2 * Ws
    d = xs.new_empty(Nb, C, H, W)
    part = _partials(N_.lib().cdn_codenet_dw_up2_range_partials(Nb, C, H, W), xs) if want_range else None
    rc = N_.lib().cdn_codenet_dw_up2_forward(_p(xs), _p(ss), _p(w_dw), _p(d), Nb, C, H, W, _p(part), _stream(xs))
    N_.check(rc, "cdn_codenet_dw_up2_forward")
    return (d, part) if want_range else d


def quantact_forward_partials(x, act, partials, want_out=True, want_state_copy=False):
    """QuantAct.forward on the device with the batch extremes from the producer's {min, max} pairs: range update in
    place + fake-quantisation, no pass over x for the range.  want_out=False: only the update (the consumer
    fake-quantises while loading); want_state_copy: also a snapshot of the device state (for the backward pass).
    Returns out / (out, state_copy) / state_copy."""
    x = x.contiguous()
    out = torch.empty_like(x) if want_out else None
    snap = torch.empty(8, dtype=torch.int32, device=x.device) if want_state_copy else None      # (all 8 words written)
    rc = N_.lib().cdn_quantact_forward_partials(_p(x), _p(out), x.numel(), _p(act.x_min), _p(act.x_max),
                                                _p(act._device_state(x.device)), _p(partials), partials.shape[0],
                                                int(act.activation_bit), float(act.momentum), 1, _p(snap), _stream(x))
    N_.check(rc, "cdn_quantact_forward_partials")
    if want_out and want_state_copy:
        return out, snap
    return out if want_out else snap


class _CodenetDW(Function):
    """d = depthwise 3x3 deformable conv of x dilated per pixel by s."""

    @staticmethod
    def forward(ctx, x, s, w_dw):
        _gpu_f32(x, s, w_dw)
        x, s, w_dw = x.contiguous(), s.contiguous(), w_dw.contiguous()
        Nb, C, H, W = x.shape
        if tuple(s.shape) != (Nb, 1, H, W):
            raise RuntimeError("s must be [%d,1,%d,%d], got %s" % (Nb, H, W, tuple(s.shape)))
        if tuple(w_dw.shape) != (C, 1, 3, 3):
            raise RuntimeError("depthwise weight must be [%d,1,3,3], got %s" % (C, tuple(w_dw.shape)))
        d = torch.empty_like(x)
        rec = _tic("dw", (C, H, W))
        rc = N_.lib().cdn_codenet_dw_forward(_p(x), _p(s), _p(w_dw), _p(d), Nb, C, H, W, _stream(x))
        _toc(rec)
        N_.check(rc, "cdn_codenet_dw_forward")
        ctx.save_for_backward(x, s, w_dw)
        return d

    @staticmethod
    @once_differentiable
    def backward(ctx, gd):
        x, s, w_dw = ctx.saved_tensors
        gd = gd.contiguous()
        Nb, C, H, W = x.shape
        if not N_.lib().cdn_codenet_dw_backward_supported(H, W):
            return _dw_backward_generic(x, s, w_dw, gd, ctx.needs_input_grad)
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gs = torch.empty_like(s) if ctx.needs_input_grad[1] else None
        gw = torch.zeros_like(w_dw) if ctx.needs_input_grad[2] else None
        # fixed-order sums for grad_s / grad_w_dw (reproducible, and faster than the float atomics: no N x C x 9 atomics
        # onto C x 9 addresses) wherever the form exists
        nws = N_.lib().cdn_codenet_dw_backward_workspace_bytes(Nb, C, H, W, 0)
        if nws:
            ws = torch.empty(nws // 4, device=x.device)
            rc = N_.lib().cdn_codenet_dw_backward_r(_p(x), _p(s), _p(w_dw), _p(gd), _p(gx), _p(gs), _p(gw),
                                                    Nb, C, H, W, _p(ws), _stream(x))
        else:
            rc = N_.lib().cdn_codenet_dw_backward(_p(x), _p(s), _p(w_dw), _p(gd), _p(gx), _p(gs), _p(gw),
                                                  Nb, C, H, W, _stream(x))
        N_.check(rc, "cdn_codenet_dw_backward")
        return gx, gs, gw


def _dw_backward_generic(x, s, w_dw, gd, needs):
    """Planes too large for the LDS-resident backward (H*W above ~17k pixels): the same gradients through
    the generic deform-conv entry points (cdn_deform_conv_backward_input / _parameters) with the offsets
    the reference materialises, offset = anchor * (s - 1) (modules/dcn_deform_conv.py:319-325)."""
    from .functions.dcn_deform_conv import deform_conv
    C = x.shape[1]
    anchor = torch.tensor([v for dy in (-1, 0, 1) for dx in (-1, 0, 1) for v in (dy, dx)],
                          dtype=x.dtype, device=x.device).view(1, 18, 1, 1)
    with torch.enable_grad():
        xg = x.detach().requires_grad_(bool(needs[0]))
        sg = s.detach().requires_grad_(bool(needs[1]))
        wg = w_dw.detach().requires_grad_(bool(needs[2]))
        d = deform_conv(xg, anchor * (sg - 1), wg, 1, 1, 1, C, 1)
        wanted = [t for t, n in zip((xg, sg, wg), needs) if n]
        got = iter(torch.autograd.grad(d, wanted, gd)) if wanted else iter(())
    return tuple(next(got) if n else None for n in needs)


codenet_dw = _CodenetDW.apply


# A/B switch (tools/train_step_bench.py --no-int8-forward): the QAT step's forward conv_channel on the int8 matrix cores
# (cdn_codenet_pointwise_i8_forward_range) where the caller vouches for <= 4-bit per-channel symmetric weights.
INT8_FORWARD = True


def codenet_pointwise(d, w_pw, bias=None, ep_scale=None, ep_shift=None, relu=False, want_range=False, d_state=None,
                      int8_weights=False, keep=None, update_act=None):
    """y = conv1x1(d; C->Co) (+bias) (*ep_scale + ep_shift) (ReLU) on f32 MFMA.  want_range: also the per-workgroup
    {min, max} pairs of y.  d_state (8-word QuantAct state tensor): d holds pre-quantisation values, fake-quantised
    with that state while the kernel loads them.  int8_weights (with d_state): w_pw is a per-channel symmetric <= 4-bit
    fake-quantised weight (q / ws) -- the exact integer form on int8 MFMA (cdn_codenet_pointwise_i8_forward_range);
    keep (a dict): receives "fwd_ws" = (tensor, aligned pointer) of that call's workspace -- the per-channel weight
    scales codenet_pointwise_dgrad_q4 reads in the backward pass."""
    _gpu_f32(d, w_pw, bias, ep_scale, ep_shift)
    d = d.contiguous()
    Nb, C, H, W = d.shape
    w = w_pw.contiguous().view(w_pw.size(0), -1)
    if w.size(1) != C:
        raise RuntimeError("pointwise weight must be [Co,%d,1,1], got %s" % (C, tuple(w_pw.shape)))
    Co = w.size(0)
    y = d.new_empty(Nb, Co, H, W)
    rec = _tic("pointwise", (C, H, W))
    lib = N_.lib()
    if (int8_weights and INT8_FORWARD and d_state is not None and ep_scale is None and not relu
            and lib.cdn_codenet_pointwise_i8_supported(Nb, C, Co, H * W)):
        need = lib.cdn_codenet_pointwise_i8_workspace_bytes(Nb, C, Co, H * W)
        ws = torch.empty(need + 256, dtype=torch.uint8, device=d.device)
        wp = (ws.data_ptr() + 255) // 256 * 256
        bp = _p(bias.contiguous() if bias is not None else None)
        if update_act is not None and FUSE_RANGE_UPDATE:
            # update_act = (QuantAct behind y, relu_range): updated by the launch's last workgroup; no partials
            act, relu_range = update_act
            rc = lib.cdn_codenet_pointwise_i8_forward_update(_p(d), _p(d_state), _p(w), bp, _p(y), Nb, C, Co, H * W, wp, need,
                                                             int(bool(relu_range)), *_update_args(act, d.device, False),
                                                             _stream(d))
            _toc(rec)
            N_.check(rc, "cdn_codenet_pointwise_i8_forward_update")
            if keep is not None:
                keep["fwd_ws"] = (ws, wp)
                keep["range_committed"] = True
            return (y, None) if want_range else y
        part = _partials(lib.cdn_codenet_pointwise_i8_range_partials(Nb, C, Co, H * W), d) if want_range else None
        rc = lib.cdn_codenet_pointwise_i8_forward_range(_p(d), _p(d_state), _p(w), bp, _p(y), Nb, C, Co, H * W,
                                                        _p(part), wp, need, _stream(d))
        _toc(rec)
        N_.check(rc, "cdn_codenet_pointwise_i8_forward_range")
        if keep is not None:
            keep["fwd_ws"] = (ws, wp)
        return (y, part) if want_range else y
    tail = (_p(w), _p(bias.contiguous() if bias is not None else None),
            _p(ep_scale.contiguous() if ep_scale is not None else None),
            _p(ep_shift.contiguous() if ep_shift is not None else None), _p(y), Nb, C, Co, H * W, int(bool(relu)))
    if want_range or d_state is not None:
        part = _partials(N_.lib().cdn_codenet_pointwise_range_partials(Nb, Co, H * W), d) if want_range else None
        rc = N_.lib().cdn_codenet_pointwise_forward_range(_p(d), _p(d_state), *tail, _p(part), _stream(d))
        _toc(rec)
        N_.check(rc, "cdn_codenet_pointwise_forward_range")
        return (y, part) if want_range else y
    args = (_p(d),) + tail
    rc = N_.lib().cdn_codenet_pointwise_forward(*args, _stream(d))
    _toc(rec)
    N_.check(rc, "cdn_codenet_pointwise_forward")
    return y


def codenet_pointwise_dgrad_q4(gy, w_pw, fwd_ws):
    """grad_d = conv1x1(grad_y, w_pw^T) for a 4-bit fake-quantised w_pw [Co,C,1,1] with exact products on bf16 MFMA
    (cdn_codenet_pointwise_dgrad_q4); fwd_ws: what codenet_pointwise(..., int8_weights=True, keep=...) kept.  None when the
    shape is not supported (the caller takes the f32 path)."""
    lib = N_.lib()
    gy = gy.contiguous()
    Nb, Co, H, W = gy.shape
    w = w_pw.contiguous().view(Co, -1)
    C = w.size(1)
    if not (DGRAD_BF16X3 and lib.cdn_codenet_pointwise_dgrad_q4_supported(Nb, C, Co, H * W)):
        return None
    gd = gy.new_empty(Nb, C, H, W)
    rec = _tic("pointwise", (Co, H, W))
    rc = lib.cdn_codenet_pointwise_dgrad_q4(_p(gy), fwd_ws[1], _p(gd), Nb, C, Co, H * W, _stream(gy))
    _toc(rec)
    N_.check(rc, "cdn_codenet_pointwise_dgrad_q4")
    return gd


# A/B switch (tools/train_step_bench.py --no-bf16-dgrad)
DGRAD_BF16X3 = True


def quantact_state(device):
    """Device scratch for cdn_quantact_forward (holds scale / zero-point after a call)."""
    nbytes = N_.lib().cdn_quantact_state_bytes()
    return torch.zeros(nbytes // 4, dtype=torch.int32, device=device)


def kth_values(x, k_lo, k_hi):
    """(k_lo-th, k_hi-th) smallest elements of the GPU float32 tensor x, flattened -- torch.kthvalue's values for both
    ranks from three histogram passes (cdn_kth_values); tensors of shape [1].  The ranks are the reference's
    round(n * pct / 100) (quant_utils.py:18-30)."""
    _gpu_f32(x)
    x = x.contiguous()
    lib = N_.lib()
    need = lib.cdn_kth_values_workspace_bytes()
    ws = torch.empty(need // 4 + 64, dtype=torch.int32, device=x.device)
    wp = (ws.data_ptr() + 255) // 256 * 256
    out = torch.empty(2, device=x.device)
    rc = lib.cdn_kth_values(_p(x), x.numel(), int(k_lo), int(k_hi), out.data_ptr(), out.data_ptr() + 4, wp, need,
                            _stream(x))
    N_.check(rc, "cdn_kth_values")
    return out[0:1], out[1:2]


def quantact_forward(x, x_min, x_max, state, bits=8, momentum=0.99, running=True,
                     batch_min=None, batch_max=None, want_codes=False, want_out=True):
    """On-device QuantAct (quant_modules.py:202-225).  Updates x_min / x_max IN PLACE when
    `running`; returns (fake-quantised fp32 tensor or None, int16 codes or None)."""
    _gpu_f32(x, x_min, x_max, batch_min, batch_max)
    x = x.contiguous()
    out = torch.empty_like(x) if want_out else None
    codes = torch.empty(x.shape, dtype=torch.int16, device=x.device) if want_codes else None
    rec = _tic("quantact", (x.numel(),))
    rc = N_.lib().cdn_quantact_forward(_p(x), _p(out), _p(codes), x.numel(), _p(x_min), _p(x_max),
                                       _p(state), _p(batch_min), _p(batch_max), int(bits),
                                       float(momentum), int(bool(running)), _stream(x))
    _toc(rec)
    N_.check(rc, "cdn_quantact_forward")
    return out, codes
