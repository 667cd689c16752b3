"""Pins the CPU oracle (oracle/dcn_oracle.c) against independent PyTorch primitives and the
portable known-answer properties of the reference's only test-like file
(lib/models/networks/DCNv2/test.py:32-95).  CPU only."""
import pytest
import torch
import torch.nn.functional as F

from oracle import dcn as O
from tests.torch_ref import ANCHOR, deform_conv_ref

torch.manual_seed(0)

CASES = [
    # N, C, H, W, Co, k, stride, pad, dil, G, DG
    (2, 4, 7, 9, 6, 3, 1, 1, 1, 1, 1),
    (2, 8, 9, 9, 8, 3, 1, 1, 1, 8, 1),      # depthwise (CoDeNet)
    (1, 6, 8, 10, 4, 3, 2, 1, 1, 2, 3),
    (2, 4, 9, 8, 4, 3, 1, 2, 2, 2, 2),
    (1, 4, 6, 6, 8, (1, 3), (1, 2), (0, 1), 1, 1, 1),
]


def _mk(case, dtype, modulated=False, off_scale=2.0):
    N, C, H, W, Co, k, s, p, d, G, DG = case
    kH, kW = (k, k) if isinstance(k, int) else k
    Ho, Wo = O.out_size(H, W, kH, kW, s, p, d)
    x = torch.randn(N, C, H, W, dtype=dtype)
    off = torch.randn(N, DG * 2 * kH * kW, Ho, Wo, dtype=dtype) * off_scale
    w = torch.randn(Co, C // G, kH, kW, dtype=dtype)
    m = torch.rand(N, DG * kH * kW, Ho, Wo, dtype=dtype) if modulated else None
    return x, off, w, m, (s, p, d, G, DG)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_forward_vs_grid_sample(case, dtype):
    x, off, w, _, cfg = _mk(case, dtype)
    got = O.deform_conv_forward(x, off, w, *cfg)
    ref = deform_conv_ref(x.double(), off.double(), w.double(), *cfg)
    tol = 1e-10 if dtype == torch.float64 else 2e-4
    assert got.shape == ref.shape
    assert (got.double() - ref).abs().max().item() < tol


@pytest.mark.parametrize("case", CASES)
def test_modulated_forward_vs_grid_sample(case):
    x, off, w, m, cfg = _mk(case, torch.float64, modulated=True)
    b = torch.randn(w.shape[0], dtype=torch.float64)
    got = O.deform_conv_forward(x, off, w, *cfg, mask=m, bias=b)
    ref = deform_conv_ref(x, off, w, *cfg, mask=m, bias=b)
    assert (got - ref).abs().max().item() < 1e-10


@pytest.mark.parametrize("case", CASES)
def test_backward_vs_autograd_fp64(case):
    x, off, w, _, cfg = _mk(case, torch.float64)
    xr, offr, wr = (t.clone().requires_grad_(True) for t in (x, off, w))
    out = deform_conv_ref(xr, offr, wr, *cfg)
    go = torch.randn_like(out)
    out.backward(go)
    gx, goff = O.deform_conv_backward_input(x, off, w, go, *cfg)
    gw = O.deform_conv_backward_params(x, off, tuple(w.shape), go, *cfg)
    assert (gx - xr.grad).abs().max().item() < 1e-9
    assert (goff - offr.grad).abs().max().item() < 1e-9
    assert (gw - wr.grad).abs().max().item() < 1e-9


@pytest.mark.parametrize("case", CASES[:4])
def test_modulated_backward_vs_autograd_fp64(case):
    x, off, w, m, cfg = _mk(case, torch.float64, modulated=True)
    b = torch.randn(w.shape[0], dtype=torch.float64)
    xr, offr, wr, mr, br = (t.clone().requires_grad_(True) for t in (x, off, w, m, b))
    out = deform_conv_ref(xr, offr, wr, *cfg, mask=mr, bias=br)
    go = torch.randn_like(out)
    out.backward(go)
    gx, goff, gm = O.deform_conv_backward_input(x, off, w, go, *cfg, mask=m)
    gw, gb = O.deform_conv_backward_params(x, off, tuple(w.shape), go, *cfg, mask=m, with_bias=True)
    assert (gx - xr.grad).abs().max().item() < 1e-9
    assert (goff - offr.grad).abs().max().item() < 1e-9
    assert (gm - mr.grad).abs().max().item() < 1e-9
    assert (gw - wr.grad).abs().max().item() < 1e-9
    assert (gb - br.grad).abs().max().item() < 1e-9


def test_grad_weight_accumulates_with_scale():
    # cpp:456-462: gradWeight is accumulated into, times `scale`
    x, off, w, _, cfg = _mk(CASES[0], torch.float64)
    go = torch.randn(2, 6, 7, 9, dtype=torch.float64)
    g1 = O.deform_conv_backward_params(x, off, tuple(w.shape), go, *cfg, scale=1.0)
    g2 = O.deform_conv_backward_params(x, off, tuple(w.shape), go, *cfg, scale=0.5)
    assert torch.allclose(g2, 0.5 * g1, atol=1e-12)


# ---- CoDeNet specialisation: known-answer identities of SURVEY.md section 8c -------------------

def _codenet(x, s, w):
    off = ANCHOR.to(x.dtype) * (s - 1)
    C = x.shape[1]
    return O.deform_conv_forward(x, off, w, 1, 1, 1, C, 1)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_codenet_s1_is_plain_depthwise(dtype):
    x = torch.randn(2, 5, 9, 11, dtype=dtype)
    w = torch.randn(5, 1, 3, 3, dtype=dtype)
    s = torch.ones(2, 1, 9, 11, dtype=dtype)
    ref = F.conv2d(x, w, padding=1, groups=5)
    assert (_codenet(x, s, w) - ref).abs().max().item() < (1e-12 if dtype == torch.float64 else 1e-5)


@pytest.mark.parametrize("d", [2, 3, 5, 8])
def test_codenet_integer_scale_is_dilated_conv(d):
    x = torch.randn(2, 5, 12, 13, dtype=torch.float64)
    w = torch.randn(5, 1, 3, 3, dtype=torch.float64)
    s = torch.full((2, 1, 12, 13), float(d), dtype=torch.float64)
    ref = F.conv2d(x, w, padding=d, dilation=d, groups=5)
    assert (_codenet(x, s, w) - ref).abs().max().item() < 1e-12
    # negative scale mirrors the stencil
    refm = F.conv2d(x, w.flip(2, 3), padding=d, dilation=d, groups=5)
    assert (_codenet(x, -s, w) - refm).abs().max().item() < 1e-12


def test_codenet_zero_scale_collapses_taps():
    x = torch.randn(2, 5, 9, 11, dtype=torch.float64)
    w = torch.randn(5, 1, 3, 3, dtype=torch.float64)
    s = torch.zeros(2, 1, 9, 11, dtype=torch.float64)
    ref = x * w.sum(dim=(2, 3)).view(1, 5, 1, 1)
    assert (_codenet(x, s, w) - ref).abs().max().item() < 1e-12


def test_codenet_random_scale_vs_grid_sample():
    x = torch.randn(2, 5, 9, 11, dtype=torch.float64)
    w = torch.randn(5, 1, 3, 3, dtype=torch.float64)
    s = torch.empty(2, 1, 9, 11, dtype=torch.float64).uniform_(-7, 8)
    off = ANCHOR.double() * (s - 1)
    ref = deform_conv_ref(x, off, w, 1, 1, 1, 5, 1)
    assert (_codenet(x, s, w) - ref).abs().max().item() < 1e-12


# ---- known-answer property of lib/models/networks/DCNv2/test.py:32-65 (check_zero_offset) ---

def test_zero_offset_identity_kernel_half_mask():
    N, C, H, W = 2, 2, 4, 4
    x = torch.randn(N, C, H, W, dtype=torch.float64)
    w = torch.zeros(C, C, 3, 3, dtype=torch.float64)
    for c in range(C):
        w[c, c, 1, 1] = 1.0
    off = torch.zeros(N, 18, H, W, dtype=torch.float64)
    m = torch.sigmoid(torch.zeros(N, 9, H, W, dtype=torch.float64))
    b = torch.zeros(C, dtype=torch.float64)
    out = O.deform_conv_forward(x, off, w, 1, 1, 1, 1, 1, mask=m, bias=b)
    assert (x - 2 * out).abs().max().item() < 1e-10


# ---- gradcheck with the tolerances of DCNv2/test.py:92-95 on the oracle's autograd wrapper --

def test_gradcheck_reference_tolerances():
    x = (torch.rand(2, 2, 4, 4, dtype=torch.float64) * 0.01).requires_grad_(True)
    # keep every sample >= 0.1 px away from the bilinear kinks so eps=1e-3 central differences
    # are valid (the op is only piecewise differentiable in the offsets)
    off = (torch.randint(-3, 3, (2, 18, 4, 4)).double() +
           torch.empty(2, 18, 4, 4, dtype=torch.float64).uniform_(0.1, 0.9)).requires_grad_(True)
    w = torch.randn(2, 2, 3, 3, dtype=torch.float64).requires_grad_(True)
    assert torch.autograd.gradcheck(
        lambda a, b, c: O.deform_conv(a, b, c, 1, 1, 1, 1, 1), (x, off, w),
        eps=1e-3, atol=1e-4, rtol=1e-2)


def test_out_of_range_samples_are_zero_and_have_zero_grad():
    x = torch.randn(1, 2, 5, 5, dtype=torch.float64)
    w = torch.randn(2, 1, 3, 3, dtype=torch.float64)
    off = torch.full((1, 18, 5, 5), 100.0, dtype=torch.float64)
    out = O.deform_conv_forward(x, off, w, 1, 1, 1, 2, 1)
    assert out.abs().max().item() == 0.0
    gx, goff = O.deform_conv_backward_input(x, off, w, torch.ones_like(out), 1, 1, 1, 2, 1)
    assert gx.abs().max().item() == 0.0 and goff.abs().max().item() == 0.0
