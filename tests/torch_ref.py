"""Independent PyTorch formulation of (modulated) deformable convolution used to PIN the C
oracle: bilinear sampling is done by ``F.grid_sample(mode='bilinear', padding_mode='zeros',
align_corners=True)`` (per-corner zeroing == the reference's rule), the
contraction by einsum.  Fully differentiable, so fp64 autograd gives independent gradients.
Shares no code with oracle/ or codenet_amd/."""
import torch
import torch.nn.functional as F


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def sample_columns(x, offset, kH, kW, stride, padding, dilation, deformable_groups, mask=None):
    """-> cols [N, C, K, Ho, Wo]"""
    N, C, H, W = x.shape
    sH, sW = _pair(stride)
    pH, pW = _pair(padding)
    dH, dW = _pair(dilation)
    Ho = (H + 2 * pH - (dH * (kH - 1) + 1)) // sH + 1
    Wo = (W + 2 * pW - (dW * (kW - 1) + 1)) // sW + 1
    DG = deformable_groups
    K = kH * kW
    cpdg = C // DG
    ho = torch.arange(Ho, dtype=x.dtype).view(1, 1, Ho, 1)
    wo = torch.arange(Wo, dtype=x.dtype).view(1, 1, 1, Wo)
    cols = []
    off = offset.view(N, DG, K, 2, Ho, Wo)
    for k in range(K):
        i, j = k // kW, k % kW
        ys = ho * sH - pH + i * dH + off[:, :, k, 0]          # [N, DG, Ho, Wo]
        xs = wo * sW - pW + j * dW + off[:, :, k, 1]
        # gate of the reference: positions <= -1 or >= size give zero; grid_sample's zero padding
        # already yields zero there because every corner is out of bounds.
        gx = 2.0 * xs / max(W - 1, 1) - 1.0
        gy = 2.0 * ys / max(H - 1, 1) - 1.0
        grid = torch.stack([gx, gy], dim=-1).view(N * DG, Ho, Wo, 2)
        xin = x.view(N * DG, cpdg, H, W)
        smp = F.grid_sample(xin, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
        smp = smp.view(N, DG, cpdg, Ho, Wo)
        if mask is not None:
            smp = smp * mask.view(N, DG, K, Ho, Wo)[:, :, k].unsqueeze(2)
        cols.append(smp.reshape(N, C, Ho, Wo))
    return torch.stack(cols, dim=2)


def deform_conv_ref(x, offset, weight, stride=1, padding=0, dilation=1, groups=1,
                    deformable_groups=1, mask=None, bias=None):
    N, C, H, W = x.shape
    Co, Cg, kH, kW = weight.shape
    cols = sample_columns(x, offset, kH, kW, stride, padding, dilation, deformable_groups, mask)
    Ho, Wo = cols.shape[-2:]
    G = groups
    cols = cols.view(N, G, Cg, kH * kW, Ho, Wo)
    w = weight.view(G, Co // G, Cg, kH * kW)
    out = torch.einsum("ngckhw,gmck->ngmhw", cols, w).reshape(N, Co, Ho, Wo)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out


ANCHOR = torch.tensor([-1, -1, -1, 0, -1, 1, 0, -1, 0, 0, 0, 1, 1, -1, 1, 0, 1, 1],
                      dtype=torch.float32).view(1, 18, 1, 1)
