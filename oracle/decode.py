"""oracle/decode.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of the reference's ctdet_decode (lib/models/decode.py:474-505 with _nms :10-16,
_topk :110-127 and lib/models/utils.py _gather_feat / _transpose_and_gather_feat).  The reference's
two-level torch.topk leaves the order of EQUAL scores unspecified; this restatement (and the HIP
kernel) order them by ascending flat index class*H*W + y*W + x.  Pinned against the reference's own
function by tests/golden/decode_ref.npz (tests/golden/make_golden.py::make_decode: inputs without ties
among the selected scores) in tests/test_oracle_decode.py.
"""
import numpy as np


def nms_mask(heat):
    """heat * (max_pool2d(heat, 3, stride 1, pad 1) == heat)  (decode.py:10-16; -inf padding)."""
    B, C, H, W = heat.shape
    p = np.full((B, C, H + 2, W + 2), -np.inf, dtype=heat.dtype)
    p[:, :, 1:-1, 1:-1] = heat
    hmax = p[:, :, 1:-1, 1:-1].copy()
    for dy in range(3):
        for dx in range(3):
            hmax = np.maximum(hmax, p[:, :, dy:dy + H, dx:dx + W])
    return heat * (hmax == heat).astype(heat.dtype)


def ctdet_decode(heat, wh, reg=None, cat_spec_wh=False, K=100):
    """-> dets [B, K, 6] float32 = x1, y1, x2, y2, score, class."""
    heat = np.asarray(heat, dtype=np.float32)
    wh = np.asarray(wh, dtype=np.float32)
    B, cat, H, W = heat.shape
    HW = H * W
    masked = nms_mask(heat).reshape(B, cat * HW) + np.float32(0.0)
    dets = np.zeros((B, K, 6), dtype=np.float32)
    for b in range(B):
        s = masked[b]
        # two-level top-K == global top-K (decode.py:114,120); ties: ascending flat index
        order = np.lexsort((np.arange(s.size), -s.astype(np.float64)))[:K]
        score = s[order]
        cls = order // HW
        pix = order % HW
        ys = (pix // W).astype(np.float32)
        xs = (pix % W).astype(np.float32)
        if reg is not None:
            r = np.asarray(reg, dtype=np.float32)[b].reshape(2, HW)
            xs = xs + r[0, pix]
            ys = ys + r[1, pix]
        else:
            xs = xs + np.float32(0.5)
            ys = ys + np.float32(0.5)
        whb = wh[b].reshape(-1, HW)
        if cat_spec_wh:
            w_, h_ = whb[2 * cls, pix], whb[2 * cls + 1, pix]
        else:
            w_, h_ = whb[0, pix], whb[1, pix]
        dets[b, :, 0] = xs - w_ / 2
        dets[b, :, 1] = ys - h_ / 2
        dets[b, :, 2] = xs + w_ / 2
        dets[b, :, 3] = ys + h_ / 2
        dets[b, :, 4] = score
        dets[b, :, 5] = cls.astype(np.float32)
    return dets
