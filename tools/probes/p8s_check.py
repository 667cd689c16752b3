"""Probe check (round 5): the int8 pointwise entry point on given shapes against an integer reference in torch
(codes L = rint(qs * a - qz), sums exact in float64), narrow and wide codes.
    python tools/with_lib.py codenet_amd/lib/libcodenet_dcn_<tag>.so tools/probes/p8s_check.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from codenet_amd import _native as N_


def run(M, K, Co, wide, lda=0, seed=0):
    lib, dev = N_.lib(), torch.device("cuda", 0)
    g = torch.Generator().manual_seed(seed)
    ld = lda or K
    a = torch.zeros(M, ld)
    a[:, :K] = torch.rand(M, K, generator=g) * 4 - 1
    qs, qz = 37.0, -31.0
    t = qs * a[:, :K].double() - qz
    near = ((t - torch.floor(t)) - 0.5).abs() < 1e-3
    a[:, :K] = torch.where(near, a[:, :K] + 0.05 / qs, a[:, :K])
    S = torch.zeros(8)
    S[2], S[3] = qs, qz
    Si = S.view(torch.int32)
    Si[6] = 1 if wide else 0
    Kp = (K + 63) // 64 * 64
    qw = torch.randint(-8, 8, (Co, K), generator=g)
    codes = torch.zeros(Co, Kp, dtype=torch.int8)
    codes[:, :K] = qw.to(torch.int8)
    ws = torch.rand(Co, generator=g) * 20 + 3
    wf = (qw.float() / ws[:, None]).contiguous()
    colsum = qw.sum(1).to(torch.int32)
    bias = torch.randn(Co, generator=g)
    a_d, S_d, codes_d, ws_d, wf_d, cs_d, b_d = [v.to(dev) for v in (a, S, codes, ws, wf, colsum, bias)]
    out = torch.full((M, Co), -7.0, device=dev)
    rmin, rmax, rst = torch.zeros(1, device=dev), torch.zeros(1, device=dev), torch.zeros(8, device=dev)
    wsz = lib.cdn_codenet_aux_workspace_bytes() if hasattr(lib, "cdn_codenet_aux_workspace_bytes") else 1 << 20
    wsp = torch.zeros(max(int(wsz), 1 << 20), dtype=torch.uint8, device=dev)
    rc = lib.cdn_codenet_pointwise_nhwc_forward(
        a_d.data_ptr(), S_d.data_ptr(), M, K, Co, ld, 0, wf_d.data_ptr(), codes_d.data_ptr(), ws_d.data_ptr(),
        cs_d.data_ptr(), b_d.data_ptr(), None, None, 1, rmin.data_ptr(), rmax.data_ptr(), rst.data_ptr(), 8, 0.99, 1,
        wsp.data_ptr(), wsp.numel(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    N_.check(rc, "pw")
    torch.cuda.synchronize()
    L = torch.round((torch.tensor(qs) * a[:, :K]) - qz)
    if not wide:
        L = L.clamp(-2040 - qz + 128, 2039 - qz + 128)
    acc = (L + qz).double() @ qw.double().t()
    ref = torch.relu(acc / (qs * ws.double()) + bias.double())
    d = (out.cpu().double() - ref).abs()
    mag = ref.abs().max().item()
    bad = (d > 1e-5 * mag).nonzero()
    ok = bad.numel() == 0 and abs(rmin.item() - out.min().item()) == 0 and abs(rmax.item() - out.max().item()) == 0
    print("M %6d K %5d Co %4d wide %d lda %5d: max diff %.3g (mag %.3g) bad %d range (%g,%g) vs (%g,%g) %s" % (
        M, K, Co, wide, ld, d.max().item(), mag, bad.shape[0], rmin.item(), rmax.item(), out.min().item(),
        out.max().item(), "OK" if ok else "FAIL"))
    if bad.numel():
        rows, cols = bad[:, 0], bad[:, 1]
        print("   bad rows mod 32:", sorted(set((rows % 32).tolist()))[:40], " cols:", sorted(set(cols.tolist()))[:16],
              " row blocks:", sorted(set((rows // 32).tolist()))[:16])
    return ok


if __name__ == "__main__":
    allok = True
    for wide in (0, 1):
        for (M, K, Co, lda) in [(16384, 1024, 256, 0), (4096, 256, 128, 0), (3200, 128, 128, 0), (8192, 128, 64, 0),
                                (1000, 512, 256, 0), (8192, 2176, 256, 2176), (640, 1024, 200, 0), (2048, 192, 96, 0)]:
            allok &= run(M, K, Co, wide, lda)
    sys.exit(0 if allok else 1)
