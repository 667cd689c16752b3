"""Layer 1, first unit, branch 2 at the BASELINE shape (batch 64, 512 x 512: 24 -> 58 channels at 128 x 128, stride-2
depthwise to 64 x 64): the stored-tensor pair (cdn_codenet_pointwise_mixed_forward + cdn_codenet_dw3x3_mixed_forward)
against the recomputing pair (cdn_codenet_pwdw_s2_forward), each alone on the GPU.  Checks equal outputs."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from codenet_amd import _native as N_


def main():
    lib, dev = N_.lib(), torch.device("cuda", 0)
    N, Cin, C, H = 64, 24, 58, 128
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(N, H * H, Cin, generator=g).abs_()).to(dev)
    codes = torch.zeros(C, 64, dtype=torch.int8)
    codes[:, :Cin] = torch.randint(-8, 8, (C, Cin), generator=g, dtype=torch.int8)
    scale = (torch.rand(C, generator=g) * 4 + 4)
    wf = (codes[:, :Cin].float() / scale[:, None]).contiguous()
    colsum = codes.int().sum(1).int()
    bias = torch.randn(C, generator=g) * 0.1
    wdw, bdw = torch.randn(C, 9, generator=g) * 0.3, torch.randn(C, generator=g) * 0.1
    codes, scale, wf, colsum, bias, wdw, bdw = (t.to(dev) for t in (codes, scale, wf, colsum, bias, wdw, bdw))
    aux = lib.cdn_codenet_aux_workspace_bytes()
    ws = torch.zeros(aux // 4 + 64, device=dev)
    ws_ptr = (ws.data_ptr() + 255) // 256 * 256
    ws_bytes = (ws.numel() * 4 - (ws_ptr - ws.data_ptr())) // 256 * 256
    st = torch.cuda.current_stream().cuda_stream

    def act():
        return [torch.zeros(1, device=dev), torch.zeros(1, device=dev), torch.zeros(8, dtype=torch.int32, device=dev)]
    a0 = act()
    # input QuantAct state: range of x
    a0[0].fill_(0.0); a0[1].fill_(float(x.max()))
    sc = 255.0 / float(x.max())
    a0[2].view(torch.float32)[2] = sc
    a0[2].view(torch.float32)[3] = round(sc * 0.0) + 128.0
    outs = {}
    for mode in ("stored", "recompute"):
        a1, a2 = act(), act()
        t1 = torch.zeros(N * H * H, 60, device=dev) if mode == "stored" else None
        t2 = torch.zeros(N * (H // 2) ** 2, 60, device=dev)

        def run():
            if mode == "stored":
                N_.check(lib.cdn_codenet_pointwise_mixed_forward(
                    x.data_ptr(), a0[2].data_ptr(), None, N * H * H, Cin, C, Cin, 60, wf.data_ptr(), codes.data_ptr(),
                    scale.data_ptr(), colsum.data_ptr(), bias.data_ptr(), None, None, 1, None, a1[0].data_ptr(),
                    a1[1].data_ptr(), a1[2].data_ptr(), 8, 0.99, 1, ws_ptr, ws_bytes, t1.data_ptr(), st), "pw")
                N_.check(lib.cdn_codenet_dw3x3_mixed_forward(
                    t1.data_ptr(), a1[2].data_ptr(), None, N, C, H, H, 0, 2, 60, 60, wdw.data_ptr(), bdw.data_ptr(), None,
                    None, 0, a2[0].data_ptr(), a2[1].data_ptr(), a2[2].data_ptr(), 8, 0.99, 1, ws_ptr, ws_bytes,
                    t2.data_ptr(), st), "dw")
            else:
                N_.check(lib.cdn_codenet_pwdw_s2_forward(
                    x.data_ptr(), a0[2].data_ptr(), N, Cin, H, H, Cin, wf.data_ptr(), codes.data_ptr(), scale.data_ptr(),
                    colsum.data_ptr(), bias.data_ptr(), a1[0].data_ptr(), a1[1].data_ptr(), a1[2].data_ptr(), C,
                    wdw.data_ptr(), bdw.data_ptr(), 60, a2[0].data_ptr(), a2[1].data_ptr(), a2[2].data_ptr(), 8, 0.99, 1,
                    ws_ptr, ws_bytes, t2.data_ptr(), st), "pwdw")
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        outs[mode] = (e0.elapsed_time(e1) / 20 * 1e3, t2.clone(), a1[1].item(), a2[0].item(), a2[1].item())
    print(json.dumps({"stored_us": round(outs["stored"][0], 1), "recompute_us": round(outs["recompute"][0], 1),
                      "equal_outputs": bool(torch.equal(outs["stored"][1], outs["recompute"][1])),
                      "ranges": [outs["stored"][2:], outs["recompute"][2:]]}))


if __name__ == "__main__":
    main()
