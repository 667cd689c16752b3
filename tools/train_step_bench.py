"""Config e (BASELINE.json configs[4]): forward + backward + Adam step of the W4A8 deform stages (the QAT step
of quant_main.py restricted to the hot path).  Per stage ONE native autograd function
(codenet_amd/functions/codenet_stage.py): scale 1x1, QuantAct, gather, QuantAct, pointwise 1x1 forward and
backward on the HIP kernels (f32 MFMA for the 1x1 convolutions' data and weight gradients, straight-through
quantisers); the per-channel weight fake-quantisation / BN fold (tiny tensors), ReLU, Upsample and Adam are torch
ops.  --graph replays the whole step as one HIP graph (a QAT step is ~500 small launches).  GPU only."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from codenet_amd import pipeline


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)     # lib/opts.py:91 default batch
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--loss", action="store_true", help="surrogate loss on the stage output instead of a fixed upstream gradient")
    ap.add_argument("--graph", action="store_true", help="capture forward + backward + optimizer step in one HIP graph")
    ap.add_argument("--foreach-adam", action="store_true",
                    help="torch.optim.Adam's default multi-tensor form (8 launches per step) instead of fused=True (1)")
    ap.add_argument("--wt-percentile", action="store_true",
                    help="weight ranges of the README's QAT command (quant_main.py --wt-percentile: 0.1 / 99.9 percentile "
                         "k-th values per output channel, 0.95 x min / max for depthwise channels)")
    ap.add_argument("--torch-weight-prep", action="store_true",
                    help="A/B: the weight transformation as the torch composition (kthvalue with --wt-percentile)")
    ap.add_argument("--no-fuse-dq", action="store_true", help="A/B: store the fake-quantised gather output instead of "
                    "quantising it in the consumers' loads")
    ap.add_argument("--no-stored-res", action="store_true", help="A/B: stages 1-2 on the materialised up-sampled tensors "
                    "(round 3's path) instead of on the stored ones (dw4_kernel<UP>, dw_bwd2u_kernel)")
    ap.add_argument("--no-int8-forward", action="store_true", help="A/B: the forward conv_channel on f32 MFMA "
                    "(pointwise_kernel) instead of the exact integer form on int8 MFMA (pwi8n_kernel)")
    ap.add_argument("--no-bf16-dgrad", action="store_true", help="A/B: the data gradient of conv_channel on f32 MFMA "
                    "instead of the exact bf16 x 3 split (pwb3n_kernel)")
    ap.add_argument("--no-bf16-wgrad", action="store_true", help="A/B: the weight gradient of conv_channel on f32 MFMA")
    ap.add_argument("--no-multi-prep", action="store_true", help="A/B: one weight-prep launch per small weight tensor")
    ap.add_argument("--atomic-dw-bwd", action="store_true", help="A/B: float atomics for grad_s / grad_w_dw (not reproducible)")
    ap.add_argument("--no-fused-update", action="store_true", help="A/B: QuantAct updates as launches of their own behind "
                    "the producers (round 5) instead of in the producers' last workgroup")
    a = ap.parse_args()
    if a.no_fused_update:
        from codenet_amd import ops as _ops3
        _ops3.FUSE_RANGE_UPDATE = False
    if a.atomic_dw_bwd:
        from codenet_amd.functions import codenet_stage as _cs5
        _cs5.REPRODUCIBLE_DW_BWD = False
    if a.no_multi_prep:
        from codenet_amd.functions import codenet_stage as _cs4
        _cs4.MULTI_WEIGHT_PREP = False
    if a.no_bf16_wgrad:
        from codenet_amd.functions import codenet_stage as _cs3
        _cs3.WGRAD_BF16X3 = False
    if a.no_bf16_dgrad:
        from codenet_amd import ops as _ops2
        _ops2.DGRAD_BF16X3 = False
    if a.no_int8_forward:
        from codenet_amd import ops as _ops
        _ops.INT8_FORWARD = False
    if a.no_stored_res:
        from codenet_amd.functions import codenet_stage as _cs1
        _cs1.STORED_RES_STAGES = False
    if a.torch_weight_prep:
        from codenet_amd.functions import codenet_stage as _cs0
        _cs0.native_weight_prep_ok = lambda *args, **kw: False
    if a.no_fuse_dq:
        from codenet_amd.functions import codenet_stage as _cs
        _cs.FUSE_DQ_ON_LOAD = False
    out = measure(batch=a.batch, res=a.res, steps=a.steps, graph=a.graph, fp32=a.fp32, loss=a.loss,
                  foreach_adam=a.foreach_adam, wt_percentile=a.wt_percentile, stored=not a.no_stored_res)
    print(json.dumps(out))


def qat_algorithmic_bytes(batch, res, stored=True):
    """ALGORITHMIC HBM bytes of one QAT step over the three stages (fp32 tensors, every kernel reading its inputs and
    writing its outputs once; weights and their gradients -- <= 1.4 MB -- excluded as in SURVEY 8(d)).  Per stage with
    P = N*H*W output pixels and Ps = stored input pixels (P/4 for stages 1-2 when they read through the up-sampling):
      forward   scale (C+1)*Ps . gather C*Ps + Ps + C*P . pointwise (C + Co)*P . ReLU/QuantAct[/Upsample] block (1 + up)*Co*P
                (up = 1 when the next stage takes the stored tensor, 4 when the x2 up-sampling is materialised)
      backward  block (up + 2)*Co*P (gradient in, pre-ReLU tensor in, gradient out) . pointwise data grad (Co + C)*P
                . weight grad (C + Co)*P . gather backward: read x C*Ps, grad_d C*P, s Ps; write grad_x C*Ps, grad_s Ps
                . scale backward: read grad_s Ps, add into grad_x (read + write) 2*C*Ps
    SURVEY 8(d): "backward ~ 3 x forward bytes"."""
    from codenet_amd import pipeline as P_
    tot_f = tot_b = 0
    shapes = P_.stage_shapes(res, False)
    for i, (C, Co, H, W) in enumerate(shapes):
        P = batch * H * W
        Ps = P // 4 if (stored and i > 0) else P
        up = 1 if (stored and i + 1 < len(shapes)) else 4      # the block's output: stored hand-over or materialised x2
        tot_f += ((C + 1) * Ps + (C * Ps + Ps + C * P) + (C + Co) * P + (Co * P + up * Co * P)) * 4
        tot_b += ((up * Co * P + 2 * Co * P) + (Co + C) * P + (C + Co) * P + (2 * C * Ps + C * P + 2 * Ps)
                  + (Ps + 2 * C * Ps)) * 4
    return {"forward": tot_f, "backward": tot_b, "step": tot_f + tot_b}


def committed_traffic():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for rnd in ("r05", "r04"):
        try:
            pm = json.load(open(os.path.join(root, "profiles", rnd, "pmc_steady_train_step.json")))
            return int(pm["bytes_per_iteration"]), "committed: profiles/%s/pmc_steady_train_step.json" % rnd
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def measure(batch=32, res=512, steps=10, graph=True, fp32=False, loss=False, foreach_adam=False, wt_percentile=False,
            stored=True):
    """One QAT step (forward + backward + Adam) of the hot path; returns the bench record (bench.py's `configs.cfg5_qat`
    leg and this tool's output)."""
    class A:
        pass
    a = A()
    a.batch, a.res, a.steps, a.graph, a.fp32, a.loss = batch, res, steps, graph, fp32, loss
    a.foreach_adam, a.wt_percentile, a.no_stored_res = foreach_adam, wt_percentile, not stored
    net = pipeline.build_hot_path(quantized=not a.fp32, wt_percentile=a.wt_percentile).cuda().train()
    for m in net.modules():                      # BN inside QuantBnConv2d is never called; plain BN in fp32
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    x = pipeline.make_input(a.batch, a.res, device="cuda").requires_grad_(True)
    # lib/opts.py:93 lr; the reference's torch.optim.Adam, in its single-kernel form unless --foreach-adam
    opt = torch.optim.Adam(net.parameters(), lr=1.25e-4, capturable=a.graph, fused=not a.foreach_adam)

    # the stages' output gradient comes from the heads in the real QAT step: a fixed upstream gradient of the
    # output's shape (a loss computed on this 134 MB tensor would add ~160 us of reductions that are not part
    # of the step); --loss keeps the old surrogate loss
    with torch.no_grad():
        go = torch.randn_like(net(x)) * 1e-3

    def step():
        opt.zero_grad(set_to_none=True)
        y = net(x)
        if a.loss:
            loss = y.square().mean()
            loss.backward()
        else:
            loss = y[0, 0, 0, 0].detach()
            y.backward(go)
        opt.step()
        return loss

    if a.graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(g):
            y = net(x)
            if a.loss:
                static_loss = y.square().mean()
                static_loss.backward()
            else:
                static_loss = y[0, 0, 0, 0].detach()
                y.backward(go)
            opt.step()

        def step():   # noqa: F811
            g.replay()
            return static_loss
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    out = {"config": "CoDeNet1x %dx%d %s QAT step over deconv_layers, batch %d" % (
        a.res, a.res, "fp32" if a.fp32 else ("W4A8 --wt-percentile" if a.wt_percentile else "W4A8"), a.batch)
        + (", one HIP graph" if a.graph else ", eager launches"),
        "ms_per_step": round(dt * 1e3, 3),
        "images_per_s": round(a.batch / dt, 1), "probe": float(loss)}
    out["stored_res_stages"] = not a.no_stored_res
    out["roofline_dw_bwd2"] = dw_bwd2_roofline(a.batch, a.res, stored=not a.no_stored_res)
    alg = qat_algorithmic_bytes(a.batch, a.res, stored=not a.no_stored_res)
    traffic, source = committed_traffic() if (a.batch, a.res, a.fp32) == (32, 512, False) else (None, None)
    out["roofline"] = {"bound": "hbm", "unit": "GB/s", "peak": 8000.0, "algorithmic_bytes_per_step": alg["step"],
                       "algorithmic_bytes": alg, "achieved": alg["step"] / dt / 1e9, "frac": alg["step"] / dt / 1e9 / 8000.0,
                       "traffic": traffic, "traffic_source": source,
                       "traffic_frac": (traffic / dt / 1e9 / 8000.0) if traffic else None,
                       "note": "whole step against HBM; its dominant kernel family (the gather backward) is bound by the "
                               "64-bit LDS atomics, see roofline_dw_bwd2"}
    del net, opt, x, go
    torch.cuda.empty_cache()
    return out


def dw_bwd2_roofline(batch, res, stored=True):
    """The step's dominant kernel family against ITS bound: dw_bwd2_kernel is not an HBM kernel (12 B per element
    in, 4 B out) -- each (pixel, channel) pair issues 25 64-bit LDS atomics, and tools/probes/probe_lds_atomics.hip
    measured 9.1 lane-ops per clock and CU for ds_add_u64 on gfx950 (x 256 CUs x 2.4 GHz = 5.59e12 / s).  Timed per
    launch with HIP events at the step's three stage shapes.  stored (round 4): stages 1-2 run dw_bwd2u_kernel on the
    stored tensors -- 25 atomics per 2x2 BLOCK and channel, i.e. a quarter of the atomics for the same gradient; the
    `frac` of those rows is priced on the atomics actually issued, `equiv_frac` on the 25 per pixel the
    full-resolution kernel needs for the same result."""
    from codenet_amd import _native as N_
    lib, dev = N_.lib(), torch.device("cuda", 0)
    peak = 9.1 * 256 * 2.4e9
    g = torch.Generator().manual_seed(0)
    rows, tot_pairs, tot_s = [], 0, 0.0
    for C, H in ((1024, res // 32), (256, res // 16), (128, res // 8)):
        up = stored and C != 1024 and bool(lib.cdn_codenet_dw_up2_supported(batch, C, H, H))
        Hx = H // 2 if up else H
        x = torch.randn(batch, C, Hx, Hx, generator=g).to(dev)
        s = (torch.rand(batch, 1, Hx, Hx, generator=g) * 2.5 + 0.5).to(dev)
        w = (torch.randn(C, 1, 3, 3, generator=g) * 0.3).to(dev)
        gd = (torch.randn(batch, C, H, H, generator=g) * 1e-3).to(dev)
        gx, gs, gw = torch.empty_like(x), torch.empty_like(s), torch.zeros_like(w)
        st = torch.cuda.current_stream().cuda_stream
        # the form the step runs: fixed-order grad_s / grad_w (its reduce launch included in the time) unless --atomic-dw-bwd
        from codenet_amd.functions import codenet_stage as CS
        nws = lib.cdn_codenet_dw_backward_workspace_bytes(batch, C, H, H, int(up)) if CS.REPRODUCIBLE_DW_BWD else 0
        ws = torch.empty(max(nws // 4, 1), device=dev)
        if nws:
            fn = lib.cdn_codenet_dw_up2_backward_r if up else lib.cdn_codenet_dw_backward_r
            tail = (ws.data_ptr(), st)
        else:
            fn = lib.cdn_codenet_dw_up2_backward if up else lib.cdn_codenet_dw_backward
            tail = (st,)

        def run():
            N_.check(fn(x.data_ptr(), s.data_ptr(), w.data_ptr(), gd.data_ptr(), gx.data_ptr(),
                        gs.data_ptr(), gw.data_ptr(), batch, C, H, H, *tail), "dw backward")
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        sec = e0.elapsed_time(e1) / 20 * 1e-3
        pairs = batch * C * H * H
        issued = pairs // 4 if up else pairs
        rows.append({"plane": "%dx%d" % (H, H), "channels": C, "kernel": "dw_bwd2u" if up else "dw_bwd2",
                     "form": "fixed-order partials + reduce" if nws else "float atomics",
                     "us_per_launch": round(sec * 1e6, 1),
                     "lds_atomics_per_s": 25 * issued / sec, "frac": 25 * issued / sec / peak,
                     "equiv_frac": 25 * pairs / sec / peak,
                     "hbm_GBps": (4 * pairs + (8 * pairs // 4 if up else 12 * pairs)) / sec / 1e9})
        tot_pairs += issued
        tot_s += sec
    return {"bound": "lds_atomic", "unit": "64-bit LDS atomics/s", "peak": peak, "achieved": 25 * tot_pairs / tot_s,
            "frac": 25 * tot_pairs / tot_s / peak, "per_stage": rows,
            "note": "25 ds_add_u64 per (pixel, channel) pair; peak = 9.1 lane-ops/clk/CU (probe_lds_atomics) x 256 CUs x "
                    "2.4 GHz; the kernel also reads x and grad_d and writes grad_x: 16 B per pair of HBM traffic (hbm_GBps)"}


if __name__ == "__main__":
    main()
