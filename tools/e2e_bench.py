"""End-to-end context for the hot path (SURVEY.md section 8d "whole-model context"): whole
PoseShuffleNetV2 + ctdet_decode images/s with the backbone and heads on PyTorch-ROCm and the three
deform stages either module-by-module or on the fused schedule; prints the hot path's time share.
GPU only.  Not the headline metric (bench.py is)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from codenet_amd import harness, pipeline


def timed(fn, steps, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--w2", action="store_true")
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    dev = "cuda"
    model = harness.create_model(w2=a.w2, quantize=not a.fp32).to(dev)
    x = torch.randn(a.batch, 3, a.res, a.res, device=dev)
    fused = pipeline.FusedHotPath(model.deconv_layers)
    fheads = pipeline.FusedHeads({h: getattr(model, h) for h in model.heads})

    def backbone(inp):
        return model.layer4(model.layer3(model.layer2(model.layer1(model.layer0(inp)))))

    def heads(f):
        out = {h: getattr(model, h)(f) for h in model.heads}
        hm = out["hm"].sigmoid_()
        return harness.ctdet_decode(hm, out["wh"], reg=out["reg"], K=100)

    def decode(out):
        hm = out["hm"].sigmoid_()
        return harness.ctdet_decode(hm, out["wh"], reg=out["reg"], K=100)

    def path_heads_fused(f):          # stages + heads on the HIP kernels, no NCHW materialisation between
        return fheads(*fused.forward_nhwc(f))

    with torch.no_grad():
        feat = backbone(x)
        up = fused(feat).clone()
        res = {}
        res["backbone_ms"] = timed(lambda: backbone(x), a.steps, 5) * 1e3
        res["hot_path_modules_ms"] = timed(lambda: model.deconv_layers(feat), a.steps, 5) * 1e3
        res["hot_path_fused_ms"] = timed(lambda: fused(feat), a.steps, 5) * 1e3
        res["heads_decode_ms"] = timed(lambda: heads(up), a.steps, 5) * 1e3
        res["e2e_modules_ms"] = timed(lambda: heads(model.deconv_layers(backbone(x))), a.steps, 5) * 1e3
        res["e2e_fused_ms"] = timed(lambda: heads(fused(backbone(x))), a.steps, 5) * 1e3
        res["path_plus_heads_fused_ms"] = timed(lambda: path_heads_fused(feat), a.steps, 5) * 1e3
        res["decode_ms"] = timed(lambda: decode({k: v.clone() for k, v in path_heads_fused(feat).items()}),
                                 a.steps, 5) * 1e3 - res["path_plus_heads_fused_ms"]
        res["e2e_fused_heads_ms"] = timed(lambda: decode(path_heads_fused(backbone(x))), a.steps, 5) * 1e3
        res["e2e_fused_heads_img_s"] = a.batch / res["e2e_fused_heads_ms"] * 1e3

        def decode_native(out):
            return harness.ctdet_decode_native(out["hm"], out["wh"], reg=out["reg"], K=100,
                                               apply_sigmoid=True, heat_out=out["hm"])
        res["decode_native_ms"] = timed(lambda: decode_native(path_heads_fused(feat)), a.steps, 5) * 1e3 \
            - res["path_plus_heads_fused_ms"]
        res["e2e_all_native_tail_ms"] = timed(lambda: decode_native(path_heads_fused(backbone(x))),
                                              a.steps, 5) * 1e3
        res["e2e_all_native_tail_img_s"] = a.batch / res["e2e_all_native_tail_ms"] * 1e3
        if pipeline.FusedBackbone.supported(model):
            fb = pipeline.FusedBackbone(model)

            def all_native():
                feat_, fq_, hw_ = fb(x)
                return decode_native(fheads(*fused.forward_nhwc(feat_, fq_, hw_)))
            res["backbone_fused_ms"] = timed(lambda: fb(x), a.steps, 5) * 1e3
            res["e2e_all_native_ms"] = timed(all_native, a.steps, 5) * 1e3
            res["e2e_all_native_img_s"] = a.batch / res["e2e_all_native_ms"] * 1e3
            try:
                g2 = torch.cuda.CUDAGraph()
                s2 = torch.cuda.Stream()
                s2.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s2):
                    for _ in range(3):
                        all_native()
                torch.cuda.current_stream().wait_stream(s2)
                with torch.cuda.graph(g2):
                    all_native()
                res["e2e_all_native_graph_ms"] = timed(g2.replay, a.steps, 5) * 1e3
                res["e2e_all_native_graph_img_s"] = a.batch / res["e2e_all_native_graph_ms"] * 1e3
            except Exception as e:     # noqa: BLE001
                res["e2e_all_native_graph_error"] = repr(e)[:200]
    # whole forward captured into one HIP graph (static input buffer)
    try:
        with torch.no_grad():
            g = torch.cuda.CUDAGraph()
            s_ = torch.cuda.Stream()
            s_.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s_):
                for _ in range(3):
                    heads(fused(backbone(x)))
            torch.cuda.current_stream().wait_stream(s_)
            with torch.cuda.graph(g):
                dets = heads(fused(backbone(x)))
            res["e2e_fused_graph_ms"] = timed(g.replay, a.steps, 5) * 1e3
            res["e2e_fused_graph_img_s"] = a.batch / res["e2e_fused_graph_ms"] * 1e3
    except Exception as e:     # noqa: BLE001
        res["e2e_fused_graph_error"] = repr(e)[:200]
    res["e2e_modules_img_s"] = a.batch / res["e2e_modules_ms"] * 1e3
    res["e2e_fused_img_s"] = a.batch / res["e2e_fused_ms"] * 1e3
    res["hot_path_share_fused"] = res["hot_path_fused_ms"] / res["e2e_fused_ms"]
    res["config"] = "CoDeNet%s %dx%d %s batch %d" % ("2x" if a.w2 else "1x", a.res, a.res,
                                                      "fp32" if a.fp32 else "W4A8", a.batch)
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in res.items()}))


if __name__ == "__main__":
    main()
