#!/usr/bin/env python
"""bench.py -- images/sec of CoDeNet's deform hot path (the three up-sampling stages,
``deconv_layers``) on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A "step" is one pass of the hot path over one batch of synthetic images whose stage-0 input is
already resident in HBM.  Default workload = BASELINE.json configs[2]:
CoDeNet1x config-c, 512x512, W4A8, batch 64 per GPU (weak scaling: every rank processes its own
64 images; the only collective is the start-up RCCL broadcast of weights / BN stats / QuantAct
ranges).  Rank 0 prints ONE JSON line with `roofline` (dominant kernel, live HIP-event timing on
its launch stream) and `cpu_baseline` (the CPU oracle timed on a bounded sample of the same
workload; N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
F32_MFMA_PEAK_TF = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--w2", action="store_true", help="CoDeNet2x (stage-0 C = 2153)")
    ap.add_argument("--fp32", action="store_true", help="fp32 hot path instead of W4A8")
    ap.add_argument("--frozen", action="store_true",
                    help="freeze QuantAct ranges (default: reference-faithful running ranges)")
    ap.add_argument("--path", choices=["fused", "modules"], default="fused",
                    help="fused: per-stage fused kernel schedule (codenet_fused.hip); modules: the "
                         "reference-shaped nn.Module chain, one op at a time")
    ap.add_argument("--no-graph", action="store_true", help="fused path: launch eagerly instead of "
                    "replaying a captured HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=2)
    return ap.parse_args()


def cpu_baseline(args, net_cpu):
    """The CPU oracle (oracle/ -- the reference has no CPU deform conv) timed
    on a bounded sample: `cpu_images` images of the same workload through the same three stages
    in the reference's algorithmic form (im2col gather + per-group contraction in C/OpenMP,
    torch-CPU conv2d / fake-quant around it).  kind = "port"."""
    import torch.nn.functional as F
    from codenet_amd import pipeline
    from oracle import dcn as O
    from oracle import quant as Q
    n = args.cpu_images
    x = pipeline.make_input(n, args.res, args.w2, seed=1)
    mods = list(net_cpu.deconv_layers)
    cores = torch.get_num_threads()
    O.lib()

    def run(x):
        if args.fp32:
            for i in range(0, len(mods), 4):
                op, bn = mods[i], mods[i + 1]
                r = Q.stage_fp32(x, op.conv_scale.weight, op.conv_scale.bias, op.conv.weight,
                                 op.conv_channel.weight)
                x = F.interpolate(torch.relu(bn(r["y"])), scale_factor=2, mode="nearest")
            return x
        for i in range(0, len(mods), 3):
            q = mods[i]
            bnm = q.quant_conv_channel_bn.bn
            bn = (bnm.weight, bnm.bias, bnm.running_mean, bnm.running_var, bnm.eps)
            r = Q.stage_w4a8(x, q.quant_conv_scale.weight, q.quant_conv_scale.bias,
                             q.quant_deform_conv.weight, q.quant_conv_channel_bn.conv.weight, bn,
                             Q.QuantActState(), Q.QuantActState())
            y = Q.QuantActState()(torch.relu(r["y"]))
            x = F.interpolate(y, scale_factor=2, mode="nearest")
        return x

    with torch.no_grad():
        run(x[:1])                       # warm caches / thread pools
        t0 = time.perf_counter()
        reps = 0
        while True:
            run(x)
            reps += 1
            if time.perf_counter() - t0 > 12.0 or reps >= 400:
                break
        dt = time.perf_counter() - t0
    return {"value": n * reps / dt, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "%d x %d images, %s hot path, %dx%d, oracle C (OpenMP) + torch CPU"
                      % (reps, n, "fp32" if args.fp32 else "W4A8", args.res, args.res)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)          # before the process group: RCCL binds to the current device
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world)   # nccl == RCCL on ROCm
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    from codenet_amd import ops, pipeline

    quantized = not args.fp32
    # rank 0 owns the weights; the other ranks start from different values and receive them
    net = pipeline.build_hot_path(w2=args.w2, quantized=quantized, seed=317 if rank == 0 else 1000 + rank)
    net_cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import copy
        net_cpu = copy.deepcopy(net)
    net = net.to(dev)
    bcast_bytes = pipeline.broadcast_parameters(net, src=0)
    if quantized:
        pipeline.set_running_stat(net, not args.frozen)
    x = pipeline.make_input(args.batch, args.res, args.w2, seed=rank, device=dev)

    import ctypes
    from codenet_amd import _native
    fused = pipeline.FusedHotPath(net.deconv_layers) if args.path == "fused" else None

    def eager_step():
        if fused is not None:
            return fused(x)
        with torch.no_grad():
            return net(x)

    def barrier():
        if world > 1:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    if quantized and args.frozen:          # ranges must exist before they can be frozen
        pipeline.set_running_stat(net, True)
        eager_step()
        pipeline.set_running_stat(net, False)
    for _ in range(max(1, args.warmup // 2)):
        eager_step()
    use_graph = fused is not None and not args.no_graph
    step = fused.capture(x) if use_graph else eager_step
    for _ in range(args.warmup):
        step()
    barrier()
    # ---- timed region: exactly K steps ---------------------------------------------------------
    names = {"scale", "dw", "pointwise", "quantact"}
    with ops.KernelTimer(names if fused is None else set()) as kt:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        barrier()
        dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()
    assert torch.isfinite(out).all()

    # ---- per-kernel durations: HIP events on the launch stream (fused path: the library's own
    #      event pairs around every kernel, K eager steps right after the timed region) ------------
    durs = {}
    if fused is not None and rank == 0:
        lib = _native.lib()
        lib.cdn_profile_enable(1)
        for _ in range(args.steps):
            eager_step()
        torch.cuda.synchronize()
        cap = args.steps * 16
        ids = (ctypes.c_int * cap)()
        tags = (ctypes.c_int * cap)()
        ms = (ctypes.c_float * cap)()
        n = lib.cdn_profile_read(cap, ids, tags, ms)
        lib.cdn_profile_enable(0)
        kname = {0: "scale", 1: "dw", 2: "pointwise", 3: "unpack"}
        for i in range(n):
            durs.setdefault((kname.get(ids[i], "other"), (tags[i],)), []).append(ms[i])
    elif rank == 0:
        durs = kt.durations_ms()

    if rank == 0:
        per_kernel = {}
        for (name, tag), v in durs.items():
            per_kernel.setdefault(name, 0.0)
            per_kernel[name] += sum(v) / args.steps          # ms per step in this kernel
        per_launch = {"%s@%s" % (name, "x".join(str(t) for t in tag)): round(sum(v) / len(v), 5)
                      for (name, tag), v in sorted(durs.items())}
        dominant = max(per_kernel, key=per_kernel.get)
        _, per_stage = pipeline.algorithmic_bytes(args.batch, args.res, args.w2)
        alg = {"scale": 0, "dw": 0, "pointwise": 0}
        for v in per_stage.values():
            for k in alg:
                alg[k] += v[k]
        shapes = pipeline.stage_shapes(args.res, args.w2)
        if fused is not None:
            # fused schedule: stages >= 1 read their input at half resolution (up-sampling folded)
            alg = {"scale": 0, "dw": 0, "pointwise": 0, "unpack": 0}
            for i, (C, Co, H, W) in enumerate(shapes):
                HWs = H * W // (1 if i == 0 else 4)
                alg["scale"] += (C + 1) * HWs * 4 * args.batch
                alg["dw"] += (C * HWs + HWs + C * H * W) * 4 * args.batch
                alg["pointwise"] += (C + Co) * H * W * 4 * args.batch
            C, Co, H, W = shapes[-1]
            alg["unpack"] = (Co * H * W + 4 * Co * H * W) * 4 * args.batch
        flops_pw = sum(2.0 * C * Co * H * W * args.batch
                       for (C, Co, H, W) in pipeline.stage_shapes(args.res, args.w2))
        if dominant == "pointwise":
            ach = flops_pw / (per_kernel["pointwise"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": ach, "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                    "frac": ach / F32_MFMA_PEAK_TF, "traffic": None,
                    "kernel": "pw2_kernel" if fused is not None else "pointwise_kernel"}
        else:
            key = dominant if dominant in alg else "dw"
            nbytes = alg.get(key, 0)
            if dominant == "quantact":      # read for min/max + read & write for the fake-quant
                nbytes = sum(3 * n[0] * 4 for (nm, n) in durs if nm == "quantact")
            ach = nbytes / (per_kernel[dominant] * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": None,
                    "kernel": ({"dw": "dw2_kernel+dw2u_kernel", "scale": "scale_*_kernel", "unpack": "unpack_kernel"}
                               if fused is not None else
                               {"dw": "dw_kernel", "scale": "scale_kernel",
                                "quantact": "minmax_kernel+fake_quant_kernel"}).get(dominant, dominant)}
        # HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes (only valid for
        # the workload they were collected on; null otherwise)
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")))
            w = pm["workload"]
            if (w["res"], w["batch"], w["w2"], w["fp32"], w["path"]) == \
                    (args.res, args.batch, args.w2, args.fp32, args.path) and roof["bound"] == "hbm":
                roof["traffic"] = pm["bytes_per_step"].get(dominant)
                roof["traffic_note"] = "bytes per step (sum over the kernel's launches), PMC FETCH_SIZE*2 + WRITE_SIZE"
                roof["algorithmic_bytes_per_step"] = nbytes
        except Exception:
            pass
        roof["launches_per_step"] = sum(1 for (nm, _t) in durs if nm == dominant)
        roof["ms_per_step_in_kernel"] = per_kernel[dominant]
        res = {
            "metric": "images/sec CoDeNet%s %dx%d ctdet inference (deform hot path: deconv_layers)"
                      % ("2x" if args.w2 else "1x", args.res, args.res),
            "value": world * args.batch * args.steps / dt,
            "unit": "images/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.fp32 else "f32 (W4A8 fake-quant: 4-bit weight / 8-bit activation codes)",
            "data": "synthetic",
            "config": {
                "workload": "CoDeNet%s config-%s %dx%d %s, batch %d per GPU, 3 deform stages "
                            "(scale 1x1 -> QuantAct -> bilinear-gather depthwise 3x3 -> QuantAct -> "
                            "pointwise 1x1 + folded BN -> ReLU -> QuantAct -> upsample x2), %s, %s"
                            % ("2x" if args.w2 else "1x", "d" if args.w2 else "c", args.res, args.res,
                               "fp32" if args.fp32 else "W4A8", args.batch,
                               "frozen QuantAct ranges" if args.frozen else
                               "running QuantAct ranges (reference-faithful)",
                               ("fused per-stage schedule" + (", HIP-graph replay" if use_graph else ", eager launches"))
                               if fused is not None else "module-by-module"),
                "global_batch": world * args.batch,
                "parallelism": "dp%d (independent image shards, start-up RCCL broadcast of %d bytes)"
                               % (world, bcast_bytes),
            },
            "roofline": roof,
            "kernel_ms_per_step": per_kernel,
            "kernel_ms_per_launch": per_launch,
        }
        if net_cpu is not None:
            res["cpu_baseline"] = cpu_baseline(args, net_cpu)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
