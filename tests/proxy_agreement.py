"""SURVEY.md section 8(d) detection-agreement PROXY at size (VERDICT r4 missing #1): the stand-in for "AP50 within 0.1 of the
reference" that an environment without VOC2007 and without the reference's checkpoints allows.

On N seeded synthetic images at 512 x 512 (the pre-processed, mean / std-normalised domain of
lib/detectors/base_detector.py:67) the build's whole network on the GPU (harness.enable_fused: backbone, deform stages,
heads and ctdet_decode on the HIP kernels) is compared, image by image, with the SAME model on the CPU -- the module
mirrors on torch-CPU with the module-level deform_conv bound to the C oracle (oracle/dcn.py), decode by the torch
restatement of lib/models/decode.py:474-505 -- i.e. the composition bench.py's cpu_baseline_e2e times.  Procedure per
image as CtdetDetector.process (lib/detectors/ctdet.py:29-46, without --flip_test).  Reported per mode:

    max |delta sigmoid(hm)|, mean |delta| of hm (logits) / wh / reg
    top-100 agreement: share of the CPU side's 100 detections for which the GPU side has a detection of the same class
    whose box centre is within `centre_tol` output pixels and whose score is within `score_tol`
    site agreement: at the CPU side's 100 peak sites (class, y, x), the GPU side's score within score_tol and its wh / reg
    within 2 centre_tol -- the same tolerances without the top-K ORDER, which a 0.5 % perturbation of a random-weight
    heat map (thousands of near-equal peaks) reshuffles
    cpu_vs_itself_one_thread: the yardstick at the same operating point -- the CPU path against itself on ONE thread

Modes: fp32 . w4a8_frozen (every QuantAct frozen on ONE common set of ranges: no cross-image coupling, no range chaos;
the fused fp32-valued schedule) . w4a8_frozen_bytes (the same ranges, the byte-code serving schedule) . w4a8_running
(reference default: ranges keep moving with every batch, both sides fed the same batches in the same order) -- next to
the yardstick tests/golden/model_noise.npz: the REFERENCE against ITSELF with 1 vs 8 CPU threads.

TEST INFRASTRUCTURE (imports oracle/): run as a script
    python tests/proxy_agreement.py --images 256 --out gpurun_out/proxy_256.json
or through tests/test_gpu_proxy.py (32 images).  tools/eval_voc.py --proxy-images N starts this script."""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch


def _bind_oracle():
    from codenet_amd.modules import dcn_deform_conv as M
    from codenet_amd.portable_quantizer import quant_modules as QM
    from oracle import dcn as O
    O.lib()
    saved = (M.deform_conv, QM.deform_conv)
    M.deform_conv = QM.deform_conv = O.deform_conv
    return saved


def _unbind(saved):
    from codenet_amd.modules import dcn_deform_conv as M
    from codenet_amd.portable_quantizer import quant_modules as QM
    M.deform_conv, QM.deform_conv = saved


def _agreement(ref, ours, centre_tol, score_tol):
    """ref / ours: [K, 6] (x1, y1, x2, y2, score, class) of one image."""
    oc = torch.stack([(ours[:, 0] + ours[:, 2]) / 2, (ours[:, 1] + ours[:, 3]) / 2], 1)
    rc = torch.stack([(ref[:, 0] + ref[:, 2]) / 2, (ref[:, 1] + ref[:, 3]) / 2], 1)
    same_cls = ours[:, 5].view(1, -1) == ref[:, 5].view(-1, 1)
    near = (oc.view(1, -1, 2) - rc.view(-1, 1, 2)).abs().max(dim=2).values <= centre_tol
    score = (ours[:, 4].view(1, -1) - ref[:, 4].view(-1, 1)).abs() <= score_tol
    return (same_cls & near & score).any(dim=1).float().mean().item()


def _site_agreement(o_ref, o_got, i, K, score_tol, box_tol):
    """Independent of the top-K ORDER (which a 0.5 % perturbation of a random-weight heat map reshuffles): at the
    reference side's K peak sites (class, y, x) of image i -- lib/models/decode.py:474-489 -- the other side's score must
    be within score_tol and its wh / reg within box_tol."""
    from codenet_amd import harness
    hm = o_ref["hm"][i:i + 1]
    _, inds, clses, ys, xs = harness._topk(harness._nms(hm), K=K)
    c, y, x = clses[0].long(), ys[0].long(), xs[0].long()
    ok = (o_ref["hm"][i, c, y, x] - o_got["hm"][i, c, y, x]).abs() <= score_tol
    for k in ("wh", "reg"):
        ok &= (o_ref[k][i, :, y, x] - o_got[k][i, :, y, x]).abs().max(dim=0).values <= box_tol
    return ok.float().mean().item()


def _set_ranges(dst, src):
    """copy every QuantAct range (x_min / x_max buffers) of src into dst (same architecture)"""
    sd = {k: v for k, v in src.state_dict().items() if k.endswith(("x_min", "x_max"))}
    missing = dst.load_state_dict({k: v.detach().cpu().clone() for k, v in sd.items()}, strict=False)
    assert not missing.unexpected_keys
    return len(sd)


def prepare_pair(mode, res, batch, seed, dev, edit=None, margin=0.05):
    """(cpu model, gpu model, info) for one mode of the proxies: one synthetic-weight network (seed 317) on both sides;
    edit(model): applied to the common base before the copies (tests/proxy_ap.py gives the wh head well-formed boxes).
    w4a8_frozen / w4a8_frozen_bytes: ONE common set of ranges -- the GPU network's running ranges settled on calibration
    batches, (bytes: widened by the serving calibration until no code saturates,) copied into the CPU model; both frozen."""
    from codenet_amd import harness, pipeline
    quant = mode != "fp32"
    base = harness.create_model(quantize=quant, seed=317)
    if edit is not None:
        edit(base)
    cpu, gpu = copy.deepcopy(base), copy.deepcopy(base).to(dev)
    info = {}
    if mode in ("w4a8_frozen", "w4a8_frozen_bytes"):
        cal = torch.randn(batch, 3, res, res, generator=torch.Generator().manual_seed(seed + 999)).to(dev)
        gpu.enable_fused()
        with torch.no_grad():
            for _ in range(60):
                gpu(cal)
        pipeline.set_running_stat(gpu, False)
        if mode == "w4a8_frozen_bytes":
            # serving calibration on the byte schedule itself over several batches of the image distribution (a
            # byte cannot hold what the reference's unclamped codes can: pipeline.calibrate_serving only widens)
            gc = torch.Generator().manual_seed(seed + 1999)
            cals = [cal] + [torch.randn(batch, 3, res, res, generator=gc).to(dev) for _ in range(7)]
            info["calibration"] = pipeline.calibrate_serving(gpu, cals, margin=margin)
        info["ranges_copied"] = _set_ranges(cpu, gpu)
        pipeline.set_running_stat(cpu, False)
        gpu.enable_fused(frozen_codes=(mode == "w4a8_frozen_bytes"))
    else:
        gpu.enable_fused()
    return cpu, gpu, info


def compare(images=256, res=512, batch=8, seed=0, modes=("fp32", "w4a8_frozen", "w4a8_frozen_bytes", "w4a8_running"),
            threads=None, log=None, yard_images=16):
    from codenet_amd import harness, pipeline
    assert torch.cuda.is_available(), "the proxy compares the GPU build with the CPU oracle path"
    dev = torch.device("cuda", 0)
    if threads:
        torch.set_num_threads(threads)
        from oracle import dcn as O
        O.set_threads(threads)
    noise = {k: float(v) for k, v in np.load(os.path.join(ROOT, "tests", "golden", "model_noise.npz")).items()}
    out = {"measured": "PROXY (no VOC2007 / checkpoint in this environment): GPU build vs the CPU oracle path on %d "
                       "synthetic %dx%d images, synthetic weights (seed 317), batches of %d" % (images, res, res, batch),
           "tolerances": {"fp32": {"centre_px": 0.05, "score": 2e-3}, "w4a8": {"centre_px": 0.5, "score": 2e-2}},
           "reference_vs_itself_w4a8_running": {
               "mean_abs_diff": {k: noise["noise_mean_" + k] for k in ("hm", "wh", "reg")},
               "top100_agreement": noise["self_agreement"],
               "what": "the reference's own Python model, 256 x 256, 1 vs 8 CPU threads (tests/golden/model_noise.npz)"}}
    saved = _bind_oracle()
    try:
        for mode in modes:
            t0 = time.perf_counter()
            quant = mode != "fp32"
            cpu, gpu, info = prepare_pair(mode, res, batch, seed, dev)
            g = torch.Generator().manual_seed(seed)
            worst_sig, agree, n_img, overflow = 0.0, [], 0, 0
            sites, yard, yard_sites, cpu_before = [], [], [], None
            means = {"hm": 0.0, "wh": 0.0, "reg": 0.0}
            yard_means = {"hm": 0.0, "wh": 0.0, "reg": 0.0}
            tol = (0.05, 2e-3) if mode == "fp32" else (0.5, 2e-2)
            while n_img < images:
                b = min(batch, images - n_img)
                if mode == "w4a8_running" and b != batch:
                    break                                   # (running ranges are batch statistics: whole batches only)
                x = torch.randn(b, 3, res, res, generator=g)
                if mode == "w4a8_running" and n_img < yard_images:
                    cpu_before = copy.deepcopy(cpu)         # (the twin must see the ranges this batch started from)
                o_c, d_c = harness.process(cpu, x, flip_test=False, native_decode=False)
                o_g, d_g = harness.process(gpu, x.to(dev), flip_test=False)
                if mode == "w4a8_frozen_bytes" and gpu.frozen_overflowed():
                    # the serving contract: a batch with a saturated byte code is recomputed on
                    # the fp32-valued frozen schedule
                    overflow += 1
                    gpu.enable_fused(frozen_codes=False)
                    o_g, d_g = harness.process(gpu, x.to(dev), flip_test=False)
                    gpu.enable_fused(frozen_codes=True)
                hm_c, hm_g = o_c["hm"], o_g["hm"].cpu()     # both after the sigmoid (ctdet.py:32)
                worst_sig = max(worst_sig, (hm_c - hm_g).abs().max().item())
                means["hm"] += (torch.logit(hm_c.clamp(1e-7, 1 - 1e-7)) - torch.logit(hm_g.clamp(1e-7, 1 - 1e-7))).abs().mean().item() * b
                for k in ("wh", "reg"):
                    means[k] += (o_c[k] - o_g[k].cpu()).abs().mean().item() * b
                d_g = d_g.cpu()
                o_gc = {k: (hm_g if k == "hm" else o_g[k].cpu()) for k in ("hm", "wh", "reg")}
                for i in range(b):
                    agree.append(_agreement(d_c[i], d_g[i], *tol))
                    sites.append(_site_agreement(o_c, o_gc, i, 100, tol[1], 2 * tol[0]))
                # the yardstick at THIS operating point: the CPU side against ITSELF on one thread (another summation order
                # inside torch-CPU's convolutions and the oracle's OpenMP loops), same model state, first batches only
                if quant and n_img < yard_images and mode != "w4a8_frozen_bytes":
                    from oracle import dcn as O
                    nthr = torch.get_num_threads()
                    twin = copy.deepcopy(cpu_before) if mode == "w4a8_running" else cpu
                    torch.set_num_threads(1)
                    O.set_threads(1)
                    o_1, d_1 = harness.process(twin, x, flip_test=False, native_decode=False)
                    torch.set_num_threads(nthr)
                    O.set_threads(nthr)
                    for i in range(b):
                        yard.append(_agreement(d_c[i], d_1[i], *tol))
                        yard_sites.append(_site_agreement(o_c, o_1, i, 100, tol[1], 2 * tol[0]))
                    for k in ("wh", "reg"):
                        yard_means[k] += (o_c[k] - o_1[k]).abs().mean().item() * b
                    yard_means["hm"] += (torch.logit(hm_c.clamp(1e-7, 1 - 1e-7))
                                         - torch.logit(o_1["hm"].clamp(1e-7, 1 - 1e-7))).abs().mean().item() * b
                n_img += b
                if log:
                    log("%s: %d / %d images, agreement so far %.4f" % (mode, n_img, images, sum(agree) / len(agree)))
            out[mode] = dict(images=n_img, max_abs_diff_sigmoid_hm=worst_sig,
                             mean_abs_diff={k: v / max(1, n_img) for k, v in means.items()},
                             top100_agreement=sum(agree) / max(1, len(agree)), worst_image_agreement=min(agree),
                             images_with_full_agreement=sum(1 for a in agree if a == 1.0),
                             site_agreement=sum(sites) / max(1, len(sites)), worst_image_site_agreement=min(sites),
                             seconds=round(time.perf_counter() - t0, 1), **info)
            if yard:
                out[mode]["cpu_vs_itself_one_thread"] = dict(
                    images=len(yard), top100_agreement=sum(yard) / len(yard), site_agreement=sum(yard_sites) / len(yard_sites),
                    mean_abs_diff={k: v / len(yard) for k, v in yard_means.items()},
                    what="the CPU oracle path against itself on ONE thread (same model state, same images): the "
                         "reproducibility of the reference-shaped computation at this operating point")
            if mode == "w4a8_frozen_bytes":
                out[mode]["batches_recomputed_after_overflow"] = overflow
            del cpu, gpu
            torch.cuda.empty_cache()
    finally:
        _unbind(saved)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=256)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0, help="CPU threads of the oracle side (default: torch's)")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    res = compare(a.images, a.res, a.batch, a.seed, threads=a.threads or None,
                  log=lambda m: print(m, file=sys.stderr, flush=True))
    txt = json.dumps(res, indent=1)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(txt + "\n")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
