"""An AP50-DELTA number through the pinned VOC07 evaluator (VERDICT r5 missing #1) -- the second half of BASELINE.json's
metric ("AP50 delta"), in the only form an environment without VOC2007 and without the reference's checkpoints allows.

The reference's procedure is test.py -> results.json -> tools/reval.py:45-63 -> voc_eval (tools/voc_eval_lib/datasets/
voc_eval.py:66-209, 11-point metric of pascal_voc.py:237-247) against the data set's annotations.  Here, on N seeded
synthetic 512 x 512 images (the normalised domain of lib/detectors/base_detector.py:67) and the synthetic-weight network:

    pseudo ground truth   the detections of the fp32 network on the CPU ORACLE path (module mirrors on torch-CPU, deform_conv
                          bound to oracle/dcn.py, torch decode) above a score threshold chosen so that an image carries
                          `gt_per_image` boxes on average; boxes in image coordinates through the pipeline's own
                          post-processing (evalio.post_process, lib/utils/post_process.py:86-103)
    evaluated             per W4A8 mode (running ranges = the reference default; every QuantAct frozen on common ranges;
                          the byte-code serving schedule) the detections of the CPU oracle path and of the GPU build
                          (whole network on the HIP kernels), both fed the same batches in the same order, each scored
                          with tools/eval_voc.py::voc_eval -- the evaluator tests/test_evalio.py pins on the reference's
                          own voc_eval (tests/golden/voc_eval_ref.npz) -- VOC07 11-point AP at IoU 0.5, mean over the
                          classes that have pseudo ground truth
    reported              AP50 per side and mode, delta = AP50(GPU) - AP50(CPU), and the yardstick: the CPU path against
                          ITSELF on one thread (another fp32 summation order), same images, same model state
    second protocol       ("self"): the CPU W4A8 path's own top detections as ground truth -- AP50(CPU) = 1 by
                          construction, AP50(GPU) and AP50(CPU on one thread) measure how much of the CPU path's
                          detections each reproduces, again through the evaluator

An AP against pseudo ground truth from a RANDOM-weight network is not a VOC AP; the delta between two implementations of
the same network through the same evaluator is what this measures.  The wh head's last conv gets a positive bias (boxes
of ~ 6 output pixels; a random head predicts negative sizes, which match nothing) -- tests/test_gpu_eval_voc.py does the same.

TEST INFRASTRUCTURE (imports oracle/): run as a script
    python tests/proxy_ap.py --images 256 --out gpurun_out/proxy_ap.json
through tools/eval_voc.py --proxy-ap N, or through tests/test_gpu_proxy.py (32 images)."""
import argparse
import copy
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch


def _eval_voc():
    spec = importlib.util.spec_from_file_location("eval_voc", os.path.join(ROOT, "tools", "eval_voc.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def well_formed_boxes(model):
    last_w, last_b = [p for n, p in model.named_parameters() if n.startswith("wh.") and p.shape[0] == 2]
    with torch.no_grad():
        last_w.mul_(0.02)
        last_b.fill_(6.0)


class Rows:
    """Detections of one run as the evaluator wants them: per class a list of (image_id, score, x1, y1, x2, y2)."""

    def __init__(self, res):
        from codenet_amd import evalio
        self.evalio = evalio
        self.meta = {"c": np.array([res / 2.0, res / 2.0], dtype=np.float32), "s": float(res),
                     "out_height": res // 4, "out_width": res // 4}
        self.rows = {c: [] for c in range(1, 21)}
        self.images = 0

    def add(self, dets, first_id):
        dets = dets.detach().cpu().float()
        for i in range(dets.shape[0]):
            per_class = self.evalio.post_process(dets[i:i + 1].clone(), self.meta, 20)
            for c in range(1, 21):
                for r in np.asarray(per_class[c]).reshape(-1, 5):
                    self.rows[c].append((first_id + i, float(r[4]), float(r[0]), float(r[1]), float(r[2]), float(r[3])))
        self.images += dets.shape[0]

    def subset(self, n_images):
        out = Rows.__new__(Rows)
        out.evalio, out.meta, out.images = self.evalio, self.meta, n_images
        out.rows = {c: [r for r in v if r[0] < n_images] for c, v in self.rows.items()}
        return out


def ground_truth(rows, per_image, n_images=None):
    """The rows above the score threshold that leaves per_image boxes per image on average -> (gts, threshold, count)."""
    n_images = n_images or rows.images
    scores = sorted((r[1] for v in rows.rows.values() for r in v if r[0] < n_images), reverse=True)
    keep = min(len(scores), int(per_image * n_images))
    thr = scores[keep - 1] if keep else float("inf")
    gts, count = {}, 0
    for c, v in rows.rows.items():
        per = {}
        for r in v:
            if r[1] >= thr and r[0] < n_images:
                per.setdefault(r[0], []).append(r[2:6])
                count += 1
        gts[c] = {k: (np.array(b, dtype=np.float64), np.zeros(len(b), dtype=bool)) for k, b in per.items()}
    return gts, thr, count


def ap50(ev, rows, gts, n_images=None):
    """mean VOC07 11-point AP at IoU 0.5 over the classes with ground truth (tools/eval_voc.py::voc_eval)"""
    aps = []
    for c in range(1, 21):
        if not gts[c]:
            continue
        det = [r for r in rows.rows[c] if n_images is None or r[0] < n_images]
        aps.append(ev.voc_eval(det, gts[c]))
    return float(np.mean(aps)) if aps else 0.0, len(aps)


def run(images=256, res=512, batch=8, seed=0, gt_per_image=8, yard_images=64, threads=None,
        modes=("w4a8_running", "w4a8_frozen", "w4a8_frozen_bytes"), log=None):
    from codenet_amd import harness
    from tests import proxy_agreement as P
    from oracle import dcn as O
    assert torch.cuda.is_available(), "the proxy compares the GPU build with the CPU oracle path"
    dev = torch.device("cuda", 0)
    if threads:
        torch.set_num_threads(threads)
        O.set_threads(threads)
    nthr = torch.get_num_threads()
    ev = _eval_voc()
    images -= images % batch
    yard_images = min(yard_images - yard_images % batch, images)
    out = {"measured": "PROXY AP50 (no VOC2007 / checkpoint in this environment): VOC07 11-point AP at IoU 0.5 through the "
                       "pinned evaluator, pseudo ground truth from the fp32 CPU-oracle path, %d synthetic %dx%d images, "
                       "synthetic weights (seed 317), batches of %d, CPU side on %d threads" % (images, res, res, batch, nthr),
           "evaluator": "tools/eval_voc.py::voc_eval == tools/voc_eval_lib/datasets/voc_eval.py:66-209 of the reference "
                        "(tests/test_evalio.py, tests/golden/voc_eval_ref.npz)", "gt_per_image": gt_per_image}

    def batches():
        g = torch.Generator().manual_seed(seed)
        for n in range(0, images, batch):
            yield n, torch.randn(batch, 3, res, res, generator=g)

    saved = P._bind_oracle()
    try:
        # -- pseudo ground truth: fp32, CPU oracle path; the GPU fp32 build scored against it on the way
        t0 = time.perf_counter()
        cpu, gpu, _ = P.prepare_pair("fp32", res, batch, seed, dev, edit=well_formed_boxes)
        r_c, r_g = Rows(res), Rows(res)
        for n, x in batches():
            r_c.add(harness.process(cpu, x, flip_test=False, native_decode=False)[1], n)
            r_g.add(harness.process(gpu, x.to(dev), flip_test=False)[1], n)
            if log:
                log("fp32: %d / %d images" % (n + batch, images))
        gts, thr, count = ground_truth(r_c, gt_per_image)
        gts_y, _, _ = ground_truth(r_c, gt_per_image, yard_images)
        a_c, ncls = ap50(ev, r_c, gts)
        a_g, _ = ap50(ev, r_g, gts)
        out["ground_truth"] = {"boxes": count, "score_threshold": thr, "classes_with_boxes": ncls}
        out["fp32"] = {"ap50_cpu": a_c, "ap50_gpu": a_g, "delta_gpu_minus_cpu": a_g - a_c,
                       "seconds": round(time.perf_counter() - t0, 1)}
        del cpu, gpu
        for mode in modes:
            t0 = time.perf_counter()
            cpu, gpu, info = P.prepare_pair(mode, res, batch, seed, dev, edit=well_formed_boxes)
            r_c, r_g, r_1 = Rows(res), Rows(res), Rows(res)
            overflow = 0
            for n, x in batches():
                before = copy.deepcopy(cpu) if (mode == "w4a8_running" and n < yard_images) else None
                r_c.add(harness.process(cpu, x, flip_test=False, native_decode=False)[1], n)
                d_g = harness.process(gpu, x.to(dev), flip_test=False)[1]
                if mode == "w4a8_frozen_bytes" and gpu.frozen_overflowed():
                    overflow += 1                       # the serving contract: recompute on the fp32-valued frozen schedule
                    gpu.enable_fused(frozen_codes=False)
                    d_g = harness.process(gpu, x.to(dev), flip_test=False)[1]
                    gpu.enable_fused(frozen_codes=True)
                r_g.add(d_g, n)
                if n < yard_images:
                    twin = before if before is not None else cpu
                    torch.set_num_threads(1)
                    O.set_threads(1)
                    r_1.add(harness.process(twin, x, flip_test=False, native_decode=False)[1], n)
                    torch.set_num_threads(nthr)
                    O.set_threads(nthr)
                if log:
                    log("%s: %d / %d images" % (mode, n + batch, images))
            a_c, _ = ap50(ev, r_c, gts)
            a_g, _ = ap50(ev, r_g, gts)
            y_c, _ = ap50(ev, r_c, gts_y, yard_images)
            y_1, _ = ap50(ev, r_1, gts_y, yard_images)
            y_g, _ = ap50(ev, r_g, gts_y, yard_images)
            # second protocol: the CPU W4A8 path's own top detections as ground truth
            sgt, sthr, scount = ground_truth(r_c, gt_per_image)
            s_g, _ = ap50(ev, r_g, sgt)
            sgt_y, _, _ = ground_truth(r_c, gt_per_image, yard_images)
            s_gy, _ = ap50(ev, r_g, sgt_y, yard_images)
            s_1y, _ = ap50(ev, r_1, sgt_y, yard_images)
            out[mode] = {
                "ap50_cpu": a_c, "ap50_gpu": a_g, "delta_gpu_minus_cpu": a_g - a_c,
                "yardstick": {"images": yard_images, "ap50_cpu": y_c, "ap50_cpu_one_thread": y_1, "ap50_gpu": y_g,
                              "delta_one_thread_minus_cpu": y_1 - y_c, "delta_gpu_minus_cpu": y_g - y_c,
                              "what": "the CPU oracle path against itself on ONE thread, same images and model state"},
                "self_ground_truth": {"boxes": scount, "score_threshold": sthr, "ap50_cpu": 1.0, "ap50_gpu": s_g,
                                      "first_images": {"images": yard_images, "ap50_gpu": s_gy,
                                                       "ap50_cpu_one_thread": s_1y}},
                "seconds": round(time.perf_counter() - t0, 1), **info}
            if mode == "w4a8_frozen_bytes":
                out[mode]["batches_recomputed_after_overflow"] = overflow
                out[mode]["batches"] = images // batch
            del cpu, gpu
            torch.cuda.empty_cache()
    finally:
        P._unbind(saved)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=256)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--gt-per-image", type=int, default=8)
    ap.add_argument("--yard-images", type=int, default=64)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    res = run(a.images, a.res, a.batch, a.seed, a.gt_per_image, a.yard_images, a.threads or None,
              log=lambda m: print(m, file=sys.stderr, flush=True))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(json.dumps(res, indent=1) + "\n")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
