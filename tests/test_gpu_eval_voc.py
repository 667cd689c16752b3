"""SURVEY.md section 8f row 4 on the GPU: the AP50 procedure (tools/eval_voc.py -- image file -> pre_process -> native
network -> ctdet_decode -> post_process -> results.json -> VOC07 11-point AP) run end to end over a synthetic
VOC-shaped tree (the data set and checkpoints are not available offline).  Known answers instead of a published AP:
with the run's own detections as ground truth every class present scores AP = 1, with shifted boxes AP = 0, and the
results.json written on the way has the reference's format (lib/datasets/dataset/pascal.py:58-79)."""
import argparse
import importlib.util
import json
import os

import numpy as np
import pytest
import torch

from codenet_amd import evalio, harness

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _eval_voc():
    spec = importlib.util.spec_from_file_location("eval_voc", os.path.join(ROOT, "tools", "eval_voc.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _tree(tmp_path, sizes):
    from PIL import Image
    rng = np.random.default_rng(3)
    root = tmp_path / "data" / "voc"
    (root / "images").mkdir(parents=True)
    (root / "annotations").mkdir()
    images = []
    for i, (w, h) in enumerate(sizes):
        name = "%06d.jpg" % (i + 1)
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(str(root / "images" / name), quality=95)
        images.append({"id": i + 1, "file_name": name, "width": w, "height": h})
    return root, images


def _write_ann(root, images, annotations):
    cats = [{"id": c, "name": str(c)} for c in range(1, 21)]
    with open(str(root / "annotations" / "pascal_test2007.json"), "w") as f:
        json.dump({"images": images, "annotations": annotations, "categories": cats}, f)


def _records(path, images):
    """results.json as the reference writes it (pascal.py:58-75): [class 0..20][image index] -> rows
    [x1, y1, x2, y2, score]; returned as flat records with COCO-style [x, y, w, h] boxes."""
    det = json.load(open(path))
    assert len(det) == 21 and det[0] == [[] for _ in images] and all(len(d) == len(images) for d in det)
    out = []
    for c in range(1, 21):
        for i, im in enumerate(images):
            for row in det[c][i]:
                assert len(row) == 5
                x1, y1, x2, y2, sc = row
                out.append({"image_id": im["id"], "category_id": c, "bbox": [x1, y1, x2 - x1, y2 - y1], "score": sc})
    return out


def _args(tmp_path, ckpt, quantize, out):
    return argparse.Namespace(data=str(tmp_path / "data"), load_model=ckpt, res=256, quantize=quantize, w2=False,
                              maxpool=False, flip_test=True, limit=0, out=str(tmp_path / out), reference_ap50=None)


@pytest.mark.parametrize("quantize", [False, True])
def test_eval_voc_end_to_end_known_answers(tmp_path, capsys, quantize):
    ev = _eval_voc()
    root, images = _tree(tmp_path, [(500, 375), (333, 500), (256, 256), (480, 360)])
    ckpt = str(tmp_path / "model_last.pth")
    model = harness.create_model(quantize=quantize, seed=11)
    # a random `wh` head predicts negative sizes (boxes with x2 < x1 match nothing, not even themselves): give its
    # last 1x1 conv a positive bias and small weights so that every box is well-formed
    last_w, last_b = [p for n, p in model.named_parameters() if n.startswith("wh.") and p.shape[0] == 2]
    with torch.no_grad():
        last_w.mul_(0.02)
        last_b.fill_(6.0)
    evalio.save_model(ckpt, 1, model)

    # pass 1: no ground truth -> every AP is 0; results.json in the reference's format
    _write_ann(root, images, [])
    ev.run_voc(_args(tmp_path, ckpt, quantize, "o1"))
    out1 = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert out1["AP50"] == 0.0 and out1["images"] == 4
    res = _records(str(tmp_path / "o1" / "results.json"), images)
    assert len(res) > 0
    per_image = {}
    for r in res:
        assert 0.0 < r["score"] < 1.0
        per_image[r["image_id"]] = per_image.get(r["image_id"], 0) + 1
    assert set(per_image) == {1, 2, 3, 4} and max(per_image.values()) <= 100   # max_per_image (ctdet.py:62-72)
    # boxes are in ORIGINAL image coordinates (transform_preds undoes the crop): they spread over each image
    for im in images:
        xs = [r["bbox"][0] + r["bbox"][2] / 2 for r in res if r["image_id"] == im["id"]]
        assert max(xs) - min(xs) > 0.3 * im["width"]

    # pass 2: the run's own detections as ground truth -> a fresh model from the same checkpoint reproduces them
    # (deterministic kernels, same QuantAct history) and every class present has AP = 1
    ann = [{"image_id": r["image_id"], "category_id": r["category_id"], "bbox": r["bbox"], "ignore": 0} for r in res]
    _write_ann(root, images, ann)
    ev.run_voc(_args(tmp_path, ckpt, quantize, "o2"))
    out2 = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    res2 = _records(str(tmp_path / "o2" / "results.json"), images)
    present = {r["category_id"] for r in res}
    if quantize:
        assert res2 == res             # whole network on the native kernels: bitwise repeatable
    # (fp32: the backbone's MIOpen convolutions may pick another algorithm on a later call; a detection swapped at
    # rank 100 costs the recall-1.0 point of the 11, nothing more)
    floor = 1.0 if res2 == res else 10.0 / 11.0
    for c, name in enumerate(ev.CLASSES, 1):
        ap = out2["per_class"][name]
        assert (floor - 1e-9 <= ap <= 1.0 + 1e-9) if c in present else ap == 0.0, (name, ap)

    # pass 3: ground truth moved far away -> no detection reaches IoU 0.5
    far = [dict(a, bbox=[a["bbox"][0] + 4000.0, a["bbox"][1] + 4000.0, a["bbox"][2], a["bbox"][3]]) for a in ann]
    _write_ann(root, images, far)
    ev.run_voc(_args(tmp_path, ckpt, quantize, "o3"))
    assert json.loads(capsys.readouterr().out.strip().splitlines()[-1])["AP50"] == 0.0
