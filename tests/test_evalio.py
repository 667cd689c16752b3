"""Checkpoint / evaluation I/O: host-side mirrors of lib/models/model.py
load_model / save_model, lib/utils/post_process.py ctdet_post_process, lib/datasets/dataset/pascal.py
results.json, and the packed 4-bit export.  CPU only."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from codenet_amd import evalio, harness


def test_checkpoint_roundtrip_reference_semantics(tmp_path):
    m = harness.create_model(quantize=False)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    p = str(tmp_path / "model_last.pth")
    evalio.save_model(p, 7, m, opt)
    ck = torch.load(p)
    assert set(ck) == {"epoch", "state_dict", "optimizer"} and ck["epoch"] == 7
    # a DataParallel-prefixed checkpoint with one wrong shape and one unknown key
    sd = {"module." + k: v for k, v in ck["state_dict"].items()}
    sd["module.hm.6.bias"] = torch.zeros(3)
    sd["module.unknown.weight"] = torch.zeros(1)
    torch.save({"epoch": 90, "state_dict": sd, "optimizer": ck["optimizer"]}, p)
    m2 = harness.create_model(quantize=False, seed=5)
    keep = m2.hm[6].bias.clone()
    opt2 = torch.optim.Adam(m2.parameters(), lr=1e-3)
    m2, opt2, ep = evalio.load_model(m2, p, opt2, resume=True, lr=1e-3, lr_step=[45, 60, 120], verbose=False)
    assert ep == 90 and abs(opt2.param_groups[0]["lr"] - 1e-5) < 1e-12
    assert torch.equal(m2.hm[6].bias, keep)                       # shape mismatch: model's own tensor kept
    assert torch.equal(m2.layer0[0].weight, m.layer0[0].weight)   # prefix stripped, value loaded


def test_quantised_checkpoint_keys_roundtrip(tmp_path):
    m = harness.create_model(quantize=True)
    for mod in m.modules():
        if hasattr(mod, "x_min") and isinstance(mod.x_min, torch.Tensor):
            mod.x_min.fill_(-1.5)
            mod.x_max.fill_(2.5)
    p = str(tmp_path / "q.pth")
    evalio.save_model(p, 1, m)
    m2 = evalio.load_model(harness.create_model(quantize=True, seed=9), p, verbose=False)
    for (ka, va), (kb, vb) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)


def test_transform_preds_and_post_process():
    # 512x512 crop of a 500x375 image: center (250, 187.5), scale = max(h, w) = 500, output map 128x128
    c, s = np.array([250.0, 187.5]), 500.0
    pts = np.array([[64.0, 64.0], [0.0, 0.0], [128.0, 128.0]])
    t = evalio.transform_preds(pts, c, s, (128, 128))
    assert np.allclose(t, [[250.0, 187.5], [0.0, -62.5], [500.0, 437.5]])
    dets = np.zeros((1, 4, 6), dtype=np.float32)
    dets[0, :, :4] = [[60, 60, 68, 70], [10, 20, 30, 40], [0, 0, 128, 128], [64, 64, 64, 64]]
    dets[0, :, 4] = [0.9, 0.8, 0.1, 0.5]
    dets[0, :, 5] = [0, 2, 0, 19]
    meta = {"c": c, "s": s, "out_height": 128, "out_width": 128}
    r = evalio.post_process(torch.from_numpy(dets), meta, 20)
    assert sorted(r) == list(range(1, 21)) and r[1].shape == (2, 5) and r[3].shape == (1, 5) and r[20].shape == (1, 5)
    k = 500.0 / 128
    assert np.allclose(r[3][0], [250 + (10 - 64) * k, 187.5 + (20 - 64) * k, 250 + (30 - 64) * k,
                                 187.5 + (40 - 64) * k, 0.8], atol=1e-4)
    merged = evalio.merge_outputs([r], 20, max_per_image=2)
    assert sum(len(v) for v in merged.values()) == 2 and len(merged[1]) == 1 and len(merged[3]) == 1


def test_results_json_format(tmp_path):
    images = [11, 42]
    res = {11: {j: np.zeros((0, 5), np.float32) for j in range(1, 21)},
           42: {j: np.zeros((0, 5), np.float32) for j in range(1, 21)}}
    res[42][7] = np.array([[1, 2, 3, 4, 0.5]], np.float32)
    path = evalio.save_results(res, images, 20, str(tmp_path))
    d = json.load(open(path))
    assert os.path.basename(path) == "results.json" and len(d) == 21 and len(d[7]) == 2
    assert d[7][1] == [[1.0, 2.0, 3.0, 4.0, 0.5]] and d[7][0] == [] and d[0] == [[], []]


def test_export_w4_is_bit_exact(tmp_path):
    m = harness.create_model(quantize=True)
    out = evalio.export_w4(m, str(tmp_path / "w4.npz"))
    z = np.load(str(tmp_path / "w4.npz"))
    assert set(z.files) == set(out)
    q = m.deconv_layers[0].quant_conv_channel_bn
    name = "deconv_layers.0.quant_conv_channel_bn"
    codes = evalio.unpack_int4(torch.from_numpy(z[name + ".codes_packed"]))
    shape = tuple(z[name + ".shape"])
    k = int(np.prod(shape[1:]))
    w = (codes[:, :k].float() / torch.from_numpy(z[name + ".scale"]).view(-1, 1)).view(shape)
    assert codes.min() >= -8 and codes.max() <= 7
    assert torch.equal(w, q.folded()[0])                      # codes / scale == the fake-quantised weight
    assert np.allclose(z[name + ".bias"], q.folded()[1].detach().numpy())
    # layer0 uses 8-bit weights (quantize_model.py:28): not part of the 4-bit export
    assert not any(f.startswith("layer0.0.") and f.endswith("codes_packed") for f in z.files)
    acts = [f for f in z.files if f.endswith(".x_min")]
    assert len(acts) >= 40
    # pack / unpack round trip on every nibble value
    allv = torch.arange(-8, 8, dtype=torch.int8).view(1, 16)
    assert torch.equal(evalio.unpack_int4(evalio.pack_int4(allv)), allv)


def test_eval_voc_pieces_known_answers():
    """tools/eval_voc.py: the pre-processing crop is the identity on a res x res image; the VOC07 11-point AP of
    perfect detections is 1, of detections on the wrong images 0, and a half-recall set gives 6/11."""
    import importlib.util
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("eval_voc", os.path.join(root, "tools", "eval_voc.py"))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)
    g = np.random.RandomState(0)
    img = g.randint(0, 256, (64, 64, 3)).astype(np.uint8)
    inp, meta = ev.pre_process(img, 64)
    want = (torch.from_numpy(img).permute(2, 0, 1).float() / 255.0 - torch.from_numpy(ev.MEAN).view(3, 1, 1)) \
        / torch.from_numpy(ev.STD).view(3, 1, 1)
    assert (inp[0] - want).abs().max().item() < 1e-5 and meta["s"] == 64.0 and tuple(meta["c"]) == (32.0, 32.0)
    # a 40 x 80 image: the long side maps onto the input, the short side is centred with zero padding
    inp, meta = ev.pre_process(g.randint(0, 256, (40, 80, 3)).astype(np.uint8), 80)
    assert meta["s"] == 80.0 and inp.shape == (1, 3, 80, 80)
    pad = (0.0 - torch.from_numpy(ev.MEAN)) / torch.from_numpy(ev.STD)
    assert (inp[0, :, 5, 40] - pad).abs().max().item() < 1e-5          # above the image: padding value
    gts = {i: (np.array([[10., 10., 50., 50.]]), np.array([False])) for i in range(10)}
    perfect = [(i, 0.9 - 0.01 * i, 10., 10., 50., 50.) for i in range(10)]
    assert abs(ev.voc_eval(perfect, gts) - 1.0) < 1e-9
    assert ev.voc_eval([(i + 100, 0.9, 10., 10., 50., 50.) for i in range(10)], gts) == 0.0
    half = perfect[:5]
    assert abs(ev.voc_eval(half, gts) - 6.0 / 11.0) < 1e-9              # recall 0.5 reached at precision 1


def test_voc_eval_reproduces_the_reference_evaluator():
    """tools/eval_voc.py::voc_eval_curve / voc_ap07 / voc_ap_area against the reference's OWN evaluator
    (tools/voc_eval_lib/datasets/voc_eval.py:31-209, run by tests/golden/make_golden.py::make_voc_eval over a synthetic
    VOC tree): rec, prec and both APs per class, exactly -- difficult objects (ignored when hit, not counted in npos),
    duplicate detections (second one a FP), partial recall, detections on images without the class, overlaps of
    0.495 / 0.515 around the threshold, a class without detections and one whose detections are all wrong."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("eval_voc", os.path.join(root, "tools", "eval_voc.py"))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)
    z = np.load(os.path.join(root, "tests", "golden", "voc_eval_ref.npz"))
    gt, dets = z["gt"], z["dets"]
    seen_nontrivial = 0
    for c, cname in enumerate(z["classes"]):
        cname = str(cname)
        gts = {}
        for im in range(int(z["n_images"])):       # every image is a key, as in the reference's class_recs
            rows = gt[(gt[:, 0] == im) & (gt[:, 1] == c)]
            gts[im] = (rows[:, 2:6].astype(np.float64), rows[:, 6].astype(bool))
        rows = [(int(r[0]), float(r[2]), *map(float, r[3:7])) for r in dets[dets[:, 1] == c]]
        rec, prec = ev.voc_eval_curve(rows, gts)
        assert rec.shape == z[cname + "_rec"].shape
        np.testing.assert_array_equal(rec, z[cname + "_rec"])
        np.testing.assert_array_equal(prec, z[cname + "_prec"])
        assert ev.voc_ap07(rec, prec) == float(z[cname + "_ap_07"])
        assert abs(ev.voc_ap_area(rec, prec) - float(z[cname + "_ap_area"])) < 1e-15
        assert ev.voc_eval(rows, gts) == float(z[cname + "_ap_07"])
        seen_nontrivial += 0.05 < float(z[cname + "_ap_07"]) < 0.95
    assert seen_nontrivial >= 3                      # the fixture pins non-trivial answers, not only AP 0 / 1


def test_pre_process_matches_affine_crop_known_answer():
    """pre_process against a closed form: a horizontal ramp image keeps its slope s/res per output pixel and the image
    centre lands on the output centre (get_affine_transform, lib/utils/image.py:30-55, rot = 0)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("eval_voc", os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "tools", "eval_voc.py"))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)
    h, w, res = 120, 200, 64
    ramp = np.broadcast_to(np.arange(w, dtype=np.float32)[None, :, None], (h, w, 3)).copy()
    img = np.clip(ramp, 0, 255).astype(np.uint8)
    inp, meta = ev.pre_process(img, res)
    assert inp.shape == (1, 3, res, res) and meta["s"] == 200.0 and list(meta["c"]) == [100.0, 60.0]
    raw = (inp[0] * torch.from_numpy(ev.STD).view(3, 1, 1) + torch.from_numpy(ev.MEAN).view(3, 1, 1)) * 255.0
    row = raw[0, res // 2]                      # a row through the image centre (inside the 120-pixel-high band)
    step = 200.0 / res
    want = torch.arange(res, dtype=torch.float32) * step + (100.0 - res / 2 * step)
    inside = (want >= 0.5) & (want <= w - 1.5)
    assert torch.allclose(row[inside], want[inside], atol=1e-2)
    # rows outside the image band (|v - res/2| * step > h/2) are the zero padding
    assert abs(float(raw[0, 0, res // 2])) < 1e-3


def test_transform_preds_and_ctdet_post_process_reproduce_the_reference():
    """evalio.transform_preds / ctdet_post_process against the REFERENCE's own lib/utils/image.py and
    lib/utils/post_process.py (tests/golden/post_process_ref.npz, make_golden.py::make_post_process: the reference's code
    with cv2.getAffineTransform given by its definition).  The reference builds the three point pairs in float32 and
    solves in float64; the closed form here agrees to 1e-4 px on coordinates up to 700 px."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "post_process_ref.npz"))
    for i in range(4):
        got = evalio.transform_preds(g["tp%d_pts" % i], g["tp%d_center" % i], g["tp%d_scale" % i].tolist()
                                     if g["tp%d_scale" % i].ndim else float(g["tp%d_scale" % i]),
                                     tuple(int(v) for v in g["tp%d_osize" % i]))
        assert np.abs(got - g["tp%d_out" % i]).max() < 1e-4, i
    res = evalio.ctdet_post_process(g["pp_dets"].copy(), g["pp_c"], g["pp_s"], 128, 128, 20)
    total = 0
    for b in range(3):
        assert sorted(res[b]) == list(range(1, 21))
        for j in range(1, 21):
            want = g["pp_out_%d_%d" % (b, j)]
            got = np.asarray(res[b][j], np.float32).reshape(-1, 5)
            assert got.shape == want.shape and (got.size == 0 or np.abs(got - want).max() < 2e-4), (b, j)
            total += want.shape[0]
    assert total == 150


def test_proxy_ap_bookkeeping_on_synthetic_detections():
    """tests/proxy_ap.py's host side (no GPU, no network): detections -> evaluator rows through the pipeline's own
    post-processing, the pseudo-ground-truth threshold, AP50 = 1 for a run against its own top detections, a lower AP for a
    run that lost half of them, 0 for boxes moved off their ground truth."""
    from tests import proxy_ap as A
    ev = A._eval_voc()
    g = torch.Generator().manual_seed(5)
    n, K = 6, 40
    cx, cy = torch.rand(n, K, generator=g) * 100 + 10, torch.rand(n, K, generator=g) * 100 + 10
    score = torch.rand(n, K, generator=g).sort(dim=1, descending=True).values
    cls = torch.randint(0, 20, (n, K), generator=g).float()
    dets = torch.stack([cx - 3, cy - 3, cx + 3, cy + 3, score, cls], 2)
    rows = A.Rows(512)
    rows.add(dets[:3], 0)
    rows.add(dets[3:], 3)
    assert rows.images == 6 and sum(len(v) for v in rows.rows.values()) == n * K
    x1 = [r for v in rows.rows.values() for r in v if r[0] == 0][0]
    assert 0 <= x1[2] < x1[4] <= 512 and abs((x1[4] - x1[2]) - 24.0) < 1e-3            # output pixels x 4
    gts, thr, count = A.ground_truth(rows, 8)
    assert count == 48 and thr == sorted(score.flatten().tolist(), reverse=True)[47]
    ap, ncls = A.ap50(ev, rows, gts)
    assert ap == pytest.approx(1.0) and ncls == sum(1 for c in gts if gts[c])
    half = A.Rows(512)
    half.add(dets[:, ::2], 0)
    ap_half, _ = A.ap50(ev, half, gts)
    assert 0.2 < ap_half < 0.9
    moved = A.Rows(512)
    far = dets.clone()
    far[:, :, :4] += 20.0
    moved.add(far, 0)
    assert A.ap50(ev, moved, gts)[0] == 0.0
    sub = rows.subset(2)
    assert sub.images == 2 and all(r[0] < 2 for v in sub.rows.values() for r in v)
