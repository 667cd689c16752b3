"""Checkpoint and evaluation I/O around the detector, host-side mirrors of the
reference's helpers with the same names and argument meaning:

    load_model / save_model          lib/models/model.py:35-100   (checkpoint dict {'epoch', 'state_dict'[, 'optimizer']})
    transform_preds                  lib/utils/image.py:14-19,22-55 (rot = 0: closed form, no cv2)
    ctdet_post_process               lib/utils/post_process.py:86-103
    post_process / merge_outputs     lib/detectors/ctdet.py:48-72
    convert_eval_format / save_results   lib/datasets/dataset/pascal.py:58-79   (results.json for tools/reval.py)
    export_w4                        packed 4-bit weight codes + per-channel scales + QuantAct ranges of a
                                     quantised model (what an integer-only deployment of the W4A8 model loads)
"""
import json
import os

import numpy as np
import torch


# ---- checkpoints ---------------------------------------------------------------------------------

def load_model(model, model_path, optimizer=None, resume=False, lr=None, lr_step=None, verbose=True):
    """Reference semantics: strips a DataParallel 'module.' prefix, keeps the model's own tensor where a
    shape differs or a key is missing, loads non-strictly; with resume=True restores the optimizer and
    decays lr by 0.1 for every passed lr_step."""
    say = print if verbose else (lambda *a, **k: None)
    start_epoch = 0
    checkpoint = torch.load(model_path, map_location=lambda storage, loc: storage)
    say("loaded {}, epoch {}".format(model_path, checkpoint["epoch"]))
    state_dict = {}
    for k, v in checkpoint["state_dict"].items():
        state_dict[k[7:] if k.startswith("module") and not k.startswith("module_list") else k] = v
    own = model.state_dict()
    for k in list(state_dict):
        if k in own:
            if state_dict[k].shape != own[k].shape:
                say("Skip loading parameter {}, required shape{}, loaded shape{}.".format(
                    k, own[k].shape, state_dict[k].shape))
                state_dict[k] = own[k]
        else:
            say("Drop parameter {}.".format(k))
    for k in own:
        if k not in state_dict:
            say("No param {}.".format(k))
            state_dict[k] = own[k]
    model.load_state_dict(state_dict, strict=False)
    if optimizer is not None and resume:
        if "optimizer" in checkpoint:
            optimizer.load_state_dict(checkpoint["optimizer"])
            start_epoch = checkpoint["epoch"]
            start_lr = lr
            for step in lr_step:
                if start_epoch >= step:
                    start_lr *= 0.1
            for group in optimizer.param_groups:
                group["lr"] = start_lr
            say("Resumed optimizer with start lr", start_lr)
        else:
            say("No optimizer parameters in checkpoint.")
    return (model, optimizer, start_epoch) if optimizer is not None else model


def save_model(path, epoch, model, optimizer=None):
    sd = model.module.state_dict() if isinstance(model, torch.nn.DataParallel) else model.state_dict()
    data = {"epoch": epoch, "state_dict": sd}
    if optimizer is not None:
        data["optimizer"] = optimizer.state_dict()
    torch.save(data, path)


# ---- detections -> image coordinates -> results.json ---------------------------------------------

def transform_preds(coords, center, scale, output_size):
    """Inverse of the un-rotated crop transform: the reference builds it from three point pairs with
    cv2.getAffineTransform; for rot = 0 that affine map is
        p_img = center + (p_out - output_size / 2) * (scale_w / output_w)      (isotropic)."""
    coords = np.asarray(coords, dtype=np.float64)
    if not isinstance(scale, (np.ndarray, list, tuple)):
        scale = [scale, scale]
    k = float(scale[0]) / float(output_size[0])
    out = np.zeros(coords.shape)
    out[:, 0] = (coords[:, 0] - output_size[0] * 0.5) * k + center[0]
    out[:, 1] = (coords[:, 1] - output_size[1] * 0.5) * k + center[1]
    return out


def ctdet_post_process(dets, c, s, h, w, num_classes):
    """dets [B, K, 6] (x1, y1, x2, y2, score, class) in output-map pixels -> per image a 1-based class
    dict of [x1, y1, x2, y2, score] lists in image pixels."""
    ret = []
    for i in range(dets.shape[0]):
        top_preds = {}
        dets[i, :, :2] = transform_preds(dets[i, :, 0:2], c[i], s[i], (w, h))
        dets[i, :, 2:4] = transform_preds(dets[i, :, 2:4], c[i], s[i], (w, h))
        classes = dets[i, :, -1]
        for j in range(num_classes):
            inds = classes == j
            top_preds[j + 1] = np.concatenate([dets[i, inds, :4].astype(np.float32),
                                               dets[i, inds, 4:5].astype(np.float32)], axis=1).tolist()
        ret.append(top_preds)
    return ret


def post_process(dets, meta, num_classes, scale=1):
    """CtdetDetector.post_process: dets tensor [1, K, 6] of ONE image (or a flip pair merged by process)."""
    dets = dets.detach().cpu().numpy()
    dets = dets.reshape(1, -1, dets.shape[2])
    dets = ctdet_post_process(dets.copy(), [meta["c"]], [meta["s"]], meta["out_height"], meta["out_width"],
                              num_classes)
    for j in range(1, num_classes + 1):
        dets[0][j] = np.array(dets[0][j], dtype=np.float32).reshape(-1, 5)
        dets[0][j][:, :4] /= scale
    return dets[0]


def merge_outputs(detections, num_classes, max_per_image=100, nms=False):
    """CtdetDetector.merge_outputs for a single test scale (soft-NMS, used only with multi-scale testing or
    --nms, is outside this port)."""
    if nms or len(detections) > 1:
        raise NotImplementedError("soft-NMS (multi-scale test / --nms) is not part of this port")
    results = {j: np.concatenate([d[j] for d in detections], axis=0).astype(np.float32)
               for j in range(1, num_classes + 1)}
    scores = np.hstack([results[j][:, 4] for j in range(1, num_classes + 1)])
    if len(scores) > max_per_image:
        kth = len(scores) - max_per_image
        thresh = np.partition(scores, kth)[kth]
        for j in range(1, num_classes + 1):
            results[j] = results[j][results[j][:, 4] >= thresh]
    return results


def convert_eval_format(all_bboxes, images, num_classes):
    """pascal.py:58-68: detections[class][image index] = list of [x1, y1, x2, y2, score]."""
    detections = [[[] for __ in range(len(images))] for _ in range(num_classes + 1)]
    for i, img_id in enumerate(images):
        for j in range(1, num_classes + 1):
            v = all_bboxes[img_id][j]
            detections[j][i] = v.tolist() if isinstance(v, np.ndarray) else v
    return detections


def save_results(results, images, num_classes, save_dir):
    path = os.path.join(save_dir, "results.json")
    with open(path, "w") as f:
        json.dump(convert_eval_format(results, images, num_classes), f)
    return path


# ---- integer export --------------------------------------------------------------------------------

def pack_int4(codes):
    """int8 codes in [-8, 7], last dim even -> uint8 with two two's-complement nibbles per byte (low first)."""
    c = codes.to(torch.int16) & 0xF
    return (c[..., 0::2] | (c[..., 1::2] << 4)).to(torch.uint8)


def unpack_int4(packed):
    p = packed.to(torch.int16)
    lo, hi = p & 0xF, (p >> 4) & 0xF
    out = torch.stack([lo, hi], dim=-1).reshape(*packed.shape[:-1], packed.shape[-1] * 2)
    return torch.where(out > 7, out - 16, out).to(torch.int8)


def export_w4(model, path=None):
    """Every per-channel symmetric <= 4-bit conv of a quantised model as packed codes + scales (the
    fake-quantised weight is codes / scale, bit-exactly), plus the folded biases and every QuantAct range.
    Returns the dict; written with numpy.savez when `path` is given."""
    from .portable_quantizer.quant_modules import QuantAct, QuantBnConv2d, Quant_Conv2d, QuantDeformConv2d
    out = {}
    with torch.no_grad():
        for name, m in model.named_modules():
            if isinstance(m, QuantAct):
                out[name + ".x_min"] = m.x_min.cpu().numpy()
                out[name + ".x_max"] = m.x_max.cpu().numpy()
                continue
            if isinstance(m, QuantBnConv2d):
                sf = m.bn.weight / torch.sqrt(m.bn.running_var + m.bn.eps)
                w = m.conv.weight * sf.reshape(-1, 1, 1, 1)
                bias = m.folded()[1]
            elif isinstance(m, (Quant_Conv2d, QuantDeformConv2d)):
                w, bias = m.weight, m.bias
            else:
                continue
            if m.full_precision_flag or not m.per_channel or m.quant_mode != "symmetric" or m.weight_bit > 4:
                continue
            co = w.shape[0]
            flat = w.reshape(co, -1)
            codes, scale, _ = m._int8_codes(flat.reshape(co, -1, 1, 1))
            codes = codes[:, :flat.shape[1]]
            if codes.shape[1] % 2:
                codes = torch.cat([codes, torch.zeros(co, 1, dtype=torch.int8, device=codes.device)], 1)
            out[name + ".codes_packed"] = pack_int4(codes).cpu().numpy()
            out[name + ".scale"] = scale.cpu().numpy()
            out[name + ".shape"] = np.array(w.shape)
            if bias is not None:
                out[name + ".bias"] = bias.detach().cpu().numpy()
    if path is not None:
        np.savez_compressed(path, **out)
    return out
