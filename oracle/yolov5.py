"""CPU ORACLE (test infrastructure, never a product path) -- detect side, rows A2-A9 of SURVEY.md section 8.

PARITY UNPINNED for A5-A9: the detector is `torch.hub.load('ultralytics/yolov5', 'custom', ...)`
(/root/reference/networks/yolo.py:58), an un-vendored, un-pinned third-party module whose weights
are pinned to release v6.0 (/root/reference/utilities/utils.py:204-209).  Its source is not under
/root/reference and cannot be fetched (no network), and the reference holds no tests or golden
vectors for it.  This file restates the published v6.0 algorithm:

  models/yolov5{s,m,l}.yaml + models/yolo.py::parse_model   -> build_graph / forward
  models/common.py::Conv (Conv2d+BN fused at load, SiLU), C3, Bottleneck, SPPF
  models/yolo.py::Detect.forward (inference branch)          -> decode inside forward()
  models/common.py::AutoShape.forward                        -> autoshape_detect
  utils/general.py::non_max_suppression, xywh2xyxy, scale_coords, clip_coords, make_divisible
  torchvision.ops.nms (CPU kernel: stable descending sort, greedy, IoU > thr suppressed)

Structural cross-check (tests/test_oracle_yolo.py): the layer table reproduces upstream's
published 7.2 M parameters / 16.5 GFLOPs for yolov5s at 640x640.

Parity anchored on the reference's own call sites: networks/yolo.py:58-99 (thresholds,
xyxy -> xywh marshal through pandas/JSON) and modules/detect.py:30-60.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from .imageops import autoshape_size, letterbox, letterbox_geometry

VARIANTS = {"yolov5s": (0.33, 0.50), "yolov5m": (0.67, 0.75), "yolov5l": (1.0, 1.0)}   # depth, width multiples
ANCHORS = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]
STRIDES = [8, 16, 32]
MAX_WH = 4096       # v6.0 non_max_suppression class offset
MAX_NMS = 30000


def _div8(x):
    return int(math.ceil(x / 8) * 8)


def build_graph(variant="yolov5s"):
    """Expand the v6.0 yaml: list of (index, kind, from, args) with resolved channels / repeats."""
    gd, gw = VARIANTS[variant]
    c = lambda ch: _div8(ch * gw)
    n = lambda r: max(round(r * gd), 1)
    return [
        (0, "conv", -1, (3, c(64), 6, 2, 2)),
        (1, "conv", -1, (c(64), c(128), 3, 2, 1)),
        (2, "c3", -1, (c(128), c(128), n(3), True)),
        (3, "conv", -1, (c(128), c(256), 3, 2, 1)),
        (4, "c3", -1, (c(256), c(256), n(6), True)),
        (5, "conv", -1, (c(256), c(512), 3, 2, 1)),
        (6, "c3", -1, (c(512), c(512), n(9), True)),
        (7, "conv", -1, (c(512), c(1024), 3, 2, 1)),
        (8, "c3", -1, (c(1024), c(1024), n(3), True)),
        (9, "sppf", -1, (c(1024), c(1024))),
        (10, "conv", -1, (c(1024), c(512), 1, 1, 0)),
        (11, "up", -1, ()),
        (12, "cat", (-1, 6), ()),
        (13, "c3", -1, (c(512) * 2, c(512), n(3), False)),
        (14, "conv", -1, (c(512), c(256), 1, 1, 0)),
        (15, "up", -1, ()),
        (16, "cat", (-1, 4), ()),
        (17, "c3", -1, (c(256) * 2, c(256), n(3), False)),
        (18, "conv", -1, (c(256), c(256), 3, 2, 1)),
        (19, "cat", (-1, 14), ()),
        (20, "c3", -1, (c(256) * 2, c(512), n(3), False)),
        (21, "conv", -1, (c(512), c(512), 3, 2, 1)),
        (22, "cat", (-1, 10), ()),
        (23, "c3", -1, (c(512) * 2, c(1024), n(3), False)),
        (24, "detect", (17, 20, 23), (c(256), c(512), c(1024))),
    ]


def conv_specs(variant="yolov5s", nc=80):
    """Every convolution as (weight_name, c_in, c_out, k, stride, pad, has_act) in forward order."""
    out = []
    for idx, kind, _, a in build_graph(variant):
        p = f"model.{idx}"
        if kind == "conv":
            out.append((p + ".conv", a[0], a[1], a[2], a[3], a[4], True))
        elif kind == "c3":
            c1, c2, rep, _ = a
            h = c2 // 2
            out.append((p + ".cv1.conv", c1, h, 1, 1, 0, True))
            for j in range(rep):
                out.append((f"{p}.m.{j}.cv1.conv", h, h, 1, 1, 0, True))
                out.append((f"{p}.m.{j}.cv2.conv", h, h, 3, 1, 1, True))
            out.append((p + ".cv2.conv", c1, h, 1, 1, 0, True))
            out.append((p + ".cv3.conv", 2 * h, c2, 1, 1, 0, True))
        elif kind == "sppf":
            c1, c2 = a
            out.append((p + ".cv1.conv", c1, c1 // 2, 1, 1, 0, True))
            out.append((p + ".cv2.conv", c1 * 2, c2, 1, 1, 0, True))
        elif kind == "detect":
            for i, ch in enumerate(a):
                out.append((f"{p}.m.{i}", ch, 3 * (nc + 5), 1, 1, 0, False))
    return out


def _r16(t):
    """Round to bfloat16 (nearest even) and back: what one bf16 store + load does to an fp32 value."""
    return t.bfloat16().float()


def _cba(x, sd, name, s, p, res=None, bf16=False):
    """Conv (BN already folded into weight/bias) + SiLU; models/common.py::Conv.forward_fuse.  `res`: a Bottleneck's shortcut, added
    after the activation.  bf16=True restates the arithmetic of the product's VC_PREC_BF16 mode: bf16 operands (the caller hands in
    rounded activations and weights), fp32 accumulate + bias + SiLU (+ shortcut), ONE rounding to bf16 on store."""
    y = F.silu(F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=s, padding=p))
    if res is not None:
        y = y + res
    return _r16(y) if bf16 else y


def _c3(x, sd, p, rep, shortcut, bf16=False):
    y = _cba(x, sd, p + ".cv1.conv", 1, 0, bf16=bf16)
    for j in range(rep):
        y = _cba(_cba(y, sd, f"{p}.m.{j}.cv1.conv", 1, 0, bf16=bf16), sd, f"{p}.m.{j}.cv2.conv", 1, 1, res=y if shortcut else None, bf16=bf16)
    return _cba(torch.cat((y, _cba(x, sd, p + ".cv2.conv", 1, 0, bf16=bf16)), 1), sd, p + ".cv3.conv", 1, 0, bf16=bf16)


def forward(sd, x, variant="yolov5s", nc=80, return_layers=False, bf16=False):
    """x: (B,3,H,W) f32 in [0,1] -> (B, n_candidates, 5+nc) decoded predictions (Detect inference output).
    bf16=True: a restatement of the SAME network in the product's benchmarked precision (every weight, the input and every
    activation -- Detect logits included -- rounded to bfloat16 once, fp32 accumulation): not what the reference computes, but what
    a bf16 implementation of it has to compute, so that the HIP bf16 path can be held to a tight tolerance (accumulation order only)."""
    sd = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in sd.items()}
    x = torch.as_tensor(x, dtype=torch.float32)
    if bf16:
        sd = {k: (_r16(v) if k.endswith(".weight") else v) for k, v in sd.items()}
        x = _r16(x)
    ys = []
    no = nc + 5
    with torch.no_grad():
        for idx, kind, frm, a in build_graph(variant):
            p = f"model.{idx}"
            if kind == "conv":
                x = _cba(x, sd, p + ".conv", a[3], a[4], bf16=bf16)
            elif kind == "c3":
                x = _c3(x, sd, p, a[2], a[3], bf16=bf16)
            elif kind == "sppf":
                x = _cba(x, sd, p + ".cv1.conv", 1, 0, bf16=bf16)
                y1 = F.max_pool2d(x, 5, 1, 2)
                y2 = F.max_pool2d(y1, 5, 1, 2)
                x = _cba(torch.cat((x, y1, y2, F.max_pool2d(y2, 5, 1, 2)), 1), sd, p + ".cv2.conv", 1, 0, bf16=bf16)
            elif kind == "up":
                x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            elif kind == "cat":
                x = torch.cat((x, ys[frm[1]]), 1)
            elif kind == "detect":
                z = []
                raw = []
                for i, src in enumerate(frm):
                    t = F.conv2d(ys[src], sd[f"{p}.m.{i}.weight"], sd[f"{p}.m.{i}.bias"])
                    if bf16:
                        t = _r16(t)
                    raw.append(t)
                    bs, _, ny, nx = t.shape
                    t = t.view(bs, 3, no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
                    yv, xv = torch.meshgrid(torch.arange(ny), torch.arange(nx), indexing="ij")
                    grid = torch.stack((xv, yv), 2).expand(1, 3, ny, nx, 2).float()
                    ag = (torch.tensor(ANCHORS[i], dtype=torch.float32).view(3, 2)).view(1, 3, 1, 1, 2).expand(1, 3, ny, nx, 2)
                    y = t.sigmoid()
                    y[..., 0:2] = (y[..., 0:2] * 2.0 - 0.5 + grid) * STRIDES[i]
                    y[..., 2:4] = (y[..., 2:4] * 2) ** 2 * ag
                    z.append(y.view(bs, -1, no))
                x = torch.cat(z, 1)
                if return_layers:
                    return x, ys, raw
                return x
            ys.append(x)
    raise AssertionError("graph has no detect layer")


# ------------------------------------------------------------------------------- A8 NMS
def box_iou_greedy_nms(boxes, scores, thr):
    """torchvision.ops.nms CPU semantics in float32: stable descending sort, suppress IoU > thr."""
    boxes = np.asarray(boxes, dtype=np.float32)
    n = len(boxes)
    if n == 0:
        return np.zeros((0,), np.int64)
    order = np.argsort(-np.asarray(scores, dtype=np.float32), kind="stable")
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    area = ((x2 - x1) * (y2 - y1)).astype(np.float32)
    dead = np.zeros(n, bool)
    keep = []
    thr = np.float32(thr)
    for a in range(n):
        i = order[a]
        if dead[i]:
            continue
        keep.append(i)
        rest = order[a + 1:]
        w = np.maximum(np.float32(0), np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest])).astype(np.float32)
        h = np.maximum(np.float32(0), np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest])).astype(np.float32)
        inter = (w * h).astype(np.float32)
        ovr = inter / ((area[i] + area[rest]).astype(np.float32) - inter).astype(np.float32)
        dead[rest[ovr > thr]] = True
    return np.asarray(keep, dtype=np.int64)


def non_max_suppression(pred, conf_thres=0.25, iou_thres=0.45, classes=None, max_det=300):
    """utils/general.py::non_max_suppression (v6.0), multi_label=False, agnostic=False, float32 throughout.

    pred: (B, n, 5+nc) -> list of (m, 6) float32 [x1,y1,x2,y2,conf,cls].
    """
    out = []
    pred = np.asarray(pred, dtype=np.float32)
    for x in pred:
        x = x[x[:, 4] > np.float32(conf_thres)]
        if not len(x):
            out.append(np.zeros((0, 6), np.float32))
            continue
        cls = (x[:, 5:] * x[:, 4:5]).astype(np.float32)
        half_w, half_h = x[:, 2] / np.float32(2), x[:, 3] / np.float32(2)
        box = np.stack((x[:, 0] - half_w, x[:, 1] - half_h, x[:, 0] + half_w, x[:, 1] + half_h), 1).astype(np.float32)
        j = cls.argmax(1)
        conf = cls[np.arange(len(cls)), j]
        keep = conf > np.float32(conf_thres)
        det = np.concatenate((box, conf[:, None], j[:, None].astype(np.float32)), 1)[keep]
        if classes is not None:
            det = det[np.isin(det[:, 5], np.asarray(classes, dtype=np.float32))]
        if not len(det):
            out.append(np.zeros((0, 6), np.float32))
            continue
        if len(det) > MAX_NMS:
            det = det[np.argsort(-det[:, 4], kind="stable")[:MAX_NMS]]
        off = (det[:, 5:6] * np.float32(MAX_WH)).astype(np.float32)
        k = box_iou_greedy_nms((det[:, :4] + off).astype(np.float32), det[:, 4], iou_thres)[:max_det]
        out.append(det[k].astype(np.float32))
    return out


def scale_coords(shape1, boxes, shape0):
    """utils/general.py::scale_coords + clip_coords (v6.0) in float32."""
    b = np.array(boxes, dtype=np.float32, copy=True)
    gain = min(shape1[0] / shape0[0], shape1[1] / shape0[1])
    padw, padh = (shape1[1] - shape0[1] * gain) / 2, (shape1[0] - shape0[0] * gain) / 2
    b[:, [0, 2]] -= np.float32(padw)
    b[:, [1, 3]] -= np.float32(padh)
    b[:, :4] /= np.float32(gain)
    b[:, [0, 2]] = b[:, [0, 2]].clip(0, shape0[1])
    b[:, [1, 3]] = b[:, [1, 3]].clip(0, shape0[0])
    return b


def preprocess(imgs_rgb, size=640):
    """AutoShape.forward preprocessing: common stride-32 shape, letterbox(114), BCHW, /255."""
    shape0 = [im.shape[:2] for im in imgs_rgb]
    shape1 = autoshape_size(shape0, size)
    x = np.stack([letterbox(im, shape1[0], shape1[1]) for im in imgs_rgb], 0)
    x = np.ascontiguousarray(x.transpose(0, 3, 1, 2)).astype(np.float32) / np.float32(255)
    return x, shape0, shape1


def autoshape_detect(sd, imgs_rgb, variant="yolov5s", nc=80, size=640, conf=0.25, iou=0.45, classes=None,
                     max_det=300, bf16=False):
    """AutoShape.forward end to end: list of HxWx3 uint8 RGB -> list of (m,6) float32 xyxy in source pixels."""
    x, shape0, shape1 = preprocess(imgs_rgb, size)
    pred = forward(sd, x, variant, nc, bf16=bf16).numpy()
    dets = non_max_suppression(pred, conf, iou, classes, max_det)
    return [np.concatenate((scale_coords(shape1, d[:, :4], s0), d[:, 4:]), 1) if len(d) else d
            for d, s0 in zip(dets, shape0)]


def marshal_like_reference(det):
    """networks/yolo.py:72-97: pandas xyxy -> to_json (10 decimals, Q9) -> xywh top-left float64."""
    if len(det) == 0:
        return {"bboxes": np.array(()), "classes": np.array(()), "scores": np.array(())}
    d = np.round(det.astype(np.float64), 10)
    boxes = np.stack((d[:, 0], d[:, 1], d[:, 2] - d[:, 0], d[:, 3] - d[:, 1]), 1)
    return {"bboxes": boxes, "classes": det[:, 5].astype(np.int64), "scores": d[:, 4]}
