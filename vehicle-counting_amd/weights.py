"""Weights for the detect + track hot path: seeded synthetic parameters and checkpoint ingestion.

No real checkpoints exist in this environment (.MISSING_LARGE_BLOBS, no network), so
benchmarks and parity tests run on *seeded synthetic* parameters with the exact tensor names and
shapes of the real ones:

  * detector: ultralytics/yolov5 v6.0 state_dict names after `model.fuse()` -- `model.{i}.conv.weight/bias`,
    `model.{i}.cv{1,2,3}.conv.*`, `model.{i}.m.{j}.cv{1,2}.conv.*`, `model.24.m.{0,1,2}.weight/bias`
    (what networks/yolo.py:58 loads through torch.hub);
  * ReID: networks/deepsort/deep/model.py names -- `conv.0.weight`, `layer2.0.downsample.1.running_var`, ...
    (what feature_extractor.py:13-14 loads from ckpt.t7['net_dict']).

`fold_bn` turns an un-fused (conv, bn) pair into the (weight, bias) the HIP engine consumes, so a real
checkpoint converted to a flat {name: ndarray} dict (e.g. safetensors) can be fed the same way.
"""
from __future__ import annotations

import math

import numpy as np

YOLO_VARIANTS = {"yolov5s": (0.33, 0.50), "yolov5m": (0.67, 0.75), "yolov5l": (1.0, 1.0)}
YOLO_BN_EPS = 1e-3      # ultralytics initialises BatchNorm2d eps=1e-3
REID_BN_EPS = 1e-5


def fold_bn(w, b, gamma, beta, mean, var, eps):
    """Conv2d(+bias) followed by eval-mode BatchNorm2d -> one conv.  float32 in, float32 out."""
    w = np.asarray(w, np.float32)
    scale = (np.asarray(gamma, np.float32) / np.sqrt(np.asarray(var, np.float32) + np.float32(eps))).astype(np.float32)
    b0 = np.zeros(w.shape[0], np.float32) if b is None else np.asarray(b, np.float32)
    return (w * scale[:, None, None, None]).astype(np.float32), ((b0 - mean) * scale + beta).astype(np.float32)


def _c8(x):
    return int(math.ceil(x / 8) * 8)


def yolo_conv_table(variant="yolov5s", nc=80):
    """(name, c_in, c_out, k) for every conv of the v6.0 graph, forward order."""
    gd, gw = YOLO_VARIANTS[variant]
    ch = [_c8(c * gw) for c in (64, 128, 256, 512, 1024)]
    rep = [max(round(r * gd), 1) for r in (3, 6, 9, 3)]
    t = []

    def conv(name, ci, co, k):
        t.append((name, ci, co, k))

    def c3(i, ci, co, n):
        h = co // 2
        conv(f"model.{i}.cv1.conv", ci, h, 1)
        conv(f"model.{i}.cv2.conv", ci, h, 1)
        conv(f"model.{i}.cv3.conv", 2 * h, co, 1)
        for j in range(n):
            conv(f"model.{i}.m.{j}.cv1.conv", h, h, 1)
            conv(f"model.{i}.m.{j}.cv2.conv", h, h, 3)

    conv("model.0.conv", 3, ch[0], 6)
    conv("model.1.conv", ch[0], ch[1], 3); c3(2, ch[1], ch[1], rep[0])
    conv("model.3.conv", ch[1], ch[2], 3); c3(4, ch[2], ch[2], rep[1])
    conv("model.5.conv", ch[2], ch[3], 3); c3(6, ch[3], ch[3], rep[2])
    conv("model.7.conv", ch[3], ch[4], 3); c3(8, ch[4], ch[4], rep[3])
    conv("model.9.cv1.conv", ch[4], ch[4] // 2, 1); conv("model.9.cv2.conv", ch[4] * 2, ch[4], 1)
    conv("model.10.conv", ch[4], ch[3], 1); c3(13, ch[3] * 2, ch[3], rep[0])
    conv("model.14.conv", ch[3], ch[2], 1); c3(17, ch[2] * 2, ch[2], rep[0])
    conv("model.18.conv", ch[2], ch[2], 3); c3(20, ch[2] * 2, ch[3], rep[0])
    conv("model.21.conv", ch[3], ch[3], 3); c3(23, ch[3] * 2, ch[4], rep[0])
    for i, c in enumerate((ch[2], ch[3], ch[4])):
        conv(f"model.24.m.{i}", c, 3 * (nc + 5), 1)
    return t


def synth_yolo(variant="yolov5s", nc=80, seed=1702, det_scale=1.0, obj_shift=0.0, box_scale=0.5, fused=True):
    """Seeded synthetic, BN-folded detector parameters: {name+'.weight': OIHW f32, name+'.bias': f32}.
    fused=False returns the SAME parameters before the fold, under upstream's un-fused names (`...conv.weight` without bias,
    `...bn.{weight,bias,running_mean,running_var}`) -- what an ultralytics/yolov5 checkpoint holds before `model.fuse()`.

    Conv weights are variance-preserving for SiLU; BN statistics are non-trivial so the fold is
    exercised.  The Detect biases follow upstream's `_initialize_biases` prior
    (obj: log(8/(640/stride)^2), cls: log(0.6/(nc-0.99))); `det_scale`/`obj_shift` widen the logit
    distribution so that a controllable number of candidates pass conf > 0.25 on synthetic frames.
    """
    rng = np.random.default_rng(seed)
    sd = {}
    for name, ci, co, k in yolo_conv_table(variant, nc):
        fan_in = ci * k * k
        if name.startswith("model.24."):
            i = int(name.rsplit(".", 1)[1])
            stride = (8, 16, 32)[i]
            w = rng.standard_normal((co, ci, 1, 1), dtype=np.float32) * np.float32(det_scale / math.sqrt(fan_in))
            # box regressors stay near their prior (sigmoid ~ 0.5 -> boxes about one anchor in size, never degenerate)
            w.reshape(3, nc + 5, ci)[:, :4, :] *= np.float32(box_scale / det_scale)
            b = np.zeros((3, nc + 5), np.float32)
            b[:, 4] += math.log(8 / (640 / stride) ** 2) + obj_shift
            b[:, 5:] += math.log(0.6 / (nc - 0.99))
            # anchor-dependent jitter so the three anchors of a cell do not tie
            b += rng.standard_normal(b.shape, dtype=np.float32) * np.float32(0.05)
            sd[name + ".weight"], sd[name + ".bias"] = w, b.reshape(-1)
            continue
        gain = 2.6
        if ".m." in name and name.endswith("cv2.conv"):
            gain = 0.4          # residual branch of a Bottleneck: keep the shortcut sum from growing
        w = rng.standard_normal((co, ci, k, k), dtype=np.float32) * np.float32(math.sqrt(gain / fan_in))
        gamma = rng.uniform(0.9, 1.1, co).astype(np.float32)
        beta = (rng.standard_normal(co) * 0.1).astype(np.float32)
        mean = (rng.standard_normal(co) * 0.1).astype(np.float32)
        var = rng.uniform(0.8, 1.2, co).astype(np.float32)
        if fused:
            sd[name + ".weight"], sd[name + ".bias"] = fold_bn(w, None, gamma, beta, mean, var, YOLO_BN_EPS)
        else:
            bn = name[: -len("conv")] + "bn"
            sd[name + ".weight"] = w
            sd[bn + ".weight"], sd[bn + ".bias"], sd[bn + ".running_mean"], sd[bn + ".running_var"] = gamma, beta, mean, var
    return sd


REID_BLOCKS = [("layer1.0", 64, 64, False), ("layer1.1", 64, 64, False),
               ("layer2.0", 64, 128, True), ("layer2.1", 128, 128, False),
               ("layer3.0", 128, 256, True), ("layer3.1", 256, 256, False),
               ("layer4.0", 256, 512, True), ("layer4.1", 512, 512, False)]


def synth_reid(seed=1702):
    """Seeded synthetic *un-fused* ReID state_dict with the reference's parameter names (deep/model.py)."""
    rng = np.random.default_rng(seed + 7)
    sd = {}

    def bn(p, c):
        sd[p + ".weight"] = rng.uniform(0.9, 1.1, c).astype(np.float32)
        sd[p + ".bias"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
        sd[p + ".running_mean"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
        sd[p + ".running_var"] = rng.uniform(0.8, 1.2, c).astype(np.float32)

    def conv(p, co, ci, k, gain=2.0):
        sd[p + ".weight"] = (rng.standard_normal((co, ci, k, k)) * math.sqrt(gain / (ci * k * k))).astype(np.float32)

    conv("conv.0", 64, 3, 3)
    sd["conv.0.bias"] = (rng.standard_normal(64) * 0.1).astype(np.float32)
    bn("conv.1", 64)
    for name, ci, co, down in REID_BLOCKS:
        conv(name + ".conv1", co, ci, 3)
        bn(name + ".bn1", co)
        conv(name + ".conv2", co, co, 3, gain=1.0)
        bn(name + ".bn2", co)
        if down or ci != co:
            conv(name + ".downsample.0", co, ci, 1, gain=1.0)
            bn(name + ".downsample.1", co)
    return sd


def fold_reid(sd):
    """Un-fused ReID state_dict -> {layer: (weight OIHW f32, bias f32)} consumed by the HIP engine."""
    out = {}

    def f(conv, bnp, bias=None):
        return fold_bn(sd[conv + ".weight"], bias, sd[bnp + ".weight"], sd[bnp + ".bias"],
                       sd[bnp + ".running_mean"], sd[bnp + ".running_var"], REID_BN_EPS)

    out["conv"] = f("conv.0", "conv.1", sd["conv.0.bias"])
    for name, ci, co, down in REID_BLOCKS:
        out[name + ".conv1"] = f(name + ".conv1", name + ".bn1")
        out[name + ".conv2"] = f(name + ".conv2", name + ".bn2")
        if down or ci != co:
            out[name + ".downsample"] = f(name + ".downsample.0", name + ".downsample.1")
    return out


def fold_yolo_state_dict(sd, eps=YOLO_BN_EPS):
    """Un-fused ultralytics/yolov5 v6.0 state_dict (`...conv.weight` + `...bn.{weight,bias,running_mean,running_var}`)
    -> the fused {name+'.weight', name+'.bias'} dict the engine consumes (what `model.fuse()` does at hub load time).
    Entries that are already fused (conv has a bias, no bn) and the Detect head (`model.24.m.i.*`) pass through."""
    out = {}
    for k in sd:
        if not k.endswith(".weight") or np.asarray(sd[k]).ndim != 4:
            continue
        name = k[: -len(".weight")]
        w = np.asarray(sd[k], np.float32)
        bn = name[: -len("conv")] + "bn" if name.endswith(".conv") else None
        if bn is not None and bn + ".running_var" in sd:
            out[name + ".weight"], out[name + ".bias"] = fold_bn(w, sd.get(name + ".bias"), np.asarray(sd[bn + ".weight"], np.float32),
                                                                 np.asarray(sd[bn + ".bias"], np.float32),
                                                                 np.asarray(sd[bn + ".running_mean"], np.float32),
                                                                 np.asarray(sd[bn + ".running_var"], np.float32), eps)
        else:
            out[name + ".weight"] = w
            out[name + ".bias"] = np.asarray(sd.get(name + ".bias", np.zeros(w.shape[0])), np.float32)
    return out
