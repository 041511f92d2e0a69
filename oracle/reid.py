"""CPU ORACLE (test infrastructure, never a product path) -- DeepSORT appearance embedding, rows B3-B5.

torch-CPU fp32 functional restatement of networks/deepsort/deep/model.py:48-98 (`Net(reid=True)`)
and networks/deepsort/deep/feature_extractor.py:18-47 (`Extractor`), driven by a plain
state_dict with the reference's own parameter names.  Pinned by tests/golden/reid_forward.npz
(input/output produced by the reference's Net with seeded random parameters).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from .imageops import resize_linear_f32

IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)   # feature_extractor.py:21 (applied to BGR, Q3)
IMAGENET_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)
REID_SIZE = 50                                                      # feature_extractor.py:18 (w, h) = (50, 50)
BN_EPS = 1e-5                                                       # nn.BatchNorm2d default

# (name, c_in, c_out, downsample) in forward order; model.py:61-68 + make_layers :38-46
BLOCKS = [("layer1.0", 64, 64, False), ("layer1.1", 64, 64, False),
          ("layer2.0", 64, 128, True), ("layer2.1", 128, 128, False),
          ("layer3.0", 128, 256, True), ("layer3.1", 256, 256, False),
          ("layer4.0", 256, 512, True), ("layer4.1", 512, 512, False)]


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        training=False, eps=BN_EPS)


def _fold(w, b, sd, p):
    """Conv + eval-mode BatchNorm as one conv (float32), for the bf16 restatement below."""
    scale = sd[p + ".weight"] / torch.sqrt(sd[p + ".running_var"] + BN_EPS)
    b0 = torch.zeros(w.shape[0]) if b is None else b
    return w * scale[:, None, None, None], (b0 - sd[p + ".running_mean"]) * scale + sd[p + ".bias"]


def reid_forward_bf16(sd, x):
    """The same network in the product's VC_PREC_BF16 arithmetic: BatchNorm folded into the convs in float32, folded weights, the
    input and every activation rounded to bfloat16 once, fp32 accumulate + bias (+ shortcut) + ReLU before the rounding, pooling and
    the L2 norm in float32.  Not what the reference computes -- what a bf16 implementation of it has to compute."""
    r = lambda t: t.bfloat16().float()
    x = r(torch.as_tensor(x, dtype=torch.float32))
    with torch.no_grad():
        w, b = _fold(sd["conv.0.weight"], sd["conv.0.bias"], sd, "conv.1")
        y = r(F.relu(F.conv2d(x, r(w), b, stride=1, padding=1)))
        y = F.max_pool2d(y, 3, 2, padding=1)
        for name, cin, cout, down in BLOCKS:
            s = 2 if down else 1
            w, b = _fold(sd[name + ".conv1.weight"], None, sd, name + ".bn1")
            z = r(F.relu(F.conv2d(y, r(w), b, stride=s, padding=1)))
            if down or cin != cout:
                w, b = _fold(sd[name + ".downsample.0.weight"], None, sd, name + ".downsample.1")
                y = r(F.conv2d(y, r(w), b, stride=s))
            w, b = _fold(sd[name + ".conv2.weight"], None, sd, name + ".bn2")
            y = r(F.relu(F.conv2d(z, r(w), b, stride=1, padding=1) + y))
        y = F.avg_pool2d(y, (4, 4), 1)
        y = y.view(y.size(0), -1)
        y = y / y.norm(p=2, dim=1, keepdim=True)
    return y.numpy()


def reid_forward(sd, x):
    """x: (k,3,50,50) f32 -> (k,512) unit-norm f32.  model.py:83-98."""
    x = torch.as_tensor(x, dtype=torch.float32)
    with torch.no_grad():
        y = F.conv2d(x, sd["conv.0.weight"], sd["conv.0.bias"], stride=1, padding=1)
        y = F.relu(_bn(y, sd, "conv.1"))
        y = F.max_pool2d(y, 3, 2, padding=1)
        for name, cin, cout, down in BLOCKS:
            s = 2 if down else 1
            z = F.conv2d(y, sd[name + ".conv1.weight"], None, stride=s, padding=1)
            z = F.relu(_bn(z, sd, name + ".bn1"))
            z = F.conv2d(z, sd[name + ".conv2.weight"], None, stride=1, padding=1)
            z = _bn(z, sd, name + ".bn2")
            if down or cin != cout:                                   # model.py:17-27
                y = _bn(F.conv2d(y, sd[name + ".downsample.0.weight"], None, stride=s), sd, name + ".downsample.1")
            y = F.relu(y + z)
        y = F.avg_pool2d(y, (4, 4), 1)                                # model.py:70
        y = y.view(y.size(0), -1)
        y = y / y.norm(p=2, dim=1, keepdim=True)                      # model.py:93
    return y.numpy()


def preprocess_crops(crops):
    """feature_extractor.py:26-39: /255 -> resize 50x50 (float bilinear) -> CHW -> (x-mean)/std."""
    out = np.zeros((len(crops), 3, REID_SIZE, REID_SIZE), np.float32)
    for i, im in enumerate(crops):
        if im.shape[0] == 0 or im.shape[1] == 0:
            raise ValueError("empty crop (cv2.resize raises in the reference, quirk Q4)")
        r = resize_linear_f32(im.astype(np.float32) / np.float32(255.0), REID_SIZE, REID_SIZE)
        r = (r - IMAGENET_MEAN) / IMAGENET_STD
        out[i] = r.transpose(2, 0, 1)
    return out


def make_embedder(sd, bf16=False):
    sd = {k: torch.as_tensor(v) for k, v in sd.items()}

    def embed(crops):
        return (reid_forward_bf16 if bf16 else reid_forward)(sd, preprocess_crops(crops))
    return embed
