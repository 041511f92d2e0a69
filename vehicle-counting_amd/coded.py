"""A WELL-CONDITIONED seeded detector for the end-to-end parity tests in the benchmarked precisions.

The seeded random head of `weights.synth_yolo` puts a continuum of confidences around `conf_thres`: bf16 (let alone fp8) rounding
flips marginal detections, a flipped detection renumbers every later track, and the only end-to-end statement left is a hit rate.
A trained detector is not like that -- its object logits sit far from the threshold -- but no trained weights exist here (SURVEY.md:
.MISSING_LARGE_BLOBS).  This module builds the synthetic stand-in: the same seeded random YOLOv5 (every conv keeps its random
weights), plus a hand-wired CARRIER path of a few channels per layer and synthetic frames whose objects carry a machine-readable
32 x 16 pixel code (a "plate"):

  * frames (`coded_frames`): `synth.synth_frames` scenes (textures' green channel limited to 40..100) + per object a plate in the
    top half of the stride-32 cell that holds the object's centre: 4 x 8 sub-blocks of 4 x 4 network pixels, each pure green (bit 1)
    or black (bit 0) -- a presence bit, three class flags, a 4-bit per-object rank (keeps the confidences of different objects
    apart, so that NMS between overlapping objects never sees a near-tie), and the box as YOLOv5's own regression logits
    (tx, ty 5 bits, tw, th 7 bits);
  * weights (`coded_yolo`): channel 0 of the stem answers "green minus (red + blue) / 2" of its 2 x 2 pixel block with 16 (bit 1) or
    ~0 (anything else), the stride-2 convs pass block means / select quadrants so that 32 channels of the stride-32 map hold the 32
    bits of the cell's plate, every conv on the way (C3.cv2 -> C3.cv3, SPPF.cv1 -> SPPF.cv2, layer 10, the concat into layer 23)
    passes them through, and `Detect.m[2]` anchor 0 reads them: objectness +6 .. +9.75 on a plate and -6 elsewhere, class logits
    +6 / -9, the box logits as binary fractions.  The other anchors and levels keep random weights with their objectness far below
    the threshold.  Every logit also carries a small random term over ALL other channels of the head's input.

Why bits and not analog levels: a level of 16 is exact in fp32, bf16 and e4m3, SiLU(16) rounds back to 16 in the narrow types and
loses 2e-6 per layer in fp32, so the 18 convolutions between the pixels and the head hand the head the SAME numbers in every precision
and the decoded boxes agree to ~0.02 px (the random term).  An analog variant (levels 4 .. 8 carrying the regression logits
directly) was built first: bf16 rounding of the carried levels moved the decoded boxes by 2 - 5 px, DeepSORT's own NMS / IoU gates
then flipped for overlapping objects and one of twelve tracks was confirmed a frame late -- a statement about the tracker's
sensitivity to box noise, not about the kernels.

What the construction does and does not test: detections no longer depend on which side of a threshold a rounding error lands, so
the whole path (detector -> NMS -> crops -> ReID -> DeepSORT -> counting) can be compared row by row between precisions.  The
numerical accuracy of the random channels is covered by the per-layer ladder tests (tests/test_gpu_nets.py), not by this one.
Only `tests/` and `bench.py` use this module.
"""
from __future__ import annotations

import math

import numpy as np

from .synth import _objects, synth_tracks
from .weights import YOLO_VARIANTS, _c8, synth_yolo

CELL = 32                  # stride of the level that detects (P5)
SB = 4                     # sub-block edge in network pixels (one output pixel of layer 1)
ANCHOR = (116.0, 90.0)     # P5 anchor 0 (models/yolov5s.yaml)
V = 16.0                   # a set bit at every layer: exact in fp32 / bf16 / e4m3, SiLU(16) = 16 (1 - 1.1e-7)
STEM_GAIN, STEM_BIAS = 40.0, -24.0      # stem pre-activation = 40 * (G - (R + B) / 2) / 255 - 24 over a 2 x 2 block: 16 on a set bit, <= -8.4 elsewhere
T_RANGE = {"tx": (-1.2, 1.2), "ty": (-1.2, 1.2), "tw": (-1.3, 0.9), "th": (-1.3, 0.9)}     # regression logits the fields span
FIELDS = (("on", 1), ("cls", 3), ("rank", 4), ("tx", 5), ("ty", 5), ("tw", 7), ("th", 7))   # bits of the plate in raster order of its 4 x 8 sub-blocks
NBITS = sum(n for _, n in FIELDS)
assert NBITS == 32


def field_slices():
    out, k = {}, 0
    for name, n in FIELDS:
        out[name] = (k, n)
        k += n
    return out


def _quant(t, name):
    """(integer code, decoded logit) of a regression logit in field `name`: t ~ lo + (hi - lo) (q + 0.5) / 2^n."""
    lo, hi = T_RANGE[name]
    n = field_slices()[name][1]
    q = int(np.clip(math.floor((t - lo) / (hi - lo) * (1 << n)), 0, (1 << n) - 1))
    return q, lo + (hi - lo) * (q + 0.5) / (1 << n)


def letterbox_geometry(h0, w0, size=640, stride=32):
    """AutoShape + letterbox(auto=False) of ONE frame geometry: (net_h, net_w, gain, left, top) (oracle/imageops.py restated)."""
    g = size / max(h0, w0)
    net_h, net_w = (int(math.ceil(x * g / stride) * stride) for x in (h0, w0))
    r = min(net_h / h0, net_w / w0)
    uw, uh = int(round(w0 * r)), int(round(h0 * r))
    return net_h, net_w, r, int(round((net_w - uw) / 2 - 0.1)), int(round((net_h - uh) / 2 - 0.1))


def coded_frames(n_frames, H, W, n_obj=12, seed=1702, size=640, bounce=True):
    """(frames (n, H, W, 3) u8 BGR, truth): `truth[t]` lists, per plate painted in frame t, (object, label, x1, y1, x2, y2) -- the box
    the plate's bits decode to, in source pixels (the drawn rectangle to the fields' resolution: ~1 px in position, ~1.5 px in size).
    Frames must reach the network unscaled or by an exact 2:1 reduction (640 x 640, 1024 x 1024, 1280 x 1280 at their own size;
    1280 x 720 at size 640), so that sub-blocks stay uniform through the letterbox resize."""
    assert n_obj <= 16
    net_h, net_w, gain, left, top = letterbox_geometry(H, W, size)
    inv = int(round(1.0 / gain))
    assert abs(gain * inv - 1.0) < 1e-12 and inv in (1, 2), f"coded_frames: {H}x{W} at size {size} is not an exact 1:1 or 2:1 letterbox"
    rng, pos, vel, wh, tex, label = _objects(n_obj, H, W, seed)
    tex = tex.copy()
    tex[..., 1] = 40 + (tex[..., 1].astype(np.int32) * 60 // 255).astype(np.uint8)        # green 40..100: never mistaken for a set bit
    yy, xx = np.mgrid[0:H, 0:W]
    bg = (96 + 40 * np.sin(xx / 97.0) * np.cos(yy / 61.0))[..., None] + np.array([0, 8, 16])
    tracks = synth_tracks(n_frames, H, W, n_obj, seed, bounce)
    frames = np.empty((n_frames, H, W, 3), np.uint8)
    truth = []
    fs = field_slices()
    logit = lambda p: math.log(p / (1.0 - p))
    sig = lambda t: 1.0 / (1.0 + math.exp(-t))
    for t in range(n_frames):
        f = bg.copy()
        rects = []
        for i, b in enumerate(tracks[t][0]):
            x1, y1, w, h = (int(round(v)) for v in b)
            x2, y2 = min(x1 + w, W), min(y1 + h, H)
            if x2 <= x1 or y2 <= y1:
                continue
            ty_ = (np.arange(y1, y2) - y1) * 16 // max(h, 1)
            tx_ = (np.arange(x1, x2) - x1) * 16 // max(w, 1)
            f[y1:y2, x1:x2] = tex[i][np.clip(ty_, 0, 15)][:, np.clip(tx_, 0, 15)]
            rects.append((i, x1, y1, x2, y2))
        f = np.clip(f, 0, 255).astype(np.uint8)
        cells, row = {}, []
        for i, x1, y1, x2, y2 in rects:                                # plates go on top of every texture; a later object takes a shared cell
            cx, cy = (x1 + x2) / 2.0 * gain + left, (y1 + y2) / 2.0 * gain + top
            cells[(int(cx // CELL), int(cy // CELL))] = (i, x1, y1, x2, y2, cx, cy)
        for (gx, gy), (i, x1, y1, x2, y2, cx, cy) in cells.items():
            px0, py0 = (gx * CELL - left) * inv, (gy * CELL - top) * inv          # the plate's corner in source pixels
            ew, eh = 8 * SB * inv, 4 * SB * inv
            if px0 < 0 or py0 < 0 or px0 + ew > W or py0 + eh > H:
                continue                                               # a plate is painted whole or not at all
            want = {"tx": logit((cx / CELL - gx + 0.5) / 2), "ty": logit((cy / CELL - gy + 0.5) / 2),
                    "tw": logit(min(math.sqrt((x2 - x1) * gain / ANCHOR[0]) / 2, 0.99)), "th": logit(min(math.sqrt((y2 - y1) * gain / ANCHOR[1]) / 2, 0.99))}
            bits = np.zeros(NBITS, np.uint8)
            bits[fs["on"][0]] = 1
            bits[fs["cls"][0] + int(label[i])] = 1
            code = {"rank": i}
            dec = {}
            for k in ("tx", "ty", "tw", "th"):
                code[k], dec[k] = _quant(want[k], k)
            for k in ("rank", "tx", "ty", "tw", "th"):
                k0, n = fs[k]
                for j in range(n):
                    bits[k0 + j] = (code[k] >> (n - 1 - j)) & 1       # most significant bit first
            plate = np.zeros((4, 8, 3), np.uint8)
            plate[..., 1] = bits.reshape(4, 8) * 255
            f[py0:py0 + eh, px0:px0 + ew] = np.repeat(np.repeat(plate, SB * inv, 0), SB * inv, 1)
            # what Detect decodes from these bits (models/yolo.py::Detect.forward), mapped back through scale_coords
            dcx, dcy = (2 * sig(dec["tx"]) - 0.5 + gx) * CELL, (2 * sig(dec["ty"]) - 0.5 + gy) * CELL
            dw, dh = (2 * sig(dec["tw"])) ** 2 * ANCHOR[0], (2 * sig(dec["th"])) ** 2 * ANCHOR[1]
            row.append((i, int(label[i]), (dcx - dw / 2 - left) / gain, (dcy - dh / 2 - top) / gain, (dcx + dw / 2 - left) / gain, (dcy + dh / 2 - top) / gain))
        frames[t] = f
        truth.append(row)
    return frames, truth


def coded_yolo(variant="yolov5s", nc=80, seed=1702, clutter=0.3):
    """`weights.synth_yolo` with the carrier path wired in (module docstring).  Folded names ({name}.weight OIHW f32, {name}.bias)."""
    assert nc >= 3
    sd = synth_yolo(variant, nc=nc, seed=seed, det_scale=1.0, obj_shift=-12.0)
    gd, gw = YOLO_VARIANTS[variant]
    ch = [_c8(c * gw) for c in (64, 128, 256, 512, 1024)]

    def wire(name, outs, carried):
        """outs: {out channel: [(in channel, tap row, tap col, weight), ...]}; `carried`: the input channels that hold plate bits.
        No other output channel reads a carried bit (a level of 16 next to activations of ~0.5 would swamp the random channels), and a
        carrier output channel reads nothing but its taps."""
        w, b = sd[name + ".weight"], sd[name + ".bias"]
        w[:, list(carried)] = 0.0
        for o, taps in outs.items():
            w[o] = 0.0
            b[o] = 0.0
            for ci, r, c, v in taps:
                w[o, ci, r, c] = v

    def passthrough(name, n, in_off=0, also=()):
        wire(name, {j: [(in_off + j, 0, 0, 1.0)] for j in range(n)}, list(range(in_off, in_off + n)) + list(also))

    def c3(i, n, h, in_off=0):
        wire(f"model.{i}.cv1.conv", {}, range(in_off, in_off + n))
        passthrough(f"model.{i}.cv2.conv", n, in_off)
        passthrough(f"model.{i}.cv3.conv", n, in_off=h)            # cv3 reads [m | cv2]: the second half

    # stem (6 x 6 / s2 / p2, RGB input): rows / columns 2i, 2i + 1 are taps 2, 3
    wire("model.0.conv", {0: [(c, r, s, v) for r in (2, 3) for s in (2, 3) for c, v in ((0, -STEM_GAIN / 8), (1, STEM_GAIN / 4), (2, -STEM_GAIN / 8))]}, [])
    sd["model.0.conv.bias"][0] = STEM_BIAS
    # 3 x 3 / s2 / p1: rows 2i, 2i + 1 are taps 1, 2
    wire("model.1.conv", {0: [(0, r, s, 0.25) for r in (1, 2) for s in (1, 2)]}, [0])                              # 2 px -> 4 px: block mean
    c3(2, 1, ch[1] // 2)
    wire("model.3.conv", {2 * a + b: [(0, 1 + a, 1 + b, 1.0)] for a in (0, 1) for b in (0, 1)}, [0])             # 4 -> 8 px: which child
    c3(4, 4, ch[2] // 2)
    wire("model.5.conv", {4 * (2 * a + b) + q: [(q, 1 + a, 1 + b, 1.0)] for a in (0, 1) for b in (0, 1) for q in range(4)}, range(4))   # 8 -> 16 px
    c3(6, 16, ch[3] // 2)
    wire("model.7.conv", {16 * B + q: [(q, 1, 1 + B, 1.0)] for B in (0, 1) for q in range(16)}, range(16))        # 16 -> 32 px: the two top quarters
    c3(8, NBITS, ch[4] // 2)
    passthrough("model.9.cv1.conv", NBITS)
    c_ = ch[4] // 2
    passthrough("model.9.cv2.conv", NBITS, also=[k * c_ + j for k in (1, 2, 3) for j in range(NBITS)])   # reads [x | y1 | y2 | y3]: the un-pooled quarter; the pooled copies feed nothing
    passthrough("model.10.conv", NBITS)
    # off the carrier's route, but readers of tensors that hold plate bits: layer 12 = [up(10) | 6], 16 = [up(14) | 4]
    for nm in ("cv1", "cv2"):
        wire(f"model.13.{nm}.conv", {}, list(range(NBITS)) + list(range(ch[3], ch[3] + 16)))
        wire(f"model.17.{nm}.conv", {}, range(ch[2], ch[2] + 4))
    c3(23, NBITS, ch[4] // 2, in_off=ch[3])                             # layer 22 = [layer 21 | layer 10]
    sd["model.24.m.2.weight"][:, :NBITS] = 0.0                          # the other anchors of the level read no plate bit either

    def chan(k):
        """Channel of the stride-32 map that holds bit k (sub-block r = k // 8, c = k % 8 of the plate): c = 4 B + 2 b2 + b1, r = 2 a2 + a1."""
        r, c = divmod(k, 8)
        B, c = divmod(c, 4)
        return 16 * B + 4 * (2 * (r // 2) + (c // 2)) + 2 * (r % 2) + (c % 2)

    rng = np.random.default_rng(seed + 24)
    w, b = sd["model.24.m.2.weight"], sd["model.24.m.2.bias"]
    no = nc + 5
    rows = w[:no, :, 0, 0]
    rows[:] = rng.standard_normal(rows.shape).astype(np.float32) * np.float32(clutter / math.sqrt(ch[4]))
    rows[:5] *= np.float32(0.1)                                         # the box and objectness logits' random term: ~0.02 (a rank step is 0.25)
    rows[:, :NBITS] = 0.0
    hb = b[:no]
    hb[:] = 0.0
    fs = field_slices()
    for r, k in enumerate(("tx", "ty", "tw", "th")):
        lo, hi = T_RANGE[k]
        k0, n = fs[k]
        hb[r] = lo + (hi - lo) * 0.5 / (1 << n)
        for j in range(n):
            rows[r, chan(k0 + j)] = (hi - lo) * (1 << (n - 1 - j)) / (1 << n) / V
    rows[4, chan(fs["on"][0])] = 12.0 / V                                # objectness: -6 + 12 on a plate ...
    for j in range(4):
        rows[4, chan(fs["rank"][0] + j)] = 0.25 * (1 << (3 - j)) / V     # ... + 0.25 per rank
    hb[4] = -6.0
    for c in range(3):
        rows[5 + c, chan(fs["cls"][0] + c)] = 15.0 / V
    hb[5:] = -9.0
    return sd
