"""Which crops differ between the two bf16 crop kernels (workgroup-per-crop / thread-per-pixel)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import vehicle_counting_amd.engine as E
from vehicle_counting_amd.weights import synth_reid
from oracle.deepsort import crop_corners
rng = np.random.default_rng(11)
H, W = 271, 523
img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
boxes = [[W - 3.0, H - 3.0, 6.0, 6.0], [2.0, 2.0, 5.0, 5.0], [W - 1.5, 100.0, 3.0, 40.0], [100.0, H - 1.5, 60.0, 3.0],
         [W / 2, H / 2, 2.0 * W, 2.0 * H], [W / 2, H / 2, W, 2.0], [W - 25.0, H - 25.0, 50.0, 50.0]]
for _ in range(41):
    w, h = rng.uniform(2, 300), rng.uniform(2, 260)
    boxes.append([rng.uniform(-20, W + 20), rng.uniform(-20, H + 20), w, h])
boxes = np.asarray([b for b in boxes if (lambda c: c[2] > c[0] and c[3] > c[1])(crop_corners(np.asarray(b), W, H))])
eng = E.Engine(None, synth_reid(1702), precision="bf16", max_crops=64, max_frame_hw=(H, W))
a = eng.embed(img, boxes)
eng.set_option("crop_per_pixel", 1)
b = eng.embed(img, boxes)
for i in range(len(boxes)):
    if not np.array_equal(a[i], b[i]):
        print(i, boxes[i], crop_corners(boxes[i], W, H), float(np.abs(a[i] - b[i]).max()))
from oracle import reid as orr
crops = []
for bx in boxes:
    x1, y1, x2, y2 = crop_corners(bx, W, H)
    crops.append(img[y1:y2, x1:x2])
ref = orr.preprocess_crops(crops).transpose(0, 2, 3, 1)
f32 = E.Engine(None, synth_reid(1702), precision="f32", max_crops=64, max_frame_hw=(H, W))
f32.embed(img, boxes); xf = f32.embed_input(len(boxes))
print("fp32 engine input == oracle bit for bit:", np.array_equal(xf, ref), float(np.abs(xf - ref).max()))
import torch
refb = torch.from_numpy(ref).to(torch.bfloat16).float().numpy()
eng.set_option("crop_per_pixel", 0); eng.embed(img, boxes); xa = eng.embed_input(len(boxes))
eng.set_option("crop_per_pixel", 1); eng.embed(img, boxes); xb = eng.embed_input(len(boxes))
print("wg kernel == bf16(oracle):", np.array_equal(xa, refb), int((xa != refb).sum()), "per-pixel kernel == bf16(oracle):", np.array_equal(xb, refb), int((xb != refb).sum()))
d = np.argwhere(xa != xb)
print("wg vs per-pixel differing elements", len(d), d[:10].tolist())
for n, y, x, c in d[:6]:
    print(n, y, x, c, "wg", xa[n, y, x, c], "pp", xb[n, y, x, c], "ref f32", ref[n, y, x, c], "fp32 eng", xf[n, y, x, c], "crop", crop_corners(boxes[n], W, H))
