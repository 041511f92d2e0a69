"""Where the fused s2 + pointwise launch differs from the two launches (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import vehicle_counting_amd.engine as E
from vehicle_counting_amd.synth import synth_frames
from vehicle_counting_amd.weights import synth_yolo
sd = synth_yolo("yolov5s", nc=80, seed=1702, det_scale=4.0, obj_shift=0.5)
B, H, W = 1, 640, 640
frames = synth_frames(B, H, W, n_obj=6, seed=13)
imgs = [f[:, :, ::-1] for f in frames]
eng = E.Engine(sd, None, precision="bf16", num_classes=80, max_batch=B, max_frame_hw=(H, W))
out = {}
for on in (1, 0):
    eng.set_option("fuse_s2_pw", on)
    eng.set_option("bneck_fused", 0); eng.set_option("bneck_cv3", 0)
    eng.detect(imgs)
    out[on] = {l: eng.debug_layer(l, batch=B) for l in (4, 3)}
a, b = out[1][4], out[0][4]
print("layer 3 equal:", np.array_equal(out[1][3], out[0][3]))
d = a != b
print("layer 4 shape", a.shape, "mismatch fraction", d.mean(), "max abs diff", np.abs(a - b).max())
ax = tuple(i for i in range(a.ndim))
for axis in range(a.ndim):
    other = tuple(i for i in range(a.ndim) if i != axis)
    frac = d.mean(axis=other)
    print("axis", axis, "len", len(frac), "mismatch by index (first 40):", np.round(frac[:40], 2))
