"""Where does the resize-in-front_fused path differ from the separate launches?  VC_FRAME_HW=h,w"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import vehicle_counting_amd.engine as E
from vehicle_counting_amd.synth import synth_frames
from vehicle_counting_amd.weights import synth_yolo
H, W = (int(v) for v in os.environ.get("VC_FRAME_HW", "273,521").split(","))
sd = synth_yolo("yolov5s", nc=4, seed=1702, det_scale=4.0, obj_shift=0.0)
fr = synth_frames(2, H, W, n_obj=4, seed=H * 7 + W)
eng = E.Engine(sd, None, precision="bf16", num_classes=4, max_batch=2, max_frame_hw=(H, W))
eng.detect([f[:, :, ::-1] for f in fr])
a1 = eng.debug_layer(1, batch=2)
dev = torch.from_numpy(fr).cuda()
eng.stream_submit(dev.data_ptr(), 2, H, W); eng.sync()
b1 = eng.debug_layer(1, batch=2)
d = (a1 != b1).any(-1)
print("shape", a1.shape, "mismatching pixels", int(d.sum()))
for n in range(2):
    ys, xs = np.nonzero(d[n])
    if len(ys):
        print("frame", n, "rows", sorted(set(ys.tolist())), "cols", sorted(set(xs.tolist()))[:40])
        y, x = ys[0], xs[0]
        print(" first", y, x, a1[n, y, x, :6], b1[n, y, x, :6])
