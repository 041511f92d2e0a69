"""front_fused_kernel / c3_fused_kernel / bneck timing (profiling events), five repetitions -- for A/B runs of two library builds (VC_LIB_PATH)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import vehicle_counting_amd.engine as E
from vehicle_counting_amd.synth import synth_frames
from vehicle_counting_amd.weights import synth_yolo
B, NC, H, W = 128, 80, 640, 640
eng = E.Engine(synth_yolo("yolov5s", nc=NC, det_scale=4.0, obj_shift=-8.0), None, precision="bf16", num_classes=NC, max_batch=B, max_frame_hw=(H, W))
fr = torch.from_numpy(synth_frames(B, H, W, 12, 1702)).cuda()
for _ in range(2):
    eng.stream_submit(fr.data_ptr(), B, H, W); eng.sync(); eng.stream_reset()
acc = {}
for rep in range(5):
    eng.profile(True); eng.profile_reset()
    eng.stream_submit(fr.data_ptr(), B, H, W); eng.sync()
    for x in eng.profile_ops().strip().split("\n"):
        for cfg in ("cfg=102", "cfg=103", "cfg=104"):
            if cfg in x:
                acc.setdefault(cfg, []).append(float(x.split("ms=")[1].split()[0]))
    eng.profile(False); eng.stream_reset()
print(os.environ.get("VC_LIB_PATH", "default lib"), {k: round(float(np.median(v)), 4) for k, v in acc.items()}, flush=True)
