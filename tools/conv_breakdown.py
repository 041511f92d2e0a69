"""Per-conv-launch timing of one bench step (profiling mode: hipEvents around every launch on the engine stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vehicle_counting_amd.engine as E
from vehicle_counting_amd.synth import synth_frames
from vehicle_counting_amd.weights import synth_reid, synth_yolo
B, NC = int(os.environ.get('VC_B', 16)), 80
MODEL, PREC = os.environ.get("VC_MODEL", "yolov5s"), os.environ.get("VC_PREC", "bf16")
S = int(os.environ.get("VC_SIZE", 640))
H, W = (int(v) for v in os.environ.get("VC_FRAME_HW", f"{S},{S}").split(","))        # frame geometry; the tensor follows AutoShape (720,1280 -> 384 x 640)
eng = E.Engine(synth_yolo(MODEL, nc=NC, det_scale=4.0, obj_shift=float(os.environ.get('VC_OBJ_SHIFT', -8.0))), synth_reid(), precision=PREC, model_name=MODEL, num_classes=NC, max_batch=B,
               img_size=S, max_frame_hw=(H, W), max_crops=B * 64, max_tracks=8192, nn_budget_cap=60, max_candidates=8192)
tr = [eng.tracker_create(max_dist=0.2, min_confidence=0.25, nms_max_overlap=0.5, max_iou_distance=0.6, max_age=30, n_init=3, nn_budget=60) for _ in range(NC)]
fr = torch.from_numpy(synth_frames(B, H, W, 12, 1702)).cuda()
if os.environ.get("VC_INJECT", "1") != "0":          # ground-truth rectangles instead of the random head's boxes (the convs are what is timed)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import injected_detections
    eng.stream_inject(*injected_detections(B, (H, W), 12, 1702))
for _ in range(3): eng.stream_run(tr, fr.data_ptr(), B, H, W)
eng.profile(True); eng.profile_reset()
eng.stream_run(tr, fr.data_ptr(), B, H, W)
lines = eng.profile_ops().strip().split("\n")
tot = 0
for l in lines:
    print(l); tot += float(l.split("ms=")[1].split()[0])
print("total conv ms", tot, "launches", len(lines), "B", B, "conv ms per frame", tot / B)
import vehicle_counting_amd._lib as L
print({k: eng.profile_read(c) for k, c in (("conv", 0), ("detect_aux", 1), ("reid_aux", 2), ("track", 3))})
eng.close()
