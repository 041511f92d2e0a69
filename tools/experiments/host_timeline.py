"""Where does a bench step spend host time?  (python-side timers around submit / run)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import vehicle_counting_amd.engine as E
from vehicle_counting_amd.synth import synth_frames
from vehicle_counting_amd.weights import synth_reid, synth_yolo
B, H, W, NC = int(os.environ.get("VC_B", 16)), 640, 640, 80
eng = E.Engine(synth_yolo("yolov5s", nc=NC, det_scale=4.0, obj_shift=1.0), synth_reid(), precision="bf16", num_classes=NC, max_batch=B,
               max_frame_hw=(H, W), max_crops=B * 64, max_tracks=8192, nn_budget_cap=60)
tr = [eng.tracker_create(max_dist=0.2, min_confidence=0.25, nms_max_overlap=0.5, max_iou_distance=0.6, max_age=30, n_init=3, nn_budget=60) for _ in range(NC)]
CLIP = 128
fr = torch.from_numpy(synth_frames(CLIP, H, W, 12, 1702)).cuda()
ptr = lambda i: fr[(i * B) % CLIP:(i * B) % CLIP + B].data_ptr()
eng.stream_submit(ptr(0), B, H, W)
ts, tr_ = [], []
for i in range(30):
    t0 = time.perf_counter(); eng.stream_submit(ptr(i + 1), B, H, W); t1 = time.perf_counter()
    eng.stream_run(tr, ptr(i), B, H, W); t2 = time.perf_counter()
    if i >= 5: ts.append(t1 - t0); tr_.append(t2 - t1)
print(f"B={B} submit {np.mean(ts)*1e3:.2f} ms  run {np.mean(tr_)*1e3:.2f} ms  -> {B/(np.mean(ts)+np.mean(tr_)):.0f} frames/s")
eng.stream_run(tr, ptr(30), B, H, W)   # drain the submission left in flight
# detector alone, GPU-bound?
torch.cuda.synchronize(); eng.sync()
t0 = time.perf_counter()
for i in range(20):
    eng.stream_submit(ptr(i), B, H, W)
    eng.sync()
    eng._drain = eng.stream_run(tr, ptr(i), B, H, W)
t1 = time.perf_counter()
print("serial submit+sync+run per step ms", (t1 - t0) / 20 * 1e3)
# tracker round trips without a concurrent detector: submit, wait for everything, then run with VC_TIMING on
if os.environ.get("VC_TIMING"):
    print("--- serial (no detector running during the tracker loop) ---")
    for i in range(20):
        eng.stream_submit(ptr(i), B, H, W)
        eng.sync(); torch.cuda.synchronize(); time.sleep(0.002)
        eng.stream_run(tr, ptr(i), B, H, W)
