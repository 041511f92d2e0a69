"""Synthetic-head calibration: detections per frame after NMS for (det_scale, obj_shift) pairs of a model / size / precision."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vehicle_counting_amd.engine as E
from vehicle_counting_amd.synth import synth_frames
from vehicle_counting_amd.weights import synth_yolo
MODEL, PREC, S = os.environ.get("VC_MODEL", "yolov5l"), os.environ.get("VC_PREC", "fp8"), int(os.environ.get("VC_SIZE", 1280))
FH, FW = (int(v) for v in os.environ.get("VC_FRAME_HW", f"{S},{S}").split(","))      # frame geometry (the tensor follows AutoShape: 720,1280 -> 384 x 640)
SCALES = [float(v) for v in os.environ.get("VC_SCALES", "0.25,1.0").split(",")]
SHIFTS = [float(v) for v in os.environ.get("VC_SHIFTS", "-12,-16,-24,-32,-48").split(",")]
S_FRAME = (FH, FW)
fr = synth_frames(4, FH, FW, int(os.environ.get("VC_OBJECTS", 16)), 1702, bounce=True)
for ds in SCALES:
    for sh in SHIFTS:
        eng = E.Engine(synth_yolo(MODEL, nc=80, det_scale=ds, obj_shift=sh), None, precision=PREC, model_name=MODEL, num_classes=80, max_batch=4,
                       img_size=S, max_frame_hw=S_FRAME, max_candidates=8192)
        try:
            d = eng.detect([f[:, :, ::-1] for f in fr])
            n = [len(x) for x in d]
        except Exception as ex:
            n = str(ex)[:60]
        print("det_scale", ds, "obj_shift", sh, "detections per frame", n, flush=True)
        eng.close()
