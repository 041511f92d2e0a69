"""CPU ORACLE (test infrastructure, never a product path) -- the whole per-video loop, row C1 of SURVEY.md section 8.

Restates /root/reference/modules/__init__.py:54-84 over the oracle stages: AutoShape detect (oracle/yolov5.py) ->
marshal (networks/yolo.py:72-97) -> skip empty frames (Q1) -> VideoTracker.run (oracle/deepsort.py) -> VideoCounting
(oracle/counting.py).  Also the `cpu_baseline` leg of bench.py (kind "port")."""
from __future__ import annotations

import numpy as np

from . import counting as oc
from . import deepsort as od
from . import reid as orr
from . import yolov5 as oy


def run_video(frames_bgr, yolo_sd, reid_sd, tracking_config, zone_path, variant="yolov5s", nc=80, conf=0.25, iou=0.45,
              max_det=300, timings=None, size=640, bf16=False):
    """bf16=True: detector and ReID net restated in the product's benchmarked precision (oracle/yolov5.py::forward, oracle/reid.py::
    reid_forward_bf16); everything downstream (NMS, DeepSORT, counting) is unchanged."""
    embed = orr.make_embedder(reid_sd, bf16=bf16)
    tracker = od.VideoTrackerOracle(nc, tracking_config, embed)
    polygon, dirs = oc.load_zone(zone_path)
    obj = {"frames": [], "tracks": [], "labels": [], "boxes": []}
    n_det = []
    for i, f in enumerate(frames_bgr):
        det = oy.autoshape_detect(yolo_sd, [f[:, :, ::-1]], variant, nc, size, conf, iou, None, max_det, bf16=bf16)[0]
        m = oy.marshal_like_reference(det)
        n_det.append(len(m["bboxes"]))
        if len(m["bboxes"]) == 0:
            continue
        res = tracker.run(f, m["bboxes"], m["classes"], m["scores"])
        for j in range(len(res["boxes"])):
            obj["frames"].append(i + 1)
            obj["tracks"].append(int(res["tracks"][j]))
            obj["labels"].append(int(res["labels"][j]))
            obj["boxes"].append(np.asarray(res["boxes"][j]))
    td = oc.build_track_dict(nc, polygon, dirs, obj["frames"], obj["tracks"], obj["labels"], obj["boxes"])
    rows = oc.csv_rows(td)
    return rows, oc.direction_counts(rows, list(dirs.keys()), nc), n_det
