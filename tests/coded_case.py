"""Shared by tests/test_oracle_coded.py, tests/test_gpu_coded.py and tests/golden/make_coded_golden.py: the well-conditioned clips
(vehicle_counting_amd/coded.py) whose fp32-oracle CSV is committed under tests/golden/coded_*.json."""
import json
import os

import numpy as np

from vehicle_counting_amd.coded import coded_frames, coded_yolo
from vehicle_counting_amd.weights import synth_reid

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NC = 80
TRACK_CFG = dict(MAX_DIST=0.2, MIN_CONFIDENCE=0.25, NMS_MAX_OVERLAP=0.5, MAX_IOU_DISTANCE=0.6, MAX_AGE=30, N_INIT=3, NN_BUDGET=60)
CASES = {
    # BASELINE.json configs[1]: YOLOv5s, 640 x 640 frames
    "s640": dict(variant="yolov5s", size=640, H=640, W=640, T=64, n_obj=12, seed=1702, zone="cam_04_halfres.json"),
    # the reference's only real geometry: 1280 x 720 frames -> 384 x 640 tensor (Q8), the reference's own zone file's directions
    "s720p": dict(variant="yolov5s", size=640, H=720, W=1280, T=64, n_obj=12, seed=1702, zone="cam_04.json"),
    # BASELINE.json configs[2]: YOLOv5m, 1024 x 1024 frames
    "m1024": dict(variant="yolov5m", size=1024, H=1024, W=1024, T=24, n_obj=12, seed=1702, zone="cam_04.json"),
    # BASELINE.json configs[4]: YOLOv5l, 1280 x 1280 frames (the fp8 engine's case)
    "l1280": dict(variant="yolov5l", size=1280, H=1280, W=1280, T=24, n_obj=12, seed=1702, zone="cam_04.json"),
}


def build(name):
    c = CASES[name]
    frames, truth = coded_frames(c["T"], c["H"], c["W"], n_obj=c["n_obj"], seed=c["seed"], size=c["size"])
    return coded_yolo(c["variant"], nc=NC), synth_reid(1702), frames, truth


def zone_file(name, out_dir):
    """The case's direction annotations with the zone polygon widened to the whole frame: every tracked row reaches the CSV."""
    c = CASES[name]
    with open(os.path.join(GOLDEN, c["zone"])) as f:
        z = json.load(f)
    for sh in z["shapes"]:
        if sh["label"] == "zone":
            sh["points"] = [[0.0, 0.0], [float(c["W"]), 0.0], [float(c["W"]), float(c["H"])], [0.0, float(c["H"])]]
    path = os.path.join(str(out_dir), f"zone_{name}.json")
    with open(path, "w") as f:
        json.dump(z, f)
    return path


def golden_path(name):
    return os.path.join(GOLDEN, f"coded_{name}.json")


def load_golden(name):
    with open(golden_path(name)) as f:
        g = json.load(f)
    for r in g["rows"]:
        r["fpoint"], r["lpoint"] = tuple(r["fpoint"]), tuple(r["lpoint"])
    return g


def key(rows):
    return [(r["label"], r["track_id"], r["frame_id"], r["direction"], r["fframe"], r["lframe"]) for r in rows]


def compare_rows(rows, ref_rows, box_px, point_px):
    """CSV equality (utilities/counting/utils.py:154-198 minus the colour column, Q10): same rows in the same order -- label, track id,
    frame, direction, first / last frame exact -- boxes within box_px, first / last points within point_px."""
    assert key(rows) == key(ref_rows), (len(rows), len(ref_rows), [a for a, b in zip(key(rows), key(ref_rows)) if a != b][:5])
    db = max((float(np.abs(np.array(r["box"]) - np.array(q["box"])).max()) for r, q in zip(rows, ref_rows)), default=0.0)
    dp = max((max(float(np.abs(np.array(r["fpoint"]) - np.array(q["fpoint"])).max()), float(np.abs(np.array(r["lpoint"]) - np.array(q["lpoint"])).max()))
              for r, q in zip(rows, ref_rows)), default=0.0)
    assert db <= box_px and dp <= point_px, (db, dp)
    return db, dp
