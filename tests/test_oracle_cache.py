"""CPU: the committed oracle cache (tests/oracle_cache.py) is what the oracle computes.

Every file is named by the hash of its inputs, so a stale file can never be read for changed inputs; what remains to check is that the
contents are the oracle's: one cached call (the clip of tests/test_gpu_pipeline.py::test_csv_parity) is recomputed from scratch here
and compared field by field, and every file decodes."""
import glob
import json
import os

import numpy as np

import oracle_cache
from oracle import pipeline as op
from vehicle_counting_amd.synth import synth_frames
from vehicle_counting_amd.weights import synth_reid, synth_yolo

TRACK_CFG = dict(MAX_DIST=0.2, MIN_CONFIDENCE=0.25, NMS_MAX_OVERLAP=0.5, MAX_IOU_DISTANCE=0.6, MAX_AGE=30, N_INIT=3, NN_BUDGET=60)


def test_round_trip_of_the_encoding():
    x = {"a": [np.arange(6, dtype=np.int64).reshape(2, 3), (1.5, 2.5)], 3: {"k": "v"}, "rows": [{"box": [1, 2, 3, 4], "fpoint": (0.5, 1.0)}]}
    y = oracle_cache._dec(json.loads(json.dumps(oracle_cache._enc(x))))
    assert y["a"][1] == (1.5, 2.5) and y[3] == {"k": "v"} and y["rows"] == x["rows"]
    np.testing.assert_array_equal(y["a"][0], x["a"][0])
    assert y["a"][0].dtype == np.int64
    assert oracle_cache.digest([x["a"][0], "s", 1]) != oracle_cache.digest([x["a"][0] + 1, "s", 1])


def test_every_cache_file_decodes():
    files = glob.glob(os.path.join(oracle_cache.DIR, "*.json"))
    assert len(files) >= 4, files
    for p in files:
        with open(p) as f:
            oracle_cache._dec(json.load(f))


def test_a_cached_csv_is_what_the_oracle_computes(golden_dir, monkeypatch):
    frames = synth_frames(18, 360, 640, n_obj=6, seed=3)
    ysd, rsd = synth_yolo("yolov5s", nc=8, seed=1702, det_scale=4.0, obj_shift=0.0), synth_reid(1702)
    zone = os.path.join(golden_dir, "cam_04_halfres.json")
    assert hasattr(op.run_video, "__wrapped__")                      # tests/conftest.py installed the cache
    live = op.run_video.__wrapped__(frames, ysd, rsd, TRACK_CFG, zone, nc=8)
    reads = []
    real_open = open

    def spy(path, *a, **k):
        reads.append(str(path))
        return real_open(path, *a, **k)
    monkeypatch.setattr("builtins.open", spy)
    cached = op.run_video(frames, ysd, rsd, TRACK_CFG, zone, nc=8)
    monkeypatch.undo()
    assert any(oracle_cache.DIR in p for p in reads), "this call is not in the committed cache"
    assert cached[2] == live[2] and cached[1] == live[1] and len(cached[0]) == len(live[0]) > 10
    for a, b in zip(cached[0], live[0]):
        assert a == b, (a, b)
