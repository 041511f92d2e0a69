"""CPU: the well-conditioned detector of vehicle_counting_amd/coded.py under the oracle, and the committed oracle CSVs of its clips.

  * every plate painted into a frame comes back as ONE detection of the plate's class whose box is the one the plate's bits decode to,
    in the oracle's fp32 arithmetic and in its restatement of the bf16 engine's arithmetic, with every logit far from conf_thres;
  * tests/golden/coded_s640.json (what the GPU parity test compares with) is what the oracle computes here: the first frames of the clip
    re-run through oracle/pipeline.py give the committed rows of those frames;
  * the restated bf16 arithmetic gives the same CSV as fp32 (the property the GPU test then demands of the bf16 / fp8 engines)."""
import numpy as np
import pytest

import coded_case as cc
from oracle import pipeline as op
from oracle import yolov5 as oy
from vehicle_counting_amd.coded import coded_frames, coded_yolo


@pytest.mark.parametrize("variant,size,hw", [("yolov5s", 640, (640, 640)), ("yolov5s", 640, (720, 1280)), ("yolov5m", 1024, (1024, 1024))])
def test_every_plate_is_one_detection_with_margin(variant, size, hw):
    nc = 8
    sd = coded_yolo(variant, nc=nc)
    frames, truth = coded_frames(2, hw[0], hw[1], n_obj=12, seed=5, size=size)
    x, s0, s1 = oy.preprocess([f[:, :, ::-1] for f in frames], size)
    for bf16 in (False, True):
        pred = oy.forward(sd, x, variant, nc, bf16=bf16).numpy()
        conf = pred[..., 4:5] * pred[..., 5:]
        best = conf.max(-1)
        assert not ((best > 0.02) & (best < 0.9)).any()                 # nothing near conf_thres = 0.25: a candidate is a plate or far below
        dets = oy.non_max_suppression(pred, 0.25, 0.45, None, 300)
        for t, d in enumerate(dets):
            d = np.concatenate((oy.scale_coords(s1, d[:, :4], s0[t]), d[:, 4:]), 1)
            assert len(d) == len(truth[t]) and len(d) >= 8, (len(d), len(truth[t]))
            for i, lab, x1, y1, x2, y2 in truth[t]:
                e = np.abs(d[:, :4] - np.array([x1, y1, x2, y2])).max(1)
                j = int(e.argmin())
                assert e[j] <= 2.5 and int(d[j, 5]) == lab, (variant, bf16, i, e[j], d[j])     # the random term of the box logits: up to ~2 px


def test_committed_oracle_rows_are_what_the_oracle_computes(tmp_path):
    name, n = "s640", 12
    g = cc.load_golden(name)
    ysd, rsd, frames, truth = cc.build(name)
    assert [len(t) for t in truth] == g["plates_per_frame"]
    rows, _, n_det = op.run_video(frames[:n], ysd, rsd, cc.TRACK_CFG, cc.zone_file(name, tmp_path), nc=cc.NC)
    assert n_det == g["n_det"][:n]
    sub = lambda rs: [(r["label"], r["track_id"], r["frame_id"], tuple(r["box"])) for r in rs if r["frame_id"] <= n]
    assert sub(rows) == sub(g["rows"]) and len(rows) > 6 * n


def test_bf16_restatement_gives_the_same_csv(tmp_path):
    name, n = "s720p", 24
    ysd, rsd, frames, _ = cc.build(name)
    zone = cc.zone_file(name, tmp_path)
    a, ca, na = op.run_video(frames[:n], ysd, rsd, cc.TRACK_CFG, zone, nc=cc.NC)
    b, cb, nb = op.run_video(frames[:n], ysd, rsd, cc.TRACK_CFG, zone, nc=cc.NC, bf16=True)
    assert na == nb and len(a) > 6 * n
    cc.compare_rows(b, a, box_px=1, point_px=1.0)
    assert ca == cb
