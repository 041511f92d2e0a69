"""GPU: parity on the configuration bench.py measures (BASELINE.json configs[1]: YOLOv5s, 80 classes, 640x640 frames, batched
asynchronous stream path, bench weights) and on the other named operating points the round-1 tests left out:
  * fp32 engine: the CSV artefact equals the oracle's (ids / frames / directions exact, boxes +-1 px);
  * bf16 engine (the benchmarked precision): a STATED track-level tolerance against the same oracle rows;
  * 1280x720 frames + the reference's real zone file demo/sample/cam_04.json through the fused stream path (letterbox resize, Q8);
  * configs[2] track leg: 256 detections per frame, 256 live tracks, full 60-sample galleries."""
import os
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import vehicle_counting_amd.engine as E  # noqa: E402
from oracle import deepsort as od  # noqa: E402
from oracle import pipeline as op  # noqa: E402
from vehicle_counting_amd.pipeline import CountingPipeline, FrameSource  # noqa: E402
from vehicle_counting_amd.synth import synth_frames  # noqa: E402
from vehicle_counting_amd.weights import synth_reid, synth_yolo  # noqa: E402

NC = 80
TRACK_CFG = dict(MAX_DIST=0.2, MIN_CONFIDENCE=0.25, NMS_MAX_OVERLAP=0.5, MAX_IOU_DISTANCE=0.6, MAX_AGE=30, N_INIT=3, NN_BUDGET=60)
T, B, H, W = 64, 16, 640, 640


def key(rows):
    return [(r["label"], r["track_id"], r["frame_id"], r["direction"], r["fframe"], r["lframe"]) for r in rows]


@pytest.fixture(scope="module")
def bench_case(golden_dir, tmp_path_factory):
    """bench.py's own inputs: weights synth_yolo(seed 1702, det_scale 4, obj_shift 1), frames synth_frames(seed 1702, 12 objects).
    The zone polygon is widened to the whole frame (directions as in cam_04) so that EVERY tracked row reaches the CSV: the
    comparison then covers the complete tracker output of the clip, not only the boxes inside the demo camera's road polygon."""
    import json
    ysd, rsd = synth_yolo("yolov5s", nc=NC, seed=1702, det_scale=4.0, obj_shift=1.0), synth_reid(1702)
    frames = synth_frames(T, H, W, n_obj=12, seed=1702)
    with open(os.path.join(golden_dir, "cam_04_halfres.json")) as f:
        z = json.load(f)
    for sh in z["shapes"]:
        if sh["label"] == "zone":
            sh["points"] = [[0.0, 0.0], [float(W), 0.0], [float(W), float(H)], [0.0, float(H)]]
    zone = str(tmp_path_factory.mktemp("zone") / "cam_04.json")
    with open(zone, "w") as f:
        json.dump(z, f)
    ref_rows, ref_counts, n_det = op.run_video(frames, ysd, rsd, TRACK_CFG, zone, nc=NC)
    assert sum(n_det) > 8 * T and len(ref_rows) > 20, (sum(n_det), len(ref_rows))
    return ysd, rsd, frames, zone, ref_rows, ref_counts


def run_product(bench_case, precision, tmp_path):
    ysd, rsd, frames, zone, _, _ = bench_case
    cfg = types.SimpleNamespace(model_name="yolov5s", min_conf=0.25, min_iou=0.45, max_det=300)
    args = types.SimpleNamespace(weight=None, mapping=None, output_path=str(tmp_path))
    eng = E.Engine(ysd, rsd, precision=precision, num_classes=NC, max_batch=B, max_frame_hw=(H, W), max_crops=B * 64,
                   max_tracks=4096, nn_budget_cap=60)
    pipe = CountingPipeline(args, cfg, {"cam": {"cam_04": {"tracking_config": TRACK_CFG}}}, engine=eng,
                            class_names=[f"c{i}" for i in range(NC)])
    rows, counts = pipe.run_stream(FrameSource(frames), "cam_04", zone, batch=B, asynchronous=True)
    eng.close()
    return rows, counts


def test_bench_config_fp32_csv_equals_oracle(bench_case, tmp_path):
    rows, counts = run_product(bench_case, "f32", tmp_path)
    ref_rows, ref_counts = bench_case[4], bench_case[5]
    assert key(rows) == key(ref_rows)
    for r, q in zip(rows, ref_rows):
        assert np.abs(np.array(r["box"]) - np.array(q["box"])).max() <= 1, (r, q)
        assert np.abs(np.array(r["fpoint"]) - np.array(q["fpoint"])).max() <= 0.5
        assert np.abs(np.array(r["lpoint"]) - np.array(q["lpoint"])).max() <= 0.5
    assert counts == ref_counts


def _iou(a, b):
    x1, y1, x2, y2 = max(a[0], b[0]), max(a[1], b[1]), min(a[2], b[2]), min(a[3], b[3])
    inter = max(x2 - x1, 0) * max(y2 - y1, 0)
    return inter / max((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter, 1e-9)


def track_level_agreement(rows, ref_rows):
    """Row-level agreement of two CSVs that may number their tracks differently: a reference row is FOUND when the other CSV
    has a row of the same class in the same frame whose box overlaps it with IoU >= 0.9; found rows vote for an id mapping
    (reference id -> product id per class), and a row is ID-CONSISTENT when its partner carries the majority id."""
    by_frame = {}
    for r in rows:
        by_frame.setdefault((r["frame_id"], r["label"]), []).append(r)
    votes, pairs, dpx = {}, [], []
    for q in ref_rows:
        best, bi = None, 0.0
        for r in by_frame.get((q["frame_id"], q["label"]), []):
            i = _iou(q["box"], r["box"])
            if i > bi:
                best, bi = r, i
        if best is not None and bi >= 0.9:
            pairs.append((q, best))
            dpx.append(np.abs(np.array(q["box"]) - np.array(best["box"])).max())
            votes.setdefault((q["label"], q["track_id"]), {}).setdefault(best["track_id"], 0)
            votes[(q["label"], q["track_id"])][best["track_id"]] += 1
    major = {k: max(v, key=v.get) for k, v in votes.items()}
    consistent = sum(1 for q, r in pairs if major[(q["label"], q["track_id"])] == r["track_id"])
    same_dir = sum(1 for q, r in pairs if q["direction"] == r["direction"])
    n = max(len(ref_rows), 1)
    return {"found": len(pairs) / n, "id_consistent": consistent / n, "same_direction": same_dir / n,
            "box_px_p95": float(np.percentile(dpx, 95)) if dpx else 0.0, "box_px_max": float(max(dpx)) if dpx else 0.0,
            "extra_rows": (len(rows) - len(pairs)) / n}


def test_bench_config_bf16_track_level_tolerance(bench_case, tmp_path):
    """The benchmarked precision (bf16 convs) against the fp32 CPU oracle on bench.py's own weights and frames.  The seeded
    synthetic detector is the hard case for this comparison: its objectness logits are dense around conf_thres, so bf16 rounding
    flips marginal detections, and a flipped detection starts / ends / re-numbers a track (with identical detections the tracker
    is exact in both precisions, tests/test_gpu_pipeline.py).  Stated tolerance ('bf16 end to end'):
    >= 70 % of the reference CSV rows are found (same frame and class, IoU >= 0.9) with a consistent track id and the same
    direction, found boxes within 5 px (95th percentile), at most 45 % of the product's rows without a reference partner, and the
    per-(direction, class) counts differ by at most 3 tracks.  Which bf16 kernels run (fused or not, the autotuner's tile families)
    moves the last bf16 bit of a few activations and with it a handful of the 55 reference rows: measured over those variants
    found 0.75-0.78, id-consistent 0.75-0.76, box p95 3.0-3.95 px, extra rows 0.20-0.36, count difference 1-2."""
    rows, counts = run_product(bench_case, "bf16", tmp_path)
    ref_rows, ref_counts = bench_case[4], bench_case[5]
    a = track_level_agreement(rows, ref_rows)
    cd = max(abs(int(x) - int(y)) for d in ref_counts for x, y in zip(counts[d], ref_counts[d]))
    print("bf16 vs oracle:", a, "rows", len(rows), "ref", len(ref_rows), "max count diff", cd)
    assert a["found"] >= 0.70 and a["id_consistent"] >= 0.70 and a["same_direction"] >= 0.70, a
    assert a["box_px_p95"] <= 5.0 and a["extra_rows"] <= 0.45, a
    assert cd <= 3, (counts, ref_counts)


@pytest.mark.parametrize("T128", [128, 256])
def test_timed_configuration_b128_autotuned_bf16_directly(T128):
    """"the exact configuration timed (B = 128, autotune on, bf16) is compared with nothing directly".  Here it is: bench.py's
    weights and its first batch (128 frames, and the 256 frames per call the headline uses since round 6) through vc_stream_submit on a
    bf16 engine with max_batch = the batch and the autotuner ON (this
    process's default: the tile configurations the timed run picks for the batch's size buckets, fused kernels included), compared
    with the fp32 oracle on frames 0, 31, 63 and the last one of the batch under the bf16 ladder of tests/test_gpu_nets.py: per-layer max-norm <= 6e-2 and
    rms <= 4e-2 of the oracle's tensor; every oracle box with conf >= 0.35 (conf_thres + the confidence tolerance) has a same-class partner with IoU >= 0.45 and >= 85 % are the
    same box (IoU >= 0.9, |dconf| <= 1e-1: the bench head multiplies logit noise by det_scale = 4 over 80 classes).  (A row-level comparison of this clip against the fp32 engine was tried first: found 0.57 /
    id-consistent 0.56 on the 128-frame bouncing clip -- lower than the 64-frame clip's 0.75 because every flipped marginal detection
    renumbers the tracks after it; a clip-dependent hit rate is not a tolerance, the numeric ladder is.)"""
    import torch

    from oracle import yolov5 as oy
    assert os.environ.get("VC_AUTOTUNE", "1") != "0", "this test is about the autotuned configuration"
    ysd, rsd = synth_yolo("yolov5s", nc=NC, seed=1702, det_scale=4.0, obj_shift=1.0), synth_reid(1702)
    frames = synth_frames(T128, H, W, n_obj=12, seed=1702, bounce=True)
    dev = torch.from_numpy(frames).cuda()
    eng = E.Engine(ysd, rsd, precision="bf16", num_classes=NC, max_batch=T128, max_frame_hw=(H, W), max_crops=T128 * 64, max_tracks=8192, nn_budget_cap=60)
    eng.stream_submit(dev.data_ptr(), T128, H, W)
    rows7, _ = eng.stream_embed(dev.data_ptr(), T128, H, W)        # detector + ReID of the whole batch; rows7 = frame, x1, y1, x2, y2, conf, label
    pick = [0, 31, 63, T128 - 1]
    imgs = [frames[f][:, :, ::-1] for f in pick]
    x, s0, s1 = oy.preprocess(imgs, 640)
    pred, ys, _ = oy.forward(ysd, x, "yolov5s", NC, return_layers=True)
    for layer in (4, 9, 17, 20, 23):
        got = eng.debug_layer(layer, batch=T128)[pick].transpose(0, 3, 1, 2)
        ref = ys[layer].numpy()
        assert got.shape == ref.shape, (layer, got.shape, ref.shape)
        assert np.abs(got - ref).max() <= 6e-2 * np.abs(ref).max(), layer
        assert np.sqrt(((got - ref) ** 2).mean()) <= 4e-2 * np.sqrt((ref ** 2).mean()), layer      # measured 3.6e-2 at layer 20 on the 80-class bench weights (8-class test weights: <= 2.6e-2)
    ref_dets = oy.autoshape_detect(ysd, imgs, "yolov5s", NC)
    n_ref = n_same = 0
    for f, r in zip(pick, ref_dets):
        d = rows7[rows7[:, 0] == f][:, 1:]
        for rb in r[r[:, 4] >= 0.35]:                          # conf_thres 0.25 + the confidence tolerance below: a box nearer the threshold may be lost to it
            same = d[d[:, 5] == rb[5]]
            assert len(same) > 0, (f, rb)
            iou = np.array([_iou(rb[:4], q[:4]) for q in same])
            j = int(iou.argmax())
            assert iou[j] >= 0.45, (f, rb, iou[j])
            n_ref += 1
            if iou[j] >= 0.9:
                n_same += 1
                assert abs(same[j, 4] - rb[4]) <= 1e-1         # measured 7.8e-2 on the bench head (det_scale 4 over 80 classes; the 8-class ladder holds 6e-2)
    assert n_ref >= 8 and n_same >= 0.85 * n_ref, (n_ref, n_same)
    eng.close()


def test_bench_config_bf16_against_the_bf16_restatement(bench_case, tmp_path):
    """How far apart are two CORRECT bf16 implementations of this network?  oracle/yolov5.py::forward(bf16=True) and oracle/reid.py::
    reid_forward_bf16 restate both networks in the product's arithmetic -- every weight, the input and every activation rounded to
    bfloat16 once, fp32 accumulation -- so the HIP bf16 path and the restatement differ ONLY in summation order (1e-5 relative before
    a rounding, i.e. ~1 % of the values of a layer land one bf16 ulp apart).  Measured on bench.py's own weights and frames: that
    seed grows to 8e-3 rms (relative) at the deepest layers -- a third of the distance between the bf16 engine and the fp32 oracle
    (2.6e-2) -- and the CSV artefacts agree on 84 % of the rows.  The synthetic detector amplifies perturbations layer by layer, and
    its confidences form a continuum around conf_thres: no bf16 implementation can reproduce the fp32 reference's CSV row for row on
    it.  Asserted: the engine is closer to the bf16 restatement than to the fp32 oracle at every probed
    layer (rms <= half), within 1.5e-2 rms / 3e-2 max-norm, the CSVs within the row-level tolerance of the fp32 comparison."""
    import torch
    ysd, rsd, frames, zone, _, _ = bench_case
    n = 32
    ref_rows, ref_counts, n_det = op.run_video(frames[:n], ysd, rsd, TRACK_CFG, zone, nc=NC, bf16=True)
    assert sum(n_det) > 8 * n and len(ref_rows) > 10, (sum(n_det), len(ref_rows))
    # tensor level: one frame, layers 4 / 9 / 17 / 23 and the decoded predictions
    from oracle import yolov5 as oy
    eng = E.Engine(ysd, None, precision="bf16", num_classes=NC, max_batch=1, max_frame_hw=(H, W))
    eng.debug_pred(arm=True)
    eng.detect([frames[0][:, :, ::-1]])
    x, _, _ = oy.preprocess([frames[0][:, :, ::-1]], 640)
    pred, ys, _ = oy.forward(ysd, x, "yolov5s", NC, return_layers=True, bf16=True)
    pred32, ys32, _ = oy.forward(ysd, x, "yolov5s", NC, return_layers=True)
    worst = worst_rms = rms32 = 0.0
    for layer in (4, 9, 17, 23):
        got, ref = eng.debug_layer(layer).transpose(0, 3, 1, 2), ys[layer].numpy()
        worst = max(worst, float(np.abs(got - ref).max() / np.abs(ref).max()))
        worst_rms = max(worst_rms, float(np.sqrt(((got - ref) ** 2).mean() / (ref ** 2).mean())))
        rms32 = max(rms32, float(np.sqrt(((got - ys32[layer].numpy()) ** 2).mean() / (ys32[layer].numpy() ** 2).mean())))
    dsig = np.abs(eng.debug_pred()[:1][..., 4:] - pred.numpy()[..., 4:])
    dp, dp999 = float(dsig.max()), float(np.quantile(dsig, 0.999))
    eng.close()
    print("bf16 engine vs bf16 restatement: worst layer max-norm rel", worst, "rms rel", worst_rms, "(vs fp32 oracle rms rel", rms32, ") max |d sigmoid|", dp,
          "99.9 %", dp999)
    # artefact level
    rows, counts = run_product((ysd, rsd, frames[:n], zone, None, None), "bf16", tmp_path)
    a = track_level_agreement(rows, ref_rows)
    cd = max(abs(int(x) - int(y)) for d in ref_counts for x, y in zip(counts[d], ref_counts[d]))
    print("bf16 vs bf16 restatement:", a, "rows", len(rows), "ref", len(ref_rows), "max count diff", cd)
    assert worst <= 3e-2 and worst_rms <= 1.5e-2 and worst_rms < rms32 / 2 and dp999 <= 2e-2, (worst, worst_rms, rms32, dp, dp999)
    assert a["found"] >= 0.70 and a["id_consistent"] >= 0.70 and a["same_direction"] >= 0.70 and a["extra_rows"] <= 0.45 and a["box_px_p95"] <= 5.0, a
    assert cd <= 3, (counts, ref_counts)


def test_720p_stream_with_the_reference_zone_file(golden_dir, tmp_path):
    """1280x720 BGR frames (the demo video's geometry) through vc_stream_run: device-side bilinear letterbox to 384x640 (Q8),
    zone / directions from the reference's own demo/sample/cam_04.json; CSV equal to the oracle's in fp32."""
    nc, n = 8, 12
    frames = synth_frames(n, 720, 1280, n_obj=8, seed=21)
    ysd, rsd = synth_yolo("yolov5s", nc=nc, seed=1702, det_scale=8.0, obj_shift=1.0), synth_reid(1702)
    zone = os.path.join(golden_dir, "cam_04.json")
    ref_rows, ref_counts, n_det = op.run_video(frames, ysd, rsd, TRACK_CFG, zone, nc=nc)
    assert sum(n_det) > 30 and len(ref_rows) > 5, (n_det, len(ref_rows))
    cfg = types.SimpleNamespace(model_name="yolov5s", min_conf=0.25, min_iou=0.45, max_det=300)
    args = types.SimpleNamespace(weight=None, mapping=None, output_path=str(tmp_path))
    for mode in ("stream", "stream_async"):
        eng = E.Engine(ysd, rsd, precision="f32", num_classes=nc, max_batch=4, max_frame_hw=(720, 1280), max_crops=512, max_tracks=1024,
                       nn_budget_cap=60)
        pipe = CountingPipeline(args, cfg, {"cam": {"cam_04": {"tracking_config": TRACK_CFG}}}, engine=eng, class_names=[f"c{i}" for i in range(nc)])
        rows, counts = pipe.run_stream(FrameSource(frames), "cam_04", zone, batch=4, asynchronous=mode == "stream_async")
        assert key(rows) == key(ref_rows), mode
        for r, q in zip(rows, ref_rows):
            assert np.abs(np.array(r["box"]) - np.array(q["box"])).max() <= 1, (mode, r, q)
        assert counts == ref_counts
        eng.close()


def _dense_scenario(n_obj, n_frames, seed):
    """n_obj objects, all visible in every frame (a few drop out now and then), unit features around per-object prototypes."""
    rng = np.random.default_rng(seed)
    protos = rng.standard_normal((n_obj, 512)).astype(np.float32)
    protos /= np.linalg.norm(protos, axis=1, keepdims=True)
    cols = int(np.ceil(np.sqrt(n_obj)))
    pos = np.stack([(np.arange(n_obj) % cols) * 75.0 + 40, (np.arange(n_obj) // cols) * 75.0 + 40], 1) + rng.uniform(-5, 5, (n_obj, 2))
    vel = rng.uniform(-1.0, 1.0, (n_obj, 2))
    wh = rng.uniform([30, 30], [60, 60], (n_obj, 2))
    frames = []
    for t in range(n_frames):
        dets = []
        for i in range(n_obj):
            if t > 5 and rng.uniform() < 0.02:
                continue
            c = pos[i] + vel[i] * t + rng.normal(0, 0.4, 2)
            s = wh[i] * (1 + rng.normal(0, 0.01, 2))
            f = protos[i] + 0.02 * rng.standard_normal(512).astype(np.float32)
            dets.append({"tlwh": np.array([c[0] - s[0] / 2, c[1] - s[1] / 2, s[0], s[1]]), "conf": float(rng.uniform(0.3, 0.95)),
                         "feature": (f / np.linalg.norm(f)).astype(np.float32)})
        order = rng.permutation(len(dets))
        frames.append([dets[j] for j in order])
    return frames


@pytest.mark.parametrize("arena_mb", [1024, 0])
def test_config2_track_leg_256_detections_full_galleries(arena_mb):
    """BASELINE.json configs[2] track side (SURVEY.md B11 worst case): ~256 detections per frame against ~256 live tracks whose
    galleries fill up to NN_BUDGET = 60 samples (2 GMAC of cosine distances per frame): ids / states / counters identical to the
    oracle tracker in every frame, posterior means within 1e-9."""
    n_obj, n_frames = 256, 66
    frames = _dense_scenario(n_obj, n_frames, 77)
    ref = od.TrackerState(0.2, 60, max_iou_distance=0.6, max_age=30, n_init=3)
    eng = E.Engine(None, synth_reid(1702), precision="f32", max_crops=64, max_frame_hw=(360, 640), max_tracks=1024, nn_budget_cap=60)
    eng.set_option("dot_arena_mb", arena_mb)       # 0: track_batch_kernel<4,false>, appearance rows by MFMA inside the walk
    tid = eng.tracker_create(max_dist=0.2, max_iou_distance=0.6, max_age=30, n_init=3, nn_budget=60)
    for t, dets in enumerate(frames):
        ref.predict()
        ref.update(dets)
        eng.tracker_step(tid, np.array([d["tlwh"] for d in dets]), np.array([d["conf"] for d in dets]),
                         np.array([d["feature"] for d in dets], dtype=np.float32))
        s = eng.tracker_state(tid, with_cov=False)
        np.testing.assert_array_equal(s["ids"], [k.tid for k in ref.tracks], err_msg=f"frame {t}")
        np.testing.assert_array_equal(s["state"], [k.state for k in ref.tracks], err_msg=f"frame {t}")
        np.testing.assert_array_equal(s["tsu"], [k.tsu for k in ref.tracks], err_msg=f"frame {t}")
        np.testing.assert_array_equal(s["hits"], [k.hits for k in ref.tracks], err_msg=f"frame {t}")
        np.testing.assert_allclose(s["mean"], np.array([k.mean for k in ref.tracks]), rtol=1e-9, atol=1e-9)
    assert len(ref.tracks) >= 250 and max(len(g) for g in ref.gallery.values()) == 60
    assert sorted(s["gallery"][s["state"] == 2]) == sorted(len(ref.gallery[k.tid]) for k in ref.tracks if k.state == 2)
    eng.close()


def test_config2_videotracker_256_boxes_per_frame():
    """VideoTracker.run with 256 boxes in one frame (configs[2]: <= 256 detections per frame): crops + ReID for all of them in one
    launch, DeepSORT NMS, rows equal to the oracle's."""
    from oracle import reid as orr
    rsd = synth_reid(1702)
    embed = orr.make_embedder(rsd)
    Hh, Ww, n_obj, n_frames = 720, 1280, 256, 5
    rng = np.random.default_rng(5)
    frames = synth_frames(n_frames, Hh, Ww, n_obj=10, seed=8)
    cols = 20
    base = np.stack([(np.arange(n_obj) % cols) * 62.0 + 10, (np.arange(n_obj) // cols) * 54.0 + 8], 1)
    wh = rng.uniform([28, 28], [50, 46], (n_obj, 2))
    labels = rng.integers(0, 3, n_obj)
    eng = E.Engine(None, rsd, precision="f32", max_crops=512, max_frame_hw=(Hh, Ww), max_tracks=2048, nn_budget_cap=60)
    tids = [eng.tracker_create(max_dist=0.2, min_confidence=0.25, nms_max_overlap=0.5, max_iou_distance=0.6, max_age=30, n_init=3, nn_budget=60)
            for _ in range(3)]
    ovt = od.VideoTrackerOracle(3, TRACK_CFG, embed)
    total = 0
    for t in range(n_frames):
        xy = base + rng.normal(0, 0.5, base.shape) + t * 0.7
        xywh = np.concatenate([xy, wh], 1).astype(np.float64)
        scores = rng.uniform(0.3, 0.95, n_obj)
        ref = ovt.run(frames[t], xywh, labels, scores)
        got = eng.videotracker_run(tids, frames[t], xywh, labels, scores)
        ref_rows = np.array([list(b) + [tr, lb] for b, tr, lb in zip(ref["boxes"], ref["tracks"], ref["labels"])], dtype=np.int64).reshape(-1, 6)
        np.testing.assert_array_equal(got, ref_rows, err_msg=f"frame {t}")
        total += len(ref_rows)
    assert total > 400
    eng.close()
