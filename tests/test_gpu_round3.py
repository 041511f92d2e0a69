"""GPU: parity holes of shipped code closed in round 3:
  * the batched stream path (vc_stream_submit / run_async / collect: ONE track_batch_kernel launch per batch) at 256 injected
    detections per frame, B = 32, fp32, rows == VideoTrackerOracle fed the same boxes -- the path bench.py's K256_injected times;
  * BASELINE.json configs[2] end to end: YOLOv5m 1024x1024, one clip through run_stream == oracle.pipeline.run_video;
  * rows of B = 128 == rows of B = 16 on the same 128-frame clip (fp32 in-process; bf16 with the tile family pinned);
  * checkpoint ingestion on the GPU: an UN-FUSED yolov5 `.pt` + `ckpt.t7` through ImageDetect(args.weight) / cam_config checkpoint
    == the oracle on independently folded parameters;
  * tracker row arena with steady 65..100 rows per frame and a tight cap; a batch that cannot be embedded is dropped, not wedged.
(track_batch_kernel<4,false>, the in-walk appearance instance, is covered by the `inwalk` parametrisation of tests/test_gpu_tracker.py
and tests/test_gpu_bench_config.py.)"""
import json
import os
import subprocess
import sys
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import vehicle_counting_amd.engine as E  # noqa: E402
from oracle import deepsort as od  # noqa: E402
from oracle import pipeline as op  # noqa: E402
from oracle import reid as orr  # noqa: E402
from oracle import yolov5 as oy  # noqa: E402
from vehicle_counting_amd.pipeline import CountingPipeline, FrameSource  # noqa: E402
from vehicle_counting_amd.synth import synth_frames, synth_tracks  # noqa: E402
from vehicle_counting_amd.weights import synth_reid, synth_yolo  # noqa: E402

TRACK_CFG = dict(MAX_DIST=0.2, MIN_CONFIDENCE=0.25, NMS_MAX_OVERLAP=0.5, MAX_IOU_DISTANCE=0.6, MAX_AGE=30, N_INIT=3, NN_BUDGET=60)
TRACK_KW = dict(max_dist=0.2, min_confidence=0.25, nms_max_overlap=0.5, max_iou_distance=0.6, max_age=30, n_init=3, nn_budget=60)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def key(rows):
    return [(r["label"], r["track_id"], r["frame_id"], r["direction"], r["fframe"], r["lframe"]) for r in rows]


def whole_frame_zone(golden_dir, tmp_path, W, H):
    with open(os.path.join(golden_dir, "cam_04_halfres.json")) as f:
        z = json.load(f)
    for sh in z["shapes"]:
        if sh["label"] == "zone":
            sh["points"] = [[0.0, 0.0], [float(W), 0.0], [float(W), float(H)], [0.0, float(H)]]
    path = str(tmp_path / "cam_04.json")
    with open(path, "w") as f:
        json.dump(z, f)
    return path


def injected(tracks):
    """synth_tracks() rectangles as detector rows [x1, y1, x2, y2, conf, cls] float32 (what vc_stream_inject takes)."""
    n = len(tracks[0][1])
    det = np.zeros((len(tracks), n, 6), np.float32)
    for f, (xywh, labels, scores) in enumerate(tracks):
        det[f, :, 0:2] = xywh[:, 0:2]
        det[f, :, 2:4] = xywh[:, 0:2] + xywh[:, 2:4]
        det[f, :, 4] = scores
        det[f, :, 5] = labels
    return det, np.full(len(tracks), n, np.int32)


def oracle_rows_for_injected(frames, det, nc, embed):
    """VideoTrackerOracle on the injected rows, marshalled like networks/yolo.py:72-97 (10-decimal JSON round trip, xywh)."""
    def compute():
        ovt = od.VideoTrackerOracle(nc, TRACK_CFG, embed)
        out = []
        for f in range(len(frames)):
            m = oy.marshal_like_reference(det[f])
            res = ovt.run(frames[f], m["bboxes"], m["classes"], m["scores"])
            out.append(np.array([list(b) + [tr, lb] for b, tr, lb in zip(res["boxes"], res["tracks"], res["labels"])], dtype=np.int64).reshape(-1, 6))
        return out
    # (tests/oracle_cache.py: 12 288 crops through the CPU ReID net took 79 s of every GPU-suite run; `embed` is always the oracle's
    # embedder of synth_reid(1702) here, named in the key)
    import oracle_cache
    return oracle_cache.memo("rows_for_injected", [frames, det, nc, {k: TRACK_CFG[k] for k in sorted(TRACK_CFG)}, "make_embedder(synth_reid(1702))"], compute)


def stream_rows(eng, tids, dev_frames, B, H, W, inject=None, cap_rows=512):
    """The batched asynchronous stream path over a whole clip: submit(i+1); run_async(i); collect(i-1) -- bench.py's loop."""
    T = dev_frames.shape[0]
    starts = list(range(0, T, B))
    out = [None] * T

    def submit(n):
        f0 = starts[n]
        b = min(B, T - f0)
        if inject is not None:
            eng.stream_inject(inject[0][f0:f0 + b], inject[1][f0:f0 + b])
        eng.stream_submit(dev_frames[f0:f0 + b].data_ptr(), b, H, W)

    def collect(n):
        rows, fidx, nd = eng.stream_collect()
        f0 = starts[n]
        b = min(B, T - f0)
        for f in range(b):
            out[f0 + f] = rows[fidx == f]
        return nd

    try:
        submit(0)
        for n, f0 in enumerate(starts):
            if n + 1 < len(starts):
                submit(n + 1)
            eng.stream_run_async(tids, dev_frames[f0:f0 + min(B, T - f0)].data_ptr(), min(B, T - f0), H, W, cap_rows=cap_rows)
            if n > 0:
                collect(n - 1)
        collect(len(starts) - 1)
    except Exception:
        eng.stream_reset()                     # a shared engine must not carry this clip's submissions into the next test
        raise
    return out


@pytest.fixture(scope="module")
def eng_f32():
    nc = 3
    e = E.Engine(synth_yolo("yolov5s", nc=nc, seed=1702, det_scale=4.0, obj_shift=-2.0), synth_reid(1702), precision="f32", num_classes=nc,
                 max_batch=32, max_frame_hw=(640, 640), max_crops=32 * 256, max_tracks=8192, nn_budget_cap=60)
    yield e
    e.close()


@pytest.fixture(scope="module")
def k256_case():
    T, H, W, n_obj, nc = 48, 640, 640, 256, 3
    frames = synth_frames(T, H, W, n_obj=n_obj, seed=1702, bounce=True)
    det, cnt = injected(synth_tracks(T, H, W, n_obj=n_obj, seed=1702, bounce=True))
    return frames, det, cnt, oracle_rows_for_injected(frames, det, nc, orr.make_embedder(synth_reid(1702)))


@pytest.mark.parametrize("arena_mb", [1024, 0])
def test_batched_stream_256_injected_detections_per_frame(eng_f32, k256_case, arena_mb):
    """256 injected detections per frame through submit / run_async / collect with B = 32 (two batches: the second
    one starts from device-resident tracker state with galleries already filled), fp32 ReID, rows identical to the oracle's
    VideoTracker.run per frame.  arena_mb = 0 repeats it on track_batch_kernel<4,false> (appearance rows computed in the walk)."""
    import torch
    T, B, H, W, nc = 48, 32, 640, 640, 3
    frames, det, cnt, ref = k256_case
    eng_f32.set_option("dot_arena_mb", arena_mb)
    tids = [eng_f32.tracker_create(**TRACK_KW) for _ in range(nc)]
    got = stream_rows(eng_f32, tids, torch.from_numpy(frames).cuda(), B, H, W, inject=(det, cnt), cap_rows=512)
    eng_f32.set_option("dot_arena_mb", 1024)
    n_rows = 0
    for f in range(T):
        # ids, labels AND boxes exact.  (This dense scene is what exposed the reference's dependence on CPython's set iteration order,
        # csrc/track_core.h::pyset_difference_order: with ascending order the ids of two new tracks swap in frame 7.)
        np.testing.assert_array_equal(got[f], ref[f], err_msg=f"frame {f}")
        n_rows += len(ref[f])
    assert n_rows > 40 * T / 2, n_rows                     # DeepSORT NMS thins the 256 overlapping rectangles; well over 20 rows per frame remain
    for t in tids:
        eng_f32.tracker_reset(t)


def test_row_arena_holds_steady_rows_with_a_tight_cap(eng_f32):
    """~70 confirmed tracks in every frame and cap_rows_per_frame = 100.  The kernel reserves output rows in
    chunks of 128 and abandons a chunk's remainder when a step does not fit, so it consumes up to twice the rows it emits; the arena
    used to be sized cap + one chunk and this batch failed with VC_ERR_CAPACITY although every frame fits the caller's cap."""
    import torch
    T, B, H, W, n = 32, 16, 640, 640, 70
    cols = 10
    base = np.stack([(np.arange(n) % cols) * 62.0 + 8, (np.arange(n) // cols) * 88.0 + 6], 1)
    det = np.zeros((T, n, 6), np.float32)
    for f in range(T):
        det[f, :, 0:2] = base + 0.5 * f
        det[f, :, 2:4] = det[f, :, 0:2] + np.array([44.0, 60.0])
        det[f, :, 4] = 0.9
        det[f, :, 5] = 1
    frames = synth_frames(T, H, W, n_obj=8, seed=9)
    tids = [eng_f32.tracker_create(**TRACK_KW) for _ in range(3)]
    got = stream_rows(eng_f32, tids, torch.from_numpy(frames).cuda(), B, H, W, inject=(det, np.full(T, n, np.int32)), cap_rows=100)
    assert [len(r) for r in got[:2]] == [0, 0] and all(len(r) == n for r in got[3:]), [len(r) for r in got]
    for t in tids:
        eng_f32.tracker_reset(t)


def test_unembeddable_batch_is_dropped_and_the_stream_continues(eng_f32):
    """a submission with a degenerate box (empty crop: the reference's cv2.resize raises there, Q4) fails ONCE, in the
    call that finds it; the next submission runs normally (it used to stay at the front of the queue and fail every later call)."""
    import torch
    B, H, W = 4, 640, 640
    frames = synth_frames(2 * B, H, W, n_obj=4, seed=5)
    dev = torch.from_numpy(frames).cuda()
    good, cnt = injected(synth_tracks(2 * B, H, W, n_obj=4, seed=5))
    bad = good[:B].copy()
    bad[2, 1, :4] = [100.2, 100.2, 100.4, 100.4]
    tids = [eng_f32.tracker_create(**TRACK_KW) for _ in range(3)]
    eng_f32.stream_inject(bad, cnt[:B])
    eng_f32.stream_submit(dev[:B].data_ptr(), B, H, W)
    with pytest.raises(E.L.VcError, match="empty crop"):
        eng_f32.stream_run_async(tids, dev[:B].data_ptr(), B, H, W)
    eng_f32._async_shapes.clear()
    eng_f32.stream_inject(good[B:], cnt[B:])
    eng_f32.stream_submit(dev[B:].data_ptr(), B, H, W)                 # the failed batch is gone: this one is the front of the queue
    rows, nd = eng_f32.stream_run(tids, dev[B:].data_ptr(), B, H, W)
    assert nd.tolist() == [4] * B
    eng_f32.stream_submit(dev[:B].data_ptr(), B, H, W)                 # and vc_stream_reset abandons a submission that was never run
    eng_f32.stream_reset()
    eng_f32.stream_inject(None)
    eng_f32.stream_submit(dev[B:].data_ptr(), B, H, W)
    eng_f32.stream_run(tids, dev[B:].data_ptr(), B, H, W)
    for t in tids:
        eng_f32.tracker_reset(t)


def test_lookahead_drop_does_not_lose_the_current_batch(eng_f32):
    """two submissions in flight, the SECOND cannot be embedded.  The calls made for the first batch (run_async,
    collect) succeed and return its rows; the error comes back from the call that consumes the second batch, once; a third batch
    then runs normally and Engine's bookkeeping (`_async_shapes`) stays in step with the native job queue."""
    import time

    import torch
    B, H, W = 4, 640, 640
    frames = synth_frames(3 * B, H, W, n_obj=4, seed=5)
    dev = torch.from_numpy(frames).cuda()
    good, cnt = injected(synth_tracks(3 * B, H, W, n_obj=4, seed=5))
    bad = good[B:2 * B].copy()
    bad[1, 2, :4] = [100.2, 100.2, 100.4, 100.4]
    tids = [eng_f32.tracker_create(**TRACK_KW) for _ in range(3)]
    p = [dev[k * B:(k + 1) * B].data_ptr() for k in range(3)]
    eng_f32.stream_inject(good[:B], cnt[:B]); eng_f32.stream_submit(p[0], B, H, W)
    eng_f32.stream_inject(bad, cnt[B:2 * B]); eng_f32.stream_submit(p[1], B, H, W)
    eng_f32.sync()                                                     # both detectors have finished: the look-ahead WILL try batch 1
    time.sleep(0.05)
    eng_f32.stream_run_async(tids, p[0], B, H, W)                      # batch 0: must not report batch 1's refusal
    rows0, fidx0, nd0 = eng_f32.stream_collect()                       # ... and its rows come back
    assert nd0.tolist() == [4] * B
    with pytest.raises(E.L.VcError, match="empty crop"):
        eng_f32.stream_run_async(tids, p[1], B, H, W)                  # reported here, by the call that consumes batch 1
    assert eng_f32._async_shapes == []
    eng_f32.stream_inject(good[2 * B:], cnt[2 * B:]); eng_f32.stream_submit(p[2], B, H, W)
    rows2, nd2 = eng_f32.stream_run(tids, p[2], B, H, W)
    assert nd2.tolist() == [4] * B
    eng_f32.stream_inject(None)
    for t in tids:
        eng_f32.tracker_reset(t)


def test_config2_yolov5m_1024_end_to_end_fp32(golden_dir, tmp_path):
    """BASELINE.json configs[2] on ONE stream: YOLOv5m at 1024x1024 -> NMS (max_det 256) -> ReID -> DeepSORT -> counting, through
    the fused stream path, fp32; CSV identical to oracle.pipeline.run_video (ids / frames / directions exact, boxes +-1 px)."""
    nc, S, T, B = 8, 1024, 12, 4
    sd = synth_yolo("yolov5m", nc=nc, seed=11, det_scale=3.0, obj_shift=0.0)
    frames = synth_frames(T, S, S, n_obj=20, seed=4)
    # the seeded weights of the deeper variants drift in activation scale: normalise the Detect inputs on frame 0 (as
    # tests/test_gpu_configs.py does) so that the synthetic head emits a few dozen boxes per frame
    x, _, _ = oy.preprocess([frames[0][:, :, ::-1]], S)
    _, ys0, _ = oy.forward(sd, x, "yolov5m", nc, return_layers=True)
    for i, layer in enumerate((17, 20, 23)):
        k = f"model.24.m.{i}.weight"
        sd[k] = (sd[k] / np.float32(np.sqrt((ys0[layer].numpy() ** 2).mean()))).astype(np.float32)
    rsd = synth_reid(1702)
    zone = whole_frame_zone(golden_dir, tmp_path, S, S)
    ref_rows, ref_counts, n_det = op.run_video(frames, sd, rsd, TRACK_CFG, zone, variant="yolov5m", nc=nc, max_det=256, size=S)
    assert sum(n_det) > 10 * T and max(n_det) <= 256 and len(ref_rows) > 20, (n_det, len(ref_rows))
    cfg = types.SimpleNamespace(model_name="yolov5m", min_conf=0.25, min_iou=0.45, max_det=256)
    args = types.SimpleNamespace(weight=None, mapping=None, output_path=str(tmp_path))
    eng = E.Engine(sd, rsd, precision="f32", model_name="yolov5m", num_classes=nc, img_size=S, max_batch=B, max_frame_hw=(S, S), max_det=256,
                   max_candidates=8192, max_crops=B * 256, max_tracks=4096, nn_budget_cap=60)
    pipe = CountingPipeline(args, cfg, {"cam": {"cam_04": {"tracking_config": TRACK_CFG}}}, engine=eng, class_names=[f"c{i}" for i in range(nc)])
    rows, counts = pipe.run_stream(FrameSource(frames), "cam_04", zone, batch=B, asynchronous=True)
    eng.close()
    assert key(rows) == key(ref_rows)
    for r, q in zip(rows, ref_rows):
        assert np.abs(np.array(r["box"]) - np.array(q["box"])).max() <= 1, (r, q)
    assert counts == ref_counts


_B128_SCRIPT = r"""
import sys
import numpy as np, torch
import vehicle_counting_amd.engine as E
from vehicle_counting_amd.synth import synth_frames
from vehicle_counting_amd.weights import synth_reid, synth_yolo
sys.path.insert(0, "tests")
from test_gpu_round3 import stream_rows, TRACK_KW
prec = sys.argv[1]
NC, T, H, W = 80, 128, 640, 640
ysd, rsd = synth_yolo("yolov5s", nc=NC, seed=1702, det_scale=4.0, obj_shift=1.0), synth_reid(1702)
frames = synth_frames(T, H, W, n_obj=12, seed=1702, bounce=True)
dev = torch.from_numpy(frames).cuda()
res = {}
for B in (16, 128):
    eng = E.Engine(ysd, rsd, precision=prec, num_classes=NC, max_batch=B, max_frame_hw=(H, W), max_crops=B * 64, max_tracks=8192, nn_budget_cap=60)
    tids = [eng.tracker_create(**TRACK_KW) for _ in range(NC)]
    res[B] = stream_rows(eng, tids, dev, B, H, W)
    eng.close()
n = 0
for f in range(T):
    assert np.array_equal(res[16][f], res[128][f]), (prec, f, res[16][f], res[128][f])
    n += len(res[16][f])
assert n > 100, n
print("B128_OK", prec, n)
"""


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_rows_of_b128_equal_rows_of_b16(prec):
    """bench.py steps 128 frames at a time, the parity tests 16 or fewer -- different conv size buckets, persistent
    grids, 104-task tracker walks.  The same 128-frame clip (bench.py's weights and frames) through the batched stream path with
    B = 16 (8 batches) and B = 128 (one batch): identical rows, frame by frame.  The run is a subprocess with VC_AUTOTUNE=0 so that
    the bf16 engines of both batch sizes take the same conv family per layer (the autotuner may pick the halo-staged 3x3 for one
    size bucket and the implicit GEMM for the other; they sum K in different orders)."""
    env = dict(os.environ, VC_AUTOTUNE="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("VC_TUNE_CACHE", None)
    r = subprocess.run([sys.executable, "-c", _B128_SCRIPT, prec], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "B128_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_unfused_checkpoints_through_the_loader_match_the_oracle(tmp_path):
    """SURVEY.md 8(f).2 on the GPU: an UN-FUSED ultralytics-style `.pt` (a pickled module tree of classes
    that are not importable here, Conv2d without bias + BatchNorm2d) handed to ImageDetect through args.weight, and a `ckpt.t7`
    ({'net_dict': un-fused ReID state_dict}, feature_extractor.py:13-14) through the cam_config's `checkpoint` -- the engine
    built from the loaded files reproduces the oracle run on parameters folded HERE with torch's own fuse_conv_bn_weights."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_checkpoint import _build_fake_upstream_model
    nc = 8
    raw = synth_yolo("yolov5s", nc=nc, seed=1702, det_scale=4.0, obj_shift=0.0, fused=False)
    model, _ = _build_fake_upstream_model(nc, 3)
    with torch.no_grad():
        for name, mod in model.named_modules():
            if isinstance(mod, torch.nn.Conv2d):
                mod.weight.copy_(torch.from_numpy(raw[name + ".weight"]))
                if mod.bias is not None:
                    mod.bias.copy_(torch.from_numpy(raw[name + ".bias"]))
            elif isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.copy_(torch.from_numpy(raw[name + ".weight"])); mod.bias.copy_(torch.from_numpy(raw[name + ".bias"]))
                mod.running_mean.copy_(torch.from_numpy(raw[name + ".running_mean"])); mod.running_var.copy_(torch.from_numpy(raw[name + ".running_var"]))
    ypath, rpath = str(tmp_path / "yolov5s_custom.pt"), str(tmp_path / "ckpt.t7")
    torch.save({"epoch": -1, "model": model, "ema": None}, ypath)
    for m in ("models", "models.common", "models.yolo"):
        sys.modules.pop(m)
    rsd = synth_reid(1702)
    torch.save({"net_dict": {k: torch.from_numpy(v) for k, v in rsd.items()}, "acc": 0.9, "epoch": 40}, rpath)
    # independent fold (torch's own), float64 -> float32
    folded = {}
    for name in [k[: -len(".weight")] for k in raw if k.endswith("conv.weight")]:
        bn = name[: -len("conv")] + "bn"
        w, b = torch.nn.utils.fusion.fuse_conv_bn_weights(
            torch.from_numpy(raw[name + ".weight"]).double(), None, torch.from_numpy(raw[bn + ".running_mean"]).double(),
            torch.from_numpy(raw[bn + ".running_var"]).double(), 1e-3, torch.from_numpy(raw[bn + ".weight"]).double(), torch.from_numpy(raw[bn + ".bias"]).double())
        folded[name + ".weight"], folded[name + ".bias"] = w.float().numpy(), b.float().numpy()
    for i in range(3):
        folded[f"model.24.m.{i}.weight"], folded[f"model.24.m.{i}.bias"] = raw[f"model.24.m.{i}.weight"], raw[f"model.24.m.{i}.bias"]
    from vehicle_counting_amd.detect import ImageDetect
    cfg = types.SimpleNamespace(model_name="yolov5s", min_conf=0.25, min_iou=0.45, max_det=300)
    args = types.SimpleNamespace(weight=ypath, mapping=None, precision="f32", output_path=str(tmp_path))
    det = ImageDetect(args, cfg, reid_checkpoint=rpath)
    assert det.engine.cfg.num_classes == nc
    frames = synth_frames(2, 360, 640, n_obj=6, seed=3)
    out = det.run({"imgs": [f[:, :, ::-1] for f in frames]})
    ref = oy.autoshape_detect(folded, [f[:, :, ::-1] for f in frames], "yolov5s", nc)
    for i in range(2):
        m = oy.marshal_like_reference(ref[i])
        assert len(m["bboxes"]) == len(out["boxes"][i]) > 0
        np.testing.assert_allclose(out["boxes"][i], m["bboxes"], atol=5e-2)
        np.testing.assert_allclose(out["scores"][i], m["scores"], atol=2e-4)
        np.testing.assert_array_equal(out["labels"][i], m["classes"])
    boxes = np.array([[100.3, 80.7, 60.2, 90.9], [320.0, 200.0, 50.0, 50.0], [300.5, 180.5, 101.0, 33.0]])
    got = det.engine.embed(frames[0], boxes)
    crops = []
    for b in boxes:
        x1, y1, x2, y2 = od.crop_corners(b, 640, 360)
        crops.append(frames[0][y1:y2, x1:x2])
    want = orr.make_embedder(rsd)(crops)
    np.testing.assert_allclose(got, want, atol=3e-5)
    det.engine.close()


def test_update_with_features_equals_deepsort_update():
    """DeepSort.update_with_features (embeddings supplied by the caller) against DeepSort.update on the same boxes: same rows."""
    from vehicle_counting_amd.track import DeepSort
    rsd = synth_reid(1702)
    eng = E.Engine(None, rsd, precision="f32", max_crops=64, max_frame_hw=(360, 640), max_tracks=256, nn_budget_cap=60)
    kw = dict(max_dist=0.2, min_confidence=0.25, nms_max_overlap=0.5, max_iou_distance=0.6, max_age=30, n_init=3, nn_budget=60)
    a, b = DeepSort(None, engine=eng, **kw), DeepSort(None, engine=eng, **kw)
    T, H, W = 10, 360, 640
    frames = synth_frames(T, H, W, n_obj=5, seed=5)
    n = 0
    for t, (xywh, labels, scores) in enumerate(synth_tracks(T, H, W, n_obj=5, seed=5)):
        xyxy = xywh.copy()
        xyxy[:, 2:] += xyxy[:, :2]
        bw, bh = xyxy[:, 2] - xyxy[:, 0], xyxy[:, 3] - xyxy[:, 1]
        feat = eng.embed(frames[t], np.stack([xyxy[:, 0] + bw / 2, xyxy[:, 1] + bh / 2, bw, bh], 1))
        r1 = np.asarray(a.update(xyxy, scores, frames[t]), np.int64).reshape(-1, 7)
        r2 = b.update_with_features(xyxy, scores, feat, H, W)
        np.testing.assert_array_equal(r1[:, :5], r2)
        n += len(r2)
    assert n > 20
    eng.close()


def test_multi_camera_batches_equal_separate_runs(golden_dir, tmp_path):
    """one engine, S = 4 cameras interleaved in every batch (vc_stream_run_async_multi: frame f is stepped on
    trackers[cam_of_frame[f]][label]) against four separate single-camera runs of the same clips: CSV rows and counts identical
    per camera (fp32: conv numerics do not depend on the tile configuration a batch size selects).  Cameras of different length:
    the exhausted ones drop out of the round-robin."""
    nc, H, W = 8, 360, 640
    ysd, rsd = synth_yolo("yolov5s", nc=nc, seed=1702, det_scale=4.0, obj_shift=0.0), synth_reid(1702)
    lens = [18, 18, 14, 9]
    clips = [synth_frames(n, H, W, n_obj=5 + c, seed=30 + c) for c, n in enumerate(lens)]
    zone = os.path.join(golden_dir, "cam_04_halfres.json")
    cfg = types.SimpleNamespace(model_name="yolov5s", min_conf=0.25, min_iou=0.45, max_det=300)
    args = types.SimpleNamespace(weight=None, mapping=None, output_path=None)
    names = [f"cam_{c:02d}" for c in range(4)]
    cam_cfg = {"cam": {n: {"tracking_config": TRACK_CFG} for n in names}}
    eng = E.Engine(ysd, rsd, precision="f32", num_classes=nc, max_batch=8, max_frame_hw=(H, W), max_crops=8 * 300, max_tracks=4096, nn_budget_cap=60)
    pipe = CountingPipeline(args, cfg, cam_cfg, engine=eng, class_names=[f"c{i}" for i in range(nc)])
    multi = pipe.run_streams([FrameSource(c) for c in clips], names, [zone] * 4, batch=8)
    total = 0
    for c in range(4):
        rows, counts = pipe.run_stream(FrameSource(clips[c]), names[c], zone, batch=4, asynchronous=True)
        assert key(multi[c][0]) == key(rows), c
        assert [r["box"] for r in multi[c][0]] == [r["box"] for r in rows], c
        assert multi[c][1] == counts, c
        total += len(rows)
    assert total > 30, total
    eng.close()
