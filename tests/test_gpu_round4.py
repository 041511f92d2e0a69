"""GPU: round-4 additions behind the C ABI --
  * conv autotune choices exported from one engine and imported into another (vc_tune_export / vc_tune_import): same text back, no
    re-timing, identical results;
  * executed-work accounting of the sparse Detect head (vc_profile_read / vc_profile_read_dense): the head's FLOPs follow the gathered
    row count, the dense figure is the reference's Detect.m[i] over every pixel;
  * cached op plans: alternating batch sizes / geometries on one engine give what a fresh engine gives;
  * the literal per-frame loop (CountingPipeline.run, batch 1, host frames) == the batched stream path, rows and counts."""
import os
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import vehicle_counting_amd.engine as E  # noqa: E402
from vehicle_counting_amd import _lib as L  # noqa: E402
from vehicle_counting_amd.pipeline import CountingPipeline, FrameSource  # noqa: E402
from vehicle_counting_amd.synth import synth_frames  # noqa: E402
from vehicle_counting_amd.weights import synth_reid, synth_yolo  # noqa: E402

NC = 8


def test_tune_export_import_roundtrip():
    sd = synth_yolo("yolov5s", nc=NC, seed=1702, det_scale=4.0, obj_shift=0.0)
    frames = synth_frames(2, 360, 640, n_obj=6, seed=3)
    imgs = [f[:, :, ::-1] for f in frames]
    a = E.Engine(sd, None, precision="bf16", num_classes=NC, max_batch=2, max_frame_hw=(360, 640))
    da = a.detect(imgs)                                           # times the tile candidates of every layer shape
    text = a.tune_export()
    lines = [l.split() for l in text.strip().splitlines()]
    assert len(lines) > 20 and all(len(l) == 2 and l[1].lstrip("-").isdigit() for l in lines)
    b = E.Engine(sd, None, precision="bf16", num_classes=NC, max_batch=2, max_frame_hw=(360, 640))
    b.tune_import(text)
    assert b.tune_export() == text                                 # adopted verbatim ...
    db = b.detect(imgs)
    assert b.tune_export() == text                                 # ... and nothing was added: no shape was timed again
    for x, y in zip(da, db):
        np.testing.assert_array_equal(x, y)                        # same kernel family per layer -> the same bits
    with pytest.raises(L.VcError):
        b.tune_import("p0_ci64_co64_k3x3_s1 9999\n")               # a tile configuration this build does not have
    with pytest.raises(L.VcError):
        b.tune_import("garbage without a number\n")
    a.close(); b.close()


def test_sparse_head_reports_executed_work():
    nc = 80
    sd = synth_yolo("yolov5s", nc=nc, seed=1702, det_scale=4.0, obj_shift=1.0)
    frames = synth_frames(4, 640, 640, n_obj=12, seed=1702)
    imgs = [f[:, :, ::-1] for f in frames]
    res = {}
    for sparse in (1, 0):
        eng = E.Engine(sd, None, precision="bf16", num_classes=nc, max_batch=4, max_frame_hw=(640, 640))
        eng.set_option("sparse_head", sparse)
        eng.detect(imgs)                                           # autotune outside the profiled pass
        eng.profile(True); eng.profile_reset()
        dets = eng.detect(imgs)
        conv = eng.profile_read(L.PROF_CONV)
        dense = eng.profile_read_dense(L.PROF_CONV)
        ops = eng.profile_ops()
        eng.profile(False)
        res[sparse] = (conv, dense, dets, ops)
        eng.close()
    (cs, ds, dets_s, ops_s), (cd, dd, dets_d, _) = res[1], res[0]
    for x, y in zip(dets_s, dets_d):
        np.testing.assert_array_equal(x, y)                        # the two heads agree (test_sparse_detect_head_equals_dense), so the work is comparable
    # dense engine: executed == dense credit; sparse engine: its dense credit equals the dense engine's executed work, its executed work is less
    assert abs(dd[0] - cd["flops"]) <= 1e-6 * cd["flops"] and abs(dd[1] - cd["bytes"]) <= 1e-6 * cd["bytes"]
    assert abs(ds[0] - cd["flops"]) <= 1e-6 * cd["flops"], (ds[0], cd["flops"])
    head_dense = sum(2.0 * 4 * (640 // s) ** 2 * 255 * c for s, c in ((8, 128), (16, 256), (32, 512)))
    assert cd["flops"] - cs["flops"] > 0.5 * head_dense            # most pixels carry no candidate: most of the head is not executed ...
    assert cs["flops"] > cd["flops"] - head_dense                  # ... the objectness rows and the gathered rows are
    for line in ops_s.strip().splitlines():                        # no launch above the chip's peak in the per-launch log
        assert float(line.split("tflops=")[1]) < 2500.0, line


def test_cached_plans_follow_shape_changes():
    sd = synth_yolo("yolov5s", nc=NC, seed=1702, det_scale=4.0, obj_shift=0.0)
    rsd = synth_reid(1702)
    fa = synth_frames(3, 360, 640, n_obj=6, seed=3)
    fb = synth_frames(2, 640, 640, n_obj=6, seed=4)
    eng = E.Engine(sd, rsd, precision="f32", num_classes=NC, max_batch=3, max_frame_hw=(640, 640), max_crops=64)
    seq = [[f[:, :, ::-1] for f in fa], [fb[0][:, :, ::-1]], [f[:, :, ::-1] for f in fa[:2]], [f[:, :, ::-1] for f in fb], [f[:, :, ::-1] for f in fa]]
    got = [eng.detect(imgs) for imgs in seq]
    boxes = np.array([[100.3, 80.7, 60.2, 90.9], [320.0, 200.0, 50.0, 50.0], [300.5, 180.5, 101.0, 33.0]])
    emb = [eng.embed(fa[0], boxes[:k]) for k in (3, 1, 2, 3, 1)]    # ReID plans per crop count, revisited
    eng.close()
    for imgs, g in zip(seq, got):
        fresh = E.Engine(sd, None, precision="f32", num_classes=NC, max_batch=3, max_frame_hw=(640, 640))
        ref = fresh.detect(imgs)
        fresh.close()
        for x, y in zip(g, ref):
            np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(emb[0], emb[3])
    np.testing.assert_array_equal(emb[1], emb[4])
    np.testing.assert_array_equal(emb[0][:2], emb[2])


def test_per_frame_loop_equals_stream_path(golden_dir):
    """bench.py's dropin_bs1 point times CountingPipeline.run; this holds its rows and counts equal to run_stream's on the same clip (bf16
    engine, one engine, so both paths run the same tile configurations)."""
    nc = 8
    sd, rsd = synth_yolo("yolov5s", nc=nc, seed=1702, det_scale=4.0, obj_shift=0.0), synth_reid(1702)
    frames = synth_frames(24, 360, 640, n_obj=6, seed=3)
    zone = os.path.join(golden_dir, "cam_04_halfres.json")
    cfg = types.SimpleNamespace(model_name="yolov5s", min_conf=0.25, min_iou=0.45, max_det=300)
    args = types.SimpleNamespace(weight=None, mapping=None, output_path=None)
    track = dict(MAX_DIST=0.2, MIN_CONFIDENCE=0.25, NMS_MAX_OVERLAP=0.5, MAX_IOU_DISTANCE=0.6, MAX_AGE=30, N_INIT=3, NN_BUDGET=60)
    eng = E.Engine(sd, rsd, precision="f32", num_classes=nc, max_batch=8, max_frame_hw=(360, 640), max_crops=512, max_tracks=1024, nn_budget_cap=60)
    pipe = CountingPipeline(args, cfg, {"cam": {"cam_04": {"tracking_config": track}}}, engine=eng, class_names=[f"c{i}" for i in range(nc)])
    rows_loop, counts_loop = pipe.run(FrameSource(frames), "cam_04", zone)
    rows_stream, counts_stream = pipe.run_stream(FrameSource(frames), "cam_04", zone, batch=8, asynchronous=True)
    eng.close()
    key = lambda rows: [(r["label"], r["track_id"], r["frame_id"], r["direction"], tuple(r["box"])) for r in rows]
    assert len(rows_loop) > 10 and key(rows_loop) == key(rows_stream)
    assert counts_loop == counts_stream


def test_front_fused_resize_random_geometries():
    """The letterbox resize folded into front_fused_kernel (round 4) against the separate letterbox + stem + conv launches over a seeded
    sweep of frame geometries: odd widths (row byte stride 3 W not a multiple of 4), portrait and landscape, down- and up-scaling, and
    sizes on either side of the kernel's staging limit (front_fused_resize_ok decides; both outcomes must give the same layer 1).
    Bit for bit, letterbox padding included."""
    import torch
    rng = np.random.default_rng(20260929)
    geoms = [(int(rng.integers(48, 1100)), int(rng.integers(48, 1400))) for _ in range(int(os.environ.get("VC_SWEEP_N", 20)))] + [(65, 1279), (1279, 65), (641, 639), (96, 96)]
    nc = 4
    sd = synth_yolo("yolov5s", nc=nc, seed=1702, det_scale=4.0, obj_shift=0.0)
    fused = 0
    for H, W in geoms:
        fr = rng.integers(0, 256, (2, H, W, 3), dtype=np.uint8)          # noise: every tap of every pixel matters
        eng = E.Engine(sd, None, precision="bf16", num_classes=nc, max_batch=2, max_frame_hw=(H, W))
        eng.detect([f[:, :, ::-1] for f in fr])
        a1 = eng.debug_layer(1, batch=2)
        dev = torch.from_numpy(fr).cuda()
        eng.stream_submit(dev.data_ptr(), 2, H, W)
        eng.sync()
        b1 = eng.debug_layer(1, batch=2)
        try:
            eng.debug_layer(0, batch=2)
        except E.L.VcError:
            fused += 1
        assert a1.shape == b1.shape, (H, W)
        assert np.array_equal(a1, b1), (H, W, float((a1 == b1).mean()))
        eng.close()
    assert fused >= len(geoms) // 2, fused


_FUSION_SWEEP = r"""
import os
import numpy as np
import vehicle_counting_amd.engine as E
from vehicle_counting_amd.synth import synth_frames
from vehicle_counting_amd.weights import synth_yolo
rng = np.random.default_rng(4)
N = int(os.environ.get("VC_SWEEP_N", 10))
geoms = [(int(rng.integers(200, 1000)), int(rng.integers(200, 1300)), int(rng.integers(1, 6))) for _ in range(N)] + [(640, 33, 2), (40, 640, 3)]
sd = synth_yolo("yolov5s", nc=5, seed=1702, det_scale=6.0, obj_shift=float(os.environ.get("VC_SWEEP_SHIFT", 4.0)))
ndet = 0
for H, W, B in geoms:
    fr = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    eng = E.Engine(sd, None, precision="bf16", num_classes=5, max_batch=B, max_frame_hw=(H, W))
    a = eng.detect(list(fr))
    la = [eng.debug_layer(l, batch=B) for l in (2, 4, 17, 23)]
    for o in ("c3_fused", "bneck_fused", "bneck_cv3", "front_fused", "sparse_head"):
        eng.set_option(o, 0)
    b = eng.detect(list(fr))
    lb = [eng.debug_layer(l, batch=B) for l in (2, 4, 17, 23)]
    for l, x, y in zip((2, 4, 17, 23), la, lb):
        assert np.array_equal(x, y), (H, W, B, l, float((x == y).mean()))
    assert len(a) == len(b) == B
    for x, y in zip(a, b):
        assert np.array_equal(x, y), (H, W, B)
        ndet += len(x)
    eng.close()
assert ndet > 0
print("SWEEP_OK", ndet)
"""


def test_detector_fusions_random_geometries():
    """Every fused detector kernel (c3_fused, bneck_fused with and without cv3, the sparse Detect head) against the one-launch-per-conv
    form over a seeded sweep of frame geometries and batch sizes, on noise frames: layers 2, 4, 17, 23 and the detections bit for bit.
    Own process with VC_AUTOTUNE=0 (the halo-staged 3x3 variants the autotuner may pick sum K slice-major, see
    test_bneck_fused_bit_identical in test_gpu_nets.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VC_AUTOTUNE="0", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("VC_TUNE_CACHE", None)
    r = subprocess.run([sys.executable, "-c", _FUSION_SWEEP], env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "SWEEP_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_embed_random_boxes_odd_frame():
    """Crop + resize + ReID embedding on a frame whose byte size is not a multiple of 4 (271 x 523), for seeded boxes that include the
    frame's corners (its very last pixel), slivers, boxes hanging over every edge and boxes larger than the frame: the fp32 engine
    against the oracle's embedder on the reference's crops (atol 3e-5, the bar of test_embed_crops), the network INPUT (resize on float
    data, /255, Normalize) against the oracle's bit for bit in both precisions, and the bf16 engine's two crop kernels
    (workgroup-per-crop, thread-per-pixel) bit for bit.  (Found by this sweep: the HIP headers' __fmul_rn / __fadd_rn are plain
    operators compiled with contraction allowed, and the backend fused the interpolation differently in the two kernels.)"""
    from oracle import reid as orr
    from oracle.deepsort import crop_corners
    rng = np.random.default_rng(11)
    H, W = 271, 523
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    boxes = [[W - 3.0, H - 3.0, 6.0, 6.0], [2.0, 2.0, 5.0, 5.0], [W - 1.5, 100.0, 3.0, 40.0], [100.0, H - 1.5, 60.0, 3.0],
             [W / 2, H / 2, 2.0 * W, 2.0 * H], [W / 2, H / 2, W, 2.0], [W - 25.0, H - 25.0, 50.0, 50.0]]
    for _ in range(41):
        w, h = rng.uniform(2, 300), rng.uniform(2, 260)
        boxes.append([rng.uniform(-20, W + 20), rng.uniform(-20, H + 20), w, h])
    keep = []
    for b in boxes:
        x1, y1, x2, y2 = crop_corners(np.asarray(b), W, H)
        if x2 > x1 and y2 > y1:
            keep.append(b)
    boxes = np.asarray(keep)
    assert len(boxes) >= 40
    sd = synth_reid(1702)
    crops = []
    for b in boxes:
        x1, y1, x2, y2 = crop_corners(b, W, H)
        crops.append(img[y1:y2, x1:x2])
    ref = orr.make_embedder(sd)(crops)
    import torch
    x_ref = orr.preprocess_crops(crops).transpose(0, 2, 3, 1)        # the network input: resize on float data, /255, Normalize
    eng = E.Engine(None, sd, precision="f32", max_crops=64, max_frame_hw=(H, W))
    np.testing.assert_allclose(eng.embed(img, boxes), ref, rtol=0, atol=3e-5)
    assert np.array_equal(eng.embed_input(len(boxes)), x_ref)         # bit for bit: every product and sum rounded like the oracle's
    eng.close()
    eng = E.Engine(None, sd, precision="bf16", max_crops=64, max_frame_hw=(H, W))
    a = eng.embed(img, boxes)
    xa = eng.embed_input(len(boxes))
    eng.set_option("crop_per_pixel", 1)
    b = eng.embed(img, boxes)
    assert np.array_equal(xa, torch.from_numpy(x_ref).to(torch.bfloat16).float().numpy())    # one RNE rounding of the fp32 value
    assert np.array_equal(xa, eng.embed_input(len(boxes)))
    assert np.array_equal(a, b)
    assert (a * ref).sum(1).min() >= 0.999
    eng.close()


def test_letterbox_random_geometries():
    """letterbox_kernel (AutoShape's letterbox: cv::resize INTER_LINEAR on 8-bit data in 11-bit fixed point, pad 114) against the oracle
    over a seeded sweep of frame and tensor geometries on noise, bit for bit: up- and down-scaling, extreme aspect ratios, 1-pixel-wide
    resized images, odd paddings."""
    from oracle import imageops as oi
    rng = np.random.default_rng(99)
    n = int(os.environ.get("VC_SWEEP_N", 16))
    geoms = [(int(rng.integers(8, 900)), int(rng.integers(8, 1200)), 32 * int(rng.integers(1, 21)), 32 * int(rng.integers(1, 21))) for _ in range(n)]
    geoms += [(8, 1200, 32, 640), (1200, 8, 640, 32), (641, 639, 640, 640), (31, 33, 640, 640)]
    for h, w, nh, nw in geoms:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        got = E.letterbox(img, nh, nw, "f32")
        ref = oi.letterbox(img, nh, nw).astype(np.float32) / np.float32(255)
        assert np.array_equal(got, ref), (h, w, nh, nw, float((got == ref).mean()))


_LOOP_STREAM_ODD = r"""
import os, types
import numpy as np
import vehicle_counting_amd.engine as E
from vehicle_counting_amd.pipeline import CountingPipeline, FrameSource
from vehicle_counting_amd.synth import synth_frames
from vehicle_counting_amd.weights import synth_reid, synth_yolo
nc = 8
sd, rsd = synth_yolo("yolov5s", nc=nc, seed=1702, det_scale=4.0, obj_shift=0.0), synth_reid(1702)
H, W = 273, 521                                                   # resize 640/521, byte size of a clip not a multiple of 4
frames = synth_frames(21, H, W, n_obj=5, seed=3)
zone = os.environ["VC_ZONE"]
cfg = types.SimpleNamespace(model_name="yolov5s", min_conf=0.25, min_iou=0.45, max_det=300)
args = types.SimpleNamespace(weight=None, mapping=None, output_path=None)
track = dict(MAX_DIST=0.2, MIN_CONFIDENCE=0.25, NMS_MAX_OVERLAP=0.5, MAX_IOU_DISTANCE=0.6, MAX_AGE=30, N_INIT=3, NN_BUDGET=60)
eng = E.Engine(sd, rsd, precision="bf16", num_classes=nc, max_batch=8, max_frame_hw=(H, W), max_crops=512, max_tracks=1024, nn_budget_cap=60)
pipe = CountingPipeline(args, cfg, {"cam": {"cam_04": {"tracking_config": track}}}, engine=eng, class_names=[f"c{i}" for i in range(nc)])
rows_loop, counts_loop = pipe.run(FrameSource(frames), "cam_04", zone)
rows_stream, counts_stream = pipe.run_stream(FrameSource(frames), "cam_04", zone, batch=8, asynchronous=True)   # 8 + 8 + 5 frames
eng.close()
key = lambda rows: [(r["label"], r["track_id"], r["frame_id"], r["direction"], tuple(r["box"])) for r in rows]
assert len(rows_loop) > 10, len(rows_loop)
assert key(rows_loop) == key(rows_stream)
assert counts_loop == counts_stream
print("LOOP_STREAM_OK", len(rows_loop))
"""


def test_per_frame_loop_equals_stream_path_bf16_odd_geometry(golden_dir):
    """The same comparison on the bf16 engine at a frame geometry that takes front_fused_kernel's resize mode, with a ragged last batch
    (21 frames in batches of 8).  Own process with VC_AUTOTUNE=0: batch 1 and batch 8 then run the same kernel family per layer (the
    autotuner chooses per shape, and its halo-staged 3x3 variants differ from the implicit-GEMM form in the last bf16 bit)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VC_AUTOTUNE="0", VC_ZONE=os.path.join(golden_dir, "cam_04_halfres.json"), PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("VC_TUNE_CACHE", None)
    r = subprocess.run([sys.executable, "-c", _LOOP_STREAM_ODD], env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "LOOP_STREAM_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_autotuned_tile_configurations_random_geometries(tmp_path):
    """Whatever tile configuration the autotuner picks (halo-staged 3x3 variants, direct 1x1 forms, persistent grids) at seeded random
    frame geometries and batch sizes, every checked layer stays within bf16 rounding noise of the untuned implicit-GEMM form
    (|a - b| / (|b| + 1) <= 2e-2; measured 8e-3): a tile hanging over a map edge or a wrong halo row would show as an O(1) error.
    tools/experiments/autotune_sweep.py, run once with VC_AUTOTUNE=0 (writes the layers) and once with the default (compares)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "tools", "experiments", "autotune_sweep.py")
    env = dict(os.environ, VC_SWEEP_N="5", VC_SWEEP_FILE=str(tmp_path / "layers.npz"), PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("VC_TUNE_CACHE", None)
    r = subprocess.run([sys.executable, script], env=dict(env, VC_AUTOTUNE="0"), cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "WROTE" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    env.pop("VC_AUTOTUNE", None)
    r = subprocess.run([sys.executable, script], env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "BAD" not in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    worst = float(r.stdout.strip().splitlines()[-1].split()[-1])
    assert worst <= 2e-2, worst


def test_stream_path_is_deterministic_run_to_run(tmp_path):
    """The asynchronous three-stream path on the benchmarked configuration (YOLOv5s 640 x 640, 80 classes, bf16, autotuned tiles, B = 32
    so that the sparse head's atomic gather, the NMS and the tracker walk all see many rows) run five times on one engine and once
    more on a fresh engine: every CSV row and every count identical.  Workgroup scheduling, the order in which the head's atomics
    append pixels and the overlap of batch i's tracker with batch i + 1's detector must not reach the result."""
    nc = 80
    sd, rsd = synth_yolo("yolov5s", nc=nc, seed=1702, det_scale=4.0, obj_shift=1.0), synth_reid(1702)
    frames = synth_frames(96, 640, 640, n_obj=12, seed=1702)
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "cam_04_halfres.json")) as f:
        z = json.load(f)
    for sh in z["shapes"]:                                            # the whole frame is the zone: every tracked row reaches the CSV
        if sh["label"] == "zone":
            sh["points"] = [[0.0, 0.0], [640.0, 0.0], [640.0, 640.0], [0.0, 640.0]]
    zone = str(tmp_path / "cam_04.json")
    with open(zone, "w") as f:
        json.dump(z, f)
    cfg = types.SimpleNamespace(model_name="yolov5s", min_conf=0.25, min_iou=0.45, max_det=300)
    args = types.SimpleNamespace(weight=None, mapping=None, output_path=None)
    track = dict(MAX_DIST=0.2, MIN_CONFIDENCE=0.25, NMS_MAX_OVERLAP=0.5, MAX_IOU_DISTANCE=0.6, MAX_AGE=30, N_INIT=3, NN_BUDGET=60)
    key = lambda rows: [(r["label"], r["track_id"], r["frame_id"], r["direction"], tuple(r["box"]), r["fframe"], r["lframe"]) for r in rows]
    runs = []
    for fresh in (False, True):
        eng = E.Engine(sd, rsd, precision="bf16", num_classes=nc, max_batch=32, max_frame_hw=(640, 640), max_crops=32 * 64, max_tracks=4096, nn_budget_cap=60)
        pipe = CountingPipeline(args, cfg, {"cam": {"cam_04": {"tracking_config": track}}}, engine=eng, class_names=[f"c{i}" for i in range(nc)])
        for _ in range(1 if fresh else 5):
            rows, counts = pipe.run_stream(FrameSource(frames), "cam_04", zone, batch=32, asynchronous=True)
            runs.append((key(rows), counts))
        eng.close()
    assert len(runs[0][0]) > 100
    for k, c in runs[1:]:
        assert k == runs[0][0] and c == runs[0][1]


def test_tracker_ids_are_given_back():
    """vc_tracker_destroy: the reference builds a new VideoTracker per video and drops the old one (modules/__init__.py:32-36); the
    drop-in's DeepSort gives its engine-side tracker back when it is closed or collected, so a process that walks a folder of videos
    with 80 classes does not hit max_trackers (256) on the fourth video (found by the determinism test above)."""
    from vehicle_counting_amd.track import VideoTracker
    eng = E.Engine(None, synth_reid(1702), precision="f32", max_crops=16, max_frame_hw=(360, 640), max_tracks=64, nn_budget_cap=10, max_trackers=4)
    ids = [eng.tracker_create(nn_budget=5) for _ in range(4)]
    with pytest.raises(L.VcError, match="max_trackers"):
        eng.tracker_create(nn_budget=5)
    rng = np.random.default_rng(0)
    f = rng.standard_normal((3, 512)).astype(np.float32)
    tlwh = np.array([[10, 10, 40, 60], [200, 100, 50, 50], [400, 200, 30, 80]], np.float64)
    for _ in range(3):
        eng.tracker_step(ids[1], tlwh, np.full(3, 0.9), f)
    assert len(eng.tracker_state(ids[1])["ids"]) == 3
    eng.tracker_destroy(ids[1])
    with pytest.raises(L.VcError, match="stale tracker handle"):
        eng.tracker_step(ids[1], tlwh, np.full(3, 0.9), f)
    with pytest.raises(L.VcError, match="stale tracker handle"):
        eng.tracker_destroy(ids[1])
    again = eng.tracker_create(nn_budget=7, max_age=5)
    assert again & 0xffff == ids[1] & 0xffff and again != ids[1]     # the slot again, under a new generation (tests/test_gpu_round5.py)
    assert len(eng.tracker_state(again)["ids"]) == 0            # a fresh tracker: no tracks, ids from 1
    eng.tracker_step(again, tlwh, np.full(3, 0.9), f)
    assert eng.tracker_state(again)["ids"].tolist() == [1, 2, 3]
    for t in [ids[0], again, ids[2], ids[3]]:                     # (ids[1] is a dead handle: its slot lives on as `again`)
        eng.tracker_destroy(t)
    cam = {"tracking_config": dict(MAX_DIST=0.2, MIN_CONFIDENCE=0.25, NMS_MAX_OVERLAP=0.5, MAX_IOU_DISTANCE=0.6, MAX_AGE=30, N_INIT=3, NN_BUDGET=10)}
    for _ in range(5):                                            # five "videos" of four classes on four tracker ids
        vt = VideoTracker(4, cam, None, engine=eng)
        assert sorted(t & 0xffff for t in vt.tracker_ids) == [0, 1, 2, 3]
        del vt                                                    # the finaliser queues the handles, the next tracker_create gives them back
    vt = VideoTracker(4, cam, None, engine=eng)
    vt.close(); vt.close()                                        # idempotent
    eng.close()


def test_folder_of_videos_on_one_pipeline(golden_dir):
    """The reference's driver walks a folder: one CountingPipeline, `run` per video, a new VideoTracker each time
    (modules/__init__.py:28-36).  Three videos of different geometry, length and batch raggedness through ONE drop-in pipeline / engine,
    the first one again at the end: its rows and counts are what they were the first time (no state of an earlier video -- tracks,
    galleries, ids, cached plans of another geometry, frames in flight -- reaches a later one), and track ids start from 1 per video."""
    nc = 8
    sd, rsd = synth_yolo("yolov5s", nc=nc, seed=1702, det_scale=4.0, obj_shift=0.0), synth_reid(1702)
    zone = os.path.join(golden_dir, "cam_04_halfres.json")
    cfg = types.SimpleNamespace(model_name="yolov5s", min_conf=0.25, min_iou=0.45, max_det=300)
    args = types.SimpleNamespace(weight=None, mapping=None, output_path=None)
    track = dict(MAX_DIST=0.2, MIN_CONFIDENCE=0.25, NMS_MAX_OVERLAP=0.5, MAX_IOU_DISTANCE=0.6, MAX_AGE=30, N_INIT=3, NN_BUDGET=60)
    eng = E.Engine(sd, rsd, precision="bf16", num_classes=nc, max_batch=8, max_frame_hw=(480, 640), max_crops=8 * 300, max_tracks=4096, nn_budget_cap=60,
                   max_trackers=2 * nc)
    pipe = CountingPipeline(args, cfg, {"cam": {"cam_04": {"tracking_config": track}}}, engine=eng, class_names=[f"c{i}" for i in range(nc)])
    videos = [synth_frames(20, 360, 640, n_obj=6, seed=3), synth_frames(13, 273, 521, n_obj=5, seed=4), synth_frames(9, 480, 352, n_obj=4, seed=5)]
    key = lambda rows: [(r["label"], r["track_id"], r["frame_id"], r["direction"], tuple(r["box"])) for r in rows]
    out = []
    for v in videos + [videos[0]]:
        rows, counts = pipe.run_stream(FrameSource(v), "cam_04", zone, batch=8, asynchronous=True)
        out.append((key(rows), counts))
    assert len(out[0][0]) > 10 and len(out[1][0]) > 0
    assert out[3] == out[0]
    assert min(k[1] for k in out[1][0]) <= 3                      # ids restart per video (a leftover tracker would continue from ~20)
    rows_loop, counts_loop = pipe.run(FrameSource(videos[1]), "cam_04", zone)      # and the per-frame loop afterwards, same engine
    assert len(rows_loop) > 0
    eng.close()


def test_a_failed_video_does_not_poison_the_next_one(golden_dir):
    """A video that exceeds a configured capacity (here: more boxes in a batch than max_crops) raises out of run_stream with batches
    still in flight.  The pipeline abandons them (vc_stream_reset) and gives the video's trackers back, so the next video on the same
    pipeline gives exactly what it gives on an engine that never saw the failure."""
    nc = 8
    quiet, busy = synth_yolo("yolov5s", nc=nc, seed=1702, det_scale=4.0, obj_shift=0.0), None
    rsd = synth_reid(1702)
    zone = os.path.join(golden_dir, "cam_04_halfres.json")
    cfg = types.SimpleNamespace(model_name="yolov5s", min_conf=0.25, min_iou=0.45, max_det=300)
    args = types.SimpleNamespace(weight=None, mapping=None, output_path=None)
    track = dict(MAX_DIST=0.2, MIN_CONFIDENCE=0.25, NMS_MAX_OVERLAP=0.5, MAX_IOU_DISTANCE=0.6, MAX_AGE=30, N_INIT=3, NN_BUDGET=60)
    good = synth_frames(20, 360, 640, n_obj=6, seed=3)
    crowded = synth_frames(24, 273, 521, n_obj=5, seed=4)          # three batches: the failure leaves submissions in flight
    key = lambda rows: [(r["label"], r["track_id"], r["frame_id"], r["direction"], tuple(r["box"])) for r in rows]
    res = []
    for fail_first in (True, False):
        eng = E.Engine(quiet, rsd, precision="f32", num_classes=nc, max_batch=8, max_frame_hw=(360, 640), max_crops=512, max_tracks=1024, nn_budget_cap=60,
                       max_trackers=nc)
        pipe = CountingPipeline(args, cfg, {"cam": {"cam_04": {"tracking_config": track}}}, engine=eng, class_names=[f"c{i}" for i in range(nc)])
        if fail_first:
            rng = np.random.default_rng(1)                      # 8 x 70 injected boxes per batch > max_crops = 512
            xy = rng.uniform(10, 200, (8, 70, 2)); wh = rng.uniform(20, 60, (8, 70, 2))
            det6 = np.concatenate([xy, xy + wh, np.full((8, 70, 1), 0.9), rng.integers(0, nc, (8, 70, 1))], 2).astype(np.float32)
            eng.stream_inject(det6, np.full(8, 70, np.int32))
            with pytest.raises(L.VcError, match="max_crops"):
                pipe.run_stream(FrameSource(crowded), "cam_04", zone, batch=8, asynchronous=True)
            eng.stream_inject()
        rows, counts = pipe.run_stream(FrameSource(good), "cam_04", zone, batch=8, asynchronous=True)   # max_trackers = nc: the failed video's ids are free again
        res.append((key(rows), counts))
        eng.close()
    assert len(res[0][0]) > 10 and res[0] == res[1]


def test_edge_arguments_end_in_a_result_or_an_error():
    """tools/experiments/edge_args.py: 33 calls with empty inputs, 1-pixel and 8 x 640 frames, boxes outside the frame, NaN boxes, zero-sized
    batches, calls out of order, zero-frame videos -- each ends in a result or a VcError with the argument named; none crashes the process
    (own process: a fault would otherwise take pytest down)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "experiments", "edge_args.py")], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "EDGE_DONE" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    out = r.stdout
    assert "pyerr" not in out, out
    for needle in ("error detect([]) -> libvcount_hip status 1", "error stream_submit(h=0) -> libvcount_hip status 1: frame size 0 x 640",
                   "error embed(box outside) -> libvcount_hip status 1: box 0 gives an empty crop", "ok    run_stream(0 frames)",
                   "error stream_collect with nothing in flight -> libvcount_hip status 3"):
        assert needle in out, (needle, out)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_multi_camera_random_layouts_equal_separate_runs(golden_dir, tmp_path, seed):
    """run_streams (S cameras interleaved in every batch, vc_stream_run_async_multi) against separate single-camera runs for seeded random
    layouts: 2 - 7 cameras, clip lengths 1 - 25 (cameras drop out of the round-robin at different times, one-frame clips included), batch
    sizes that do and do not divide the interleaved length, on one engine that has already served other layouts (fp32: conv numerics do
    not depend on the tile configuration a batch size selects)."""
    import json
    rng = np.random.default_rng(seed)
    nc, H, W = 6, 273, 521
    ysd, rsd = synth_yolo("yolov5s", nc=nc, seed=1702, det_scale=8.0, obj_shift=1.0), synth_reid(1702)
    with open(os.path.join(golden_dir, "cam_04_halfres.json")) as f:
        z = json.load(f)
    for sh in z["shapes"]:                                            # the whole frame is the zone: every tracked row reaches the CSV
        if sh["label"] == "zone":
            sh["points"] = [[0.0, 0.0], [float(W), 0.0], [float(W), float(H)], [0.0, float(H)]]
    zone = str(tmp_path / "zone.json")
    with open(zone, "w") as f:
        json.dump(z, f)
    cfg = types.SimpleNamespace(model_name="yolov5s", min_conf=0.25, min_iou=0.45, max_det=300)
    args = types.SimpleNamespace(weight=None, mapping=None, output_path=None)
    track = dict(MAX_DIST=0.2, MIN_CONFIDENCE=0.25, NMS_MAX_OVERLAP=0.5, MAX_IOU_DISTANCE=0.6, MAX_AGE=30, N_INIT=3, NN_BUDGET=60)
    eng = E.Engine(ysd, rsd, precision="f32", num_classes=nc, max_batch=8, max_frame_hw=(H, W), max_crops=8 * 300, max_tracks=4096, nn_budget_cap=60,
                   max_trackers=8 * nc)
    key = lambda rows: [(r["label"], r["track_id"], r["frame_id"], r["direction"], tuple(r["box"])) for r in rows]
    total = 0
    for _ in range(2):
        S = int(rng.integers(2, 8))
        lens = [int(rng.integers(1, 26)) for _ in range(S)]
        clips = [synth_frames(n, H, W, n_obj=3 + c % 4, seed=100 * seed + c) for c, n in enumerate(lens)]
        names = [f"cam_{c:02d}" for c in range(S)]
        pipe = CountingPipeline(args, cfg, {"cam": {n: {"tracking_config": track} for n in names}}, engine=eng, class_names=[f"c{i}" for i in range(nc)])
        multi = pipe.run_streams([FrameSource(c) for c in clips], names, [zone] * S, batch=int(rng.integers(1, 9)))
        for c in range(S):
            rows, counts = pipe.run_stream(FrameSource(clips[c]), names[c], zone, batch=int(rng.integers(1, 9)), asynchronous=bool(rng.integers(0, 2)))
            assert key(multi[c][0]) == key(rows), (S, lens, c)
            assert multi[c][1] == counts, (S, lens, c)
            total += len(rows)
    assert total >= 5, total                                      # (clips shorter than N_INIT frames contribute no rows)
    eng.close()


def test_head_side_stream_gives_the_same_detections():
    """The P3 / P4 Detect-head ops run on the engine's head stream beside the neck layers that follow them (Op::side, joined before the
    decode): detections and the head's gathered logits are what the single-stream order gives, for several batch sizes on one engine
    (cached plans) and back and forth between the two modes."""
    nc = 80
    sd = synth_yolo("yolov5s", nc=nc, seed=1702, det_scale=4.0, obj_shift=1.0)
    frames = synth_frames(6, 640, 640, n_obj=12, seed=1702)         # bench.py's workload: ~15 boxes per frame from the random head
    eng = E.Engine(sd, None, precision="bf16", num_classes=nc, max_batch=6, max_frame_hw=(640, 640))
    imgs = [f[:, :, ::-1] for f in frames]
    ref = {}
    for mode in (1, 0, 1, 0):
        eng.set_option("head_side", mode)
        for b in (6, 1, 3):
            d = eng.detect(imgs[:b])
            l23 = eng.debug_layer(23, batch=b)
            if b not in ref:
                ref[b] = (d, l23)
                assert sum(len(x) for x in d) > 0
            else:
                assert len(d) == len(ref[b][0])
                for x, y in zip(d, ref[b][0]):
                    np.testing.assert_array_equal(x, y)
                np.testing.assert_array_equal(l23, ref[b][1])
    eng.close()
