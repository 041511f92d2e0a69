"""GPU: round-5 additions behind the C ABI --
  * tracker handles carry a generation: a handle kept past vc_tracker_destroy is refused by every entry point, also once its slot
    belongs to a later tracker; a closed VideoTracker drives nothing; finalisers queue their handles instead of blocking;
  * vc_tune_import reaches op plans that have already run (cached plans resolve their tile configuration again);
  * ReID passes with more crops than the plan cache keeps (transient plans) equal the cached path;
  * conv3x3_halo_v2_kernel (tile configuration 55) bit for bit against conv3x3_halo_kernel on the detector's and the ReID net's shapes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import vehicle_counting_amd.engine as E  # noqa: E402
from vehicle_counting_amd import _lib as L  # noqa: E402
from vehicle_counting_amd.synth import synth_frames  # noqa: E402
from vehicle_counting_amd.weights import synth_reid, synth_yolo  # noqa: E402

NC = 8


def test_stale_tracker_handles_are_refused():
    from vehicle_counting_amd.track import VideoTracker
    eng = E.Engine(None, synth_reid(1702), precision="f32", max_crops=16, max_frame_hw=(360, 640), max_tracks=64, nn_budget_cap=10, max_trackers=2)
    rng = np.random.default_rng(0)
    f = rng.standard_normal((2, 512)).astype(np.float32)
    tlwh = np.array([[10, 10, 40, 60], [200, 100, 50, 50]], np.float64)
    old = eng.tracker_create(nn_budget=5)
    eng.tracker_step(old, tlwh, np.full(2, 0.9), f)
    eng.tracker_destroy(old)
    new = eng.tracker_create(nn_budget=5)                          # takes the slot the old tracker gave back
    assert new & 0xffff == old & 0xffff and new != old
    eng.tracker_step(new, tlwh[:1], np.full(1, 0.9), f[:1])
    for call in (lambda: eng.tracker_step(old, tlwh, np.full(2, 0.9), f), lambda: eng.tracker_state(old), lambda: eng.tracker_reset(old),
                 lambda: eng.tracker_destroy(old), lambda: eng.tracker_snapshot(old)):
        with pytest.raises(L.VcError, match="stale tracker handle"):
            call()
    assert eng.tracker_state(new)["ids"].tolist() == [1]           # the stale calls touched nothing
    frame = np.zeros((360, 640, 3), np.uint8)
    with pytest.raises(L.VcError):
        eng.videotracker_run([old], frame, np.array([[10., 10., 40., 60.]]), np.array([0]), np.array([0.9]))
    eng.tracker_destroy(new)
    cam = {"tracking_config": dict(MAX_DIST=0.2, MIN_CONFIDENCE=0.25, NMS_MAX_OVERLAP=0.5, MAX_IOU_DISTANCE=0.6, MAX_AGE=30, N_INIT=3, NN_BUDGET=10)}
    vt = VideoTracker(2, cam, None, engine=eng)
    handles = list(vt.tracker_ids)
    vt.close()
    with pytest.raises(RuntimeError, match="after close"):
        vt.run(frame, np.array([[10., 10., 40., 60.]]), np.array([0]), np.array([0.9]))
    vt2 = VideoTracker(2, cam, None, engine=eng)                   # the next video's trackers: the same slots, other handles
    assert sorted(t & 0xffff for t in vt2.tracker_ids) == sorted(t & 0xffff for t in handles) and not set(vt2.tracker_ids) & set(handles)
    del vt2                                                        # finaliser: queued, not destroyed ...
    assert len(eng._deferred_destroy) == 2
    vt3 = VideoTracker(2, cam, None, engine=eng)                   # ... until the next tracker is created (max_trackers = 2: it would not fit otherwise)
    assert len(eng._deferred_destroy) == 0
    vt3.close()
    eng.close()


def test_tune_import_reaches_plans_that_have_run():
    sd = synth_yolo("yolov5s", nc=NC, seed=1702, det_scale=4.0, obj_shift=0.0)
    frames = synth_frames(2, 352, 640, n_obj=6, seed=3)
    imgs = [f[:, :, ::-1] for f in frames]

    def cfgs(eng):                                                  # tile configuration of every plain conv launch of one pass
        eng.profile(True); eng.profile_reset(); eng.detect(imgs); log = eng.profile_ops(); eng.profile(False)
        return [int(l.split("cfg=")[1].split()[0]) for l in log.splitlines() if l.startswith("conv")]

    a = E.Engine(sd, None, precision="bf16", num_classes=NC, max_batch=2, max_frame_hw=(352, 640))
    a.detect(imgs)                                                  # plans built, every op's choice resolved
    before = cfgs(a)
    text = a.tune_export()
    # every 3 x 3 / s1 layer that chose anything but the 128 x 128 implicit GEMM (configuration 2: a halo-staged kernel, a split-K tile at this
    # small batch) is sent there instead
    lines, changed = [], 0
    for l in text.strip().splitlines():
        k, c = l.split()
        if "_k3x3_s1_" in k and int(c) != 2:
            c, changed = "2", changed + 1
        lines.append(f"{k} {c}")
    assert changed > 0
    a.tune_import("\n".join(lines) + "\n")
    after = cfgs(a)
    assert len(after) == len(before)
    assert not any(c in (28, 29, 30, 31, 36, 37, 38, 39, 55) for c in after) and any(b != c for b, c in zip(before, after))
    a.close()


def test_reid_transient_plans_equal_cached_plans():
    eng = E.Engine(None, synth_reid(1702), precision="f32", max_crops=640, max_frame_hw=(360, 640))
    rng = np.random.default_rng(5)
    frame = rng.integers(0, 255, (360, 640, 3), dtype=np.uint8)
    n = 300                                                         # > VC_REID_PLAN_CACHE_MAX_K: a transient plan per call
    cx, cy = rng.uniform(40, 600, n), rng.uniform(40, 320, n)
    boxes = np.stack([cx, cy, rng.uniform(10, 60, n), rng.uniform(10, 60, n)], 1)
    big1 = eng.embed(frame, boxes)
    big2 = eng.embed(frame, boxes)
    np.testing.assert_array_equal(big1, big2)
    small = np.concatenate([eng.embed(frame, boxes[i:i + 100]) for i in range(0, n, 100)])     # cached plans (k = 100)
    np.testing.assert_allclose(big1, small, rtol=0, atol=1e-6)      # fp32: the same fmaf chains whatever the batch
    eng.close()


@pytest.mark.parametrize("shape", [(16, 40, 40, 128, 128, 1, 2), (16, 20, 20, 256, 256, 1, 2), (96, 13, 13, 128, 128, 2, 1), (200, 7, 7, 256, 256, 2, 1),
                                   (300, 4, 4, 512, 512, 2, 1), (3, 38, 50, 64, 192, 1, 0)])
def test_halo_v2_equals_halo_kernel(shape, monkeypatch):
    B, H, W, Ci, Co, act, rm = shape
    rng = np.random.default_rng(H * 131 + Ci)
    x = rng.standard_normal((B, H, W, Ci), dtype=np.float32)
    w = (rng.standard_normal((Co, Ci, 3, 3), dtype=np.float32) / np.sqrt(Ci * 9)).astype(np.float32)
    b = rng.standard_normal(Co, dtype=np.float32) * 0.1
    res = rng.standard_normal((B, H, W, Co), dtype=np.float32) if rm else None
    out = {}
    for cfg in (30, 55):
        monkeypatch.setenv("VC_CONV_CFG", str(cfg))
        out[cfg] = E.conv2d(x, w, b, stride=1, pad=1, act=act, res=res, res_mode=rm, precision="bf16")
    np.testing.assert_array_equal(out[30], out[55])


def test_upsample_folded_into_its_consumer_gives_the_same_bits():
    """nn.Upsample + Concat in front of C3.cv1 | cv2 (YOLOv5 layers 11-13, 15-17) read by conv_igemm_kernel<..., UP> straight from the
    half-size map: layers 13 / 17 / 20 / 23 and the detections equal the pass with upsample2x_kernel, and the concat layers a debug read
    asks for are produced on demand."""
    sd = synth_yolo("yolov5s", nc=NC, seed=1702, det_scale=4.0, obj_shift=0.5)
    # both orders of the switch: an op caches its tile choice at its first launch, and a choice made for the plain kernel need not exist
    # for the fold-in (vc_engine_set_option sends every cached op back to the autotuner's table)
    for (B, H, W), order in (((3, 352, 640), (1, 0, 1)), ((2, 640, 640), (0, 1, 0)), ((1, 333, 500), (1, 0, 1)), ((2, 384, 640), (0, 1, 0))):
        frames = synth_frames(B, H, W, n_obj=6, seed=7)
        imgs = [f[:, :, ::-1] for f in frames]
        eng = E.Engine(sd, None, precision="bf16", num_classes=NC, max_batch=B, max_frame_hw=(H, W))
        out = {}
        for on in order:
            eng.set_option("fuse_upsample", on)
            dets = eng.detect(imgs)
            layers = {l: eng.debug_layer(l, batch=B) for l in (13, 17, 20, 23, 12, 16)}
            if on in out:
                continue
            out[on] = (dets, layers)
        for a, b in zip(out[1][0], out[0][0]):
            np.testing.assert_array_equal(a, b)
        for l in out[1][1]:
            np.testing.assert_array_equal(out[1][1][l], out[0][1][l])
        eng.close()


def test_halo_v2_random_geometries(monkeypatch):
    """Seeded sweep over what a caller can vary: map sizes down to 3 x 4 and up to 64 wide (tiles of one row up to tiles that hold several
    images), ragged last tiles, 2 - 6 channel slices, ragged channel tails, every activation / residual mode -- conv3x3_halo_v2_kernel against
    conv3x3_halo_kernel bit for bit wherever the launcher accepts the shape (it must accept the detector's and the ReID net's)."""
    rng = np.random.default_rng(20250929)
    ran = 0
    for case in range(40):
        W = int(rng.choice([4, 5, 7, 13, 16, 20, 25, 31, 40, 48, 64]))
        H = int(rng.integers(3, 41))
        B = int(rng.integers(1, 7))
        Ci = int(rng.choice([64, 128, 192, 256, 384]))
        Co = int(rng.choice([8, 40, 64, 128, 136, 256]))
        act, rm = int(rng.integers(0, 3)), int(rng.integers(0, 3))
        rows = min(256 // W, 320 // W - 2, B * H)
        if rows < 1 or rows * W < 160:                              # conv_halo_v2.hip::v2_applicable
            continue
        x = rng.standard_normal((B, H, W, Ci), dtype=np.float32)
        w = (rng.standard_normal((Co, Ci, 3, 3), dtype=np.float32) / np.sqrt(Ci * 9)).astype(np.float32)
        b = rng.standard_normal(Co, dtype=np.float32) * 0.1
        res = rng.standard_normal((B, H, W, Co), dtype=np.float32) if rm else None
        out = {}
        for cfg in (30, 55):
            monkeypatch.setenv("VC_CONV_CFG", str(cfg))
            out[cfg] = E.conv2d(x, w, b, stride=1, pad=1, act=act, res=res, res_mode=rm, precision="bf16")
        np.testing.assert_array_equal(out[30], out[55], err_msg=f"case {case}: B {B} H {H} W {W} Ci {Ci} Co {Co} act {act} res {rm}")
        ran += 1
    assert ran >= 15


@pytest.mark.parametrize("precision", ["bf16", "f32", "fp8"])
def test_sppf_register_form_gives_the_same_bits(precision):
    """SPPF's three chained MaxPool2d(5, 1, 2) (ultralytics/yolov5 v6.0 models/common.py::SPPF, layer 9 of the detector
    /root/reference/networks/yolo.py:58 loads) as row / column window maxima in registers (sppf_pool_sep_kernel) against the LDS-plane kernels
    it replaced: layer 9 (SPPF.cv2 reads all four slices of the concat) and the detections are the same bits, on square, letterboxed (12 x 20,
    14 x 20) and 1280^2 (40 x 40) planes."""
    sd = synth_yolo("yolov5s", nc=NC, seed=1702, det_scale=4.0, obj_shift=0.5)
    for (B, H, W, size) in ((3, 640, 640, 640), (2, 360, 640, 640), (1, 333, 500, 640), (1, 1280, 1280, 1280), (1, 720, 1280, 1280)):
        frames = synth_frames(B, H, W, n_obj=6, seed=11)
        imgs = [f[:, :, ::-1] for f in frames]
        eng = E.Engine(sd, None, precision=precision, num_classes=NC, img_size=size, max_batch=B, max_frame_hw=(H, W))
        out = {}
        for on in (1, 0):
            eng.set_option("sppf_sep", on)
            dets = eng.detect(imgs)
            out[on] = (dets, eng.debug_layer(9, batch=B))
        for a, b in zip(out[1][0], out[0][0]):
            np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(out[1][1], out[0][1])
        assert np.isfinite(out[1][1]).all() and np.abs(out[1][1]).max() > 0
        eng.close()


def test_s2_conv_with_its_pointwise_reader_gives_the_same_bits():
    """YOLOv5s layer 3 (Conv 64 -> 128, 3x3 / s2: ultralytics/yolov5 v6.0 models/yolo.py as loaded by /root/reference/networks/yolo.py:58) and
    C3.cv1 | cv2 of layer 4, its only reader, in one launch (conv3x3s2_halo_kernel<..., F2>): layers 4, 6, 9, 17 and the detections equal the
    two-launch pass bit for bit, and layer 3 itself -- never written by the fused pass -- is produced on demand for vc_detect_debug_layer.
    The stand-alone layer 3 is pinned to the halo-staged stride-2 kernel (tile configuration 49): that family walks K by tap parity class, the
    implicit GEMM tap by tap, and the two orders round differently (every tile-configuration choice of a 3 x 3 / s2 layer has that effect)."""
    sd = synth_yolo("yolov5s", nc=NC, seed=1702, det_scale=4.0, obj_shift=0.5)
    for (B, H, W) in ((3, 352, 640), (2, 640, 640), (1, 333, 500), (5, 96, 160)):
        frames = synth_frames(B, H, W, n_obj=6, seed=13)
        imgs = [f[:, :, ::-1] for f in frames]
        eng = E.Engine(sd, None, precision="bf16", num_classes=NC, max_batch=B, max_frame_hw=(H, W))
        eng.set_option("fuse_s2_pw", 0)
        eng.detect(imgs)
        lines, pinned = [], 0
        for l in eng.tune_export().strip().splitlines():
            k, c = l.split()
            if "_ci64_co128_k3x3_s2_" in k:
                c, pinned = "49", pinned + 1
            lines.append(f"{k} {c}")
        assert pinned > 0
        eng.tune_import("\n".join(lines) + "\n")
        out = {}
        for on in (1, 0, 1):
            eng.set_option("fuse_s2_pw", on)
            dets = eng.detect(imgs)
            layers = {l: eng.debug_layer(l, batch=B) for l in (4, 6, 9, 17, 3)}
            if on in out:
                continue
            out[on] = (dets, layers)
        for a, b in zip(out[1][0], out[0][0]):
            np.testing.assert_array_equal(a, b)
        for l in out[1][1]:
            np.testing.assert_array_equal(out[1][1][l], out[0][1][l], err_msg=f"layer {l} at {B}x{H}x{W}")
        assert np.abs(out[1][1][3]).max() > 0
        eng.close()
