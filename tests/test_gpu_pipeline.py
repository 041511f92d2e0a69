"""GPU: end-to-end parity of the CSV artefact (SURVEY.md 8d ladder step 4) -- the oracle's per-video loop against the
product's reference-shaped loop (host frames) and its fused stream path (device frames), fp32 mode."""
import os
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import vehicle_counting_amd.engine as E  # noqa: E402
from oracle import pipeline as op  # noqa: E402
from vehicle_counting_amd.pipeline import CountingPipeline, FrameSource  # noqa: E402
from vehicle_counting_amd.synth import synth_frames  # noqa: E402
from vehicle_counting_amd.weights import synth_reid, synth_yolo  # noqa: E402

NC = 8
TRACK_CFG = dict(MAX_DIST=0.2, MIN_CONFIDENCE=0.25, NMS_MAX_OVERLAP=0.5, MAX_IOU_DISTANCE=0.6, MAX_AGE=30, N_INIT=3, NN_BUDGET=60)


def key(rows):
    return [(r["label"], r["track_id"], r["frame_id"], r["direction"], r["fframe"], r["lframe"]) for r in rows]


def test_csv_parity(golden_dir, tmp_path):
    T, H, W = 18, 360, 640
    frames = synth_frames(T, H, W, n_obj=6, seed=3)
    ysd, rsd = synth_yolo("yolov5s", nc=NC, seed=1702, det_scale=4.0, obj_shift=0.0), synth_reid(1702)
    zone = os.path.join(golden_dir, "cam_04_halfres.json")
    ref_rows, ref_counts, n_det = op.run_video(frames, ysd, rsd, TRACK_CFG, zone, nc=NC)
    assert sum(n_det) > 50 and len(ref_rows) > 10, (n_det, len(ref_rows))

    cfg = types.SimpleNamespace(model_name="yolov5s", min_conf=0.25, min_iou=0.45, max_det=300)
    args = types.SimpleNamespace(weight=None, mapping=None, output_path=str(tmp_path))
    cam_cfg = {"cam": {"cam_04": {"tracking_config": TRACK_CFG}}}
    for mode in ("loop", "pipelined", "stream", "stream_async", "stream_async_host", "stream_host", "frame_sharded"):
        eng = E.Engine(ysd, rsd, precision="f32", num_classes=NC, max_batch=8, max_frame_hw=(H, W), max_crops=512,
                       max_tracks=1024, nn_budget_cap=60)
        pipe = CountingPipeline(args, cfg, cam_cfg, engine=eng, class_names=[f"c{i}" for i in range(NC)])
        src = FrameSource(frames)
        if mode == "loop":
            rows, counts = pipe.run(src, "cam_04", zone)
        elif mode == "pipelined":                        # the reference's loop over the same loader, stage calls asynchronous (round 6)
            rows, counts = pipe.run_pipelined(src, "cam_04", zone)
        elif mode == "frame_sharded":                    # SURVEY.md 8f.1 driver on one rank: detect + embed + external-feature tracker
            rows, counts = pipe.run_frame_sharded(src, "cam_04", zone, chunk=4)
        else:                                           # "_host": frames stay in pinned host memory, staged two batches ahead (5 / 3 batches)
            rows, counts = pipe.run_stream(src, "cam_04", zone, batch=4 if "async" in mode else 8, asynchronous="async" in mode, host_frames=mode.endswith("_host"))
        # track_id / label / frame / direction / first-last frame exact; boxes within 1 px (int truncation of an fp64 state
        # that only depends on fp32-identical detections); fpoint/lpoint within 0.5
        assert key(rows) == key(ref_rows), mode
        for r, q in zip(rows, ref_rows):
            assert np.abs(np.array(r["box"]) - np.array(q["box"])).max() <= 1, (mode, r, q)
            assert np.abs(np.array(r["fpoint"]) - np.array(q["fpoint"])).max() <= 0.5
            assert np.abs(np.array(r["lpoint"]) - np.array(q["lpoint"])).max() <= 0.5
        assert counts == ref_counts, mode
        assert os.path.exists(os.path.join(str(tmp_path), "cam_04.csv"))
        eng.close()


def test_async_stream_same_rows_and_errors():
    """vc_stream_run_async / vc_stream_collect on the bf16 engine: the worker-thread tracker loop returns exactly the rows of the
    synchronous call (same engine, trackers reset in between), in order, one batch late; misuse is reported, not hung."""
    import torch
    from vehicle_counting_amd._lib import VcError
    B, H, W, NB = 8, 360, 640, 4
    frames = synth_frames(B * NB, H, W, n_obj=8, seed=11)
    ysd, rsd = synth_yolo("yolov5s", nc=NC, seed=1702, det_scale=4.0, obj_shift=0.0), synth_reid(1702)
    eng = E.Engine(ysd, rsd, precision="bf16", num_classes=NC, max_batch=B, max_frame_hw=(H, W), max_crops=B * 64, max_tracks=2048, nn_budget_cap=60)
    trk = [eng.tracker_create(max_dist=0.2, min_confidence=0.25, nms_max_overlap=0.5, max_iou_distance=0.6, max_age=30, n_init=3, nn_budget=60)
           for _ in range(NC)]
    dev = torch.from_numpy(frames).cuda()
    ptr = lambda i: dev[i * B:(i + 1) * B].data_ptr()
    sync_rows = []
    eng.stream_submit(ptr(0), B, H, W)
    for i in range(NB):
        if i + 1 < NB:
            eng.stream_submit(ptr(i + 1), B, H, W)
        rows, fidx, nd = eng.stream_run_packed(trk, ptr(i), B, H, W)
        sync_rows.append((rows, fidx, nd))
    assert sum(len(r[0]) for r in sync_rows) > 20
    for t in trk:
        eng.tracker_reset(t)
    with pytest.raises(VcError):
        eng._async_shapes = [(B, 512)]
        eng.stream_collect()                                  # nothing outstanding
    eng._async_shapes = []
    got = []
    eng.stream_submit(ptr(0), B, H, W)
    for i in range(NB):
        if i + 1 < NB:
            eng.stream_submit(ptr(i + 1), B, H, W)
        eng.stream_run_async(trk, ptr(i), B, H, W)
        if i > 0:
            got.append(eng.stream_collect())
    with pytest.raises(VcError):
        eng.stream_run_packed(trk, ptr(0), B, H, W)           # synchronous call while a batch is outstanding
    got.append(eng.stream_collect())
    for (r0, f0, n0), (r1, f1, n1) in zip(sync_rows, got):
        np.testing.assert_array_equal(n0, n1)
        np.testing.assert_array_equal(f0, f1)
        np.testing.assert_array_equal(r0, r1)
    eng.close()


def test_count_allgather_through_the_c_abi():
    """vc_comm_unique_id / vc_comm_init / vc_allgather_counts with one rank: RCCL on the engine's stream gathers the tensor with
    itself (the N-rank path is the same calls with world > 1; the driver's scaling run exercises it)."""
    from vehicle_counting_amd import parallel
    eng = E.Engine(None, synth_reid(1702), precision="bf16", max_crops=8, max_frame_hw=(64, 64), max_tracks=16, nn_budget_cap=4)
    x = np.arange(2 * 3 * 5, dtype=np.int32).reshape(2, 3, 5)
    out = parallel.allgather_counts_native(eng, x)
    np.testing.assert_array_equal(out, x)
    out = parallel.allgather_counts_native(eng, x + 7)           # communicator is reused
    np.testing.assert_array_equal(out, x + 7)
    eng.close()


def test_host_frames_ingest_same_rows():
    """vc_stream_submit_host / vc_stream_stage_host (pinned host frames copied on the engine's copy stream, four staging slots) return
    exactly the rows of the device-resident path, batch after batch, with the copies of later batches overlapping the work of earlier
    ones -- staged one batch ahead of the submission (the bench's host-frames path) as well as stage + submit in one call."""
    import torch
    B, H, W, NB = 8, 360, 640, 6
    frames = synth_frames(B * NB, H, W, n_obj=8, seed=13)
    ysd, rsd = synth_yolo("yolov5s", nc=NC, seed=1702, det_scale=4.0, obj_shift=0.0), synth_reid(1702)
    out = {}
    for mode in ("device", "host", "staged"):
        eng = E.Engine(ysd, rsd, precision="bf16", num_classes=NC, max_batch=B, max_frame_hw=(H, W), max_crops=B * 64, max_tracks=2048, nn_budget_cap=60)
        trk = [eng.tracker_create(max_dist=0.2, min_confidence=0.25, nms_max_overlap=0.5, max_iou_distance=0.6, max_age=30, n_init=3, nn_budget=60)
               for _ in range(NC)]
        dev = torch.from_numpy(frames).cuda()
        host = torch.from_numpy(frames).pin_memory()
        ptrs = {}

        def stage(i):
            ptrs[i] = eng.stream_stage_host(host[i * B:(i + 1) * B].data_ptr(), B, H, W)

        def submit(i):
            if mode == "host":
                ptrs[i] = eng.stream_submit_host(host[i * B:(i + 1) * B].data_ptr(), B, H, W)
            elif mode == "staged":                                  # the copy one batch further ahead than the detector
                eng.stream_submit(ptrs[i], B, H, W)
            else:
                ptrs[i] = dev[i * B:(i + 1) * B].data_ptr()
                eng.stream_submit(ptrs[i], B, H, W)

        got = []
        if mode == "staged":
            stage(0); stage(1)
        submit(0)
        for i in range(NB):
            if mode == "staged" and i + 2 < NB:
                stage(i + 2)
            if i + 1 < NB:
                submit(i + 1)
            eng.stream_run_async(trk, ptrs[i], B, H, W)
            if i > 0:
                got.append(eng.stream_collect())
        got.append(eng.stream_collect())
        out[mode] = got
        eng.close()
    assert sum(len(r[0]) for r in out["device"]) > 20
    for other in ("host", "staged"):
        for (r0, f0, n0), (r1, f1, n1) in zip(out["device"], out[other]):
            np.testing.assert_array_equal(n0, n1)
            np.testing.assert_array_equal(f0, f1)
            np.testing.assert_array_equal(r0, r1)
    # a fifth host batch while four are alive is refused (its staging slot still belongs to an uncollected batch)
    eng = E.Engine(ysd, rsd, precision="bf16", num_classes=NC, max_batch=B, max_frame_hw=(H, W), max_crops=B * 64, max_tracks=2048, nn_budget_cap=60)
    host = torch.from_numpy(frames).pin_memory()
    for i in range(4):
        eng.stream_stage_host(host[i * B:(i + 1) * B].data_ptr(), B, H, W)
    from vehicle_counting_amd._lib import VcError
    with pytest.raises(VcError):
        eng.stream_stage_host(host[4 * B:5 * B].data_ptr(), B, H, W)
    eng.stream_reset()
    eng.stream_stage_host(host[:B].data_ptr(), B, H, W)               # after a reset the slots are free again
    eng.close()
