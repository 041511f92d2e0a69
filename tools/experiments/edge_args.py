"""Edge arguments through the Python mirror / C ABI: empty inputs, tiny and extreme frames.  Every case must end in a result or a VcError
(never a crash); prints one line per case."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import vehicle_counting_amd.engine as E
from vehicle_counting_amd import _lib as L
from vehicle_counting_amd.pipeline import CountingPipeline, FrameSource
from vehicle_counting_amd.synth import synth_frames
from vehicle_counting_amd.weights import synth_reid, synth_yolo
nc = 4
sd, rsd = synth_yolo("yolov5s", nc=nc, seed=1702, det_scale=4.0, obj_shift=0.0), synth_reid(1702)
eng = E.Engine(sd, rsd, precision="bf16", num_classes=nc, max_batch=4, max_frame_hw=(360, 640), max_crops=256, max_tracks=256, nn_budget_cap=20)
tr = [eng.tracker_create(nn_budget=10) for _ in range(nc)]
rng = np.random.default_rng(0)
img = rng.integers(0, 256, (360, 640, 3), dtype=np.uint8)
def case(name, fn):
    try:
        r = fn()
        print("ok   ", name, "->", (type(r).__name__, getattr(r, "shape", None) or (len(r) if hasattr(r, "__len__") else r)))
    except L.VcError as ex:
        print("error", name, "->", str(ex)[:110])
    except (ValueError, AssertionError, IndexError, TypeError) as ex:
        print("pyerr", name, "->", type(ex).__name__, str(ex)[:90])
case("detect([])", lambda: eng.detect([]))
case("detect(1x1)", lambda: eng.detect([np.zeros((1, 1, 3), np.uint8)]))
case("detect(2x3)", lambda: eng.detect([rng.integers(0, 256, (2, 3, 3), dtype=np.uint8)]))
case("detect(8x640)", lambda: eng.detect([rng.integers(0, 256, (8, 640, 3), dtype=np.uint8)]))
case("detect(360x1)", lambda: eng.detect([rng.integers(0, 256, (360, 1, 3), dtype=np.uint8)]))
case("detect(too big)", lambda: eng.detect([np.zeros((361, 640, 3), np.uint8)]))
case("detect(5 images > max_batch)", lambda: eng.detect([img] * 5))
case("embed(k=0)", lambda: eng.embed(img, np.zeros((0, 4))))
case("embed(box outside)", lambda: eng.embed(img, np.array([[900.0, 900.0, 10.0, 10.0]])))
case("embed(1 px box)", lambda: eng.embed(img, np.array([[100.0, 100.0, 1.0, 1.0]])))
case("embed(2 px box)", lambda: eng.embed(img, np.array([[100.0, 100.0, 2.2, 2.2]])))
case("embed(nan box)", lambda: eng.embed(img, np.array([[np.nan, 100.0, 20.0, 20.0]])))
case("embed(huge box)", lambda: eng.embed(img, np.array([[1e12, 100.0, 1e13, 20.0]])))
case("embed_tensor(k=0)", lambda: eng.embed_tensor(np.zeros((0, 3, 50, 50), np.float32)))
dev = torch.from_numpy(np.stack([img] * 4)).cuda()
case("stream_submit(b=0)", lambda: eng.stream_submit(dev.data_ptr(), 0, 360, 640))
case("stream_run(b=0)", lambda: eng.stream_run(tr, dev.data_ptr(), 0, 360, 640))
eng.stream_reset()
case("stream_submit(b=5 > max_batch)", lambda: eng.stream_submit(dev.data_ptr(), 5, 360, 640))
case("stream_submit(h=0)", lambda: eng.stream_submit(dev.data_ptr(), 1, 0, 640))
case("stream_run without submit", lambda: eng.stream_run(tr, dev.data_ptr(), 4, 360, 640))
eng.stream_reset()
case("stream_collect with nothing in flight", lambda: eng.stream_collect())
case("videotracker_run(no boxes)", lambda: eng.videotracker_run(tr, img, np.zeros((0, 4)), np.zeros(0, np.int64), np.zeros(0)))
case("videotracker_run(label out of range)", lambda: eng.videotracker_run(tr, img, np.array([[10.0, 10.0, 30.0, 30.0]]), np.array([7]), np.array([0.9])))
case("videotracker_run(negative label)", lambda: eng.videotracker_run(tr, img, np.array([[10.0, 10.0, 30.0, 30.0]]), np.array([-1]), np.array([0.9])))
case("videotracker_run(zero-size box)", lambda: eng.videotracker_run(tr, img, np.array([[10.0, 10.0, 0.0, 0.0]]), np.array([1]), np.array([0.9])))
case("videotracker_run(nan box)", lambda: eng.videotracker_run(tr, img, np.array([[np.nan, 10.0, 20.0, 20.0]]), np.array([1]), np.array([0.9])))
case("deepsort_update(k=0)", lambda: eng.deepsort_update(tr[0], np.zeros((0, 4)), np.zeros(0), img))
case("tracker_step(nan)", lambda: eng.tracker_step(tr[0], np.array([[np.nan, 1.0, 2.0, 3.0]]), np.array([0.9]), np.zeros((1, 512), np.float32)))
case("tracker_state after nan", lambda: eng.tracker_state(tr[0])["ids"])
case("tracker_step(zero feature)", lambda: eng.tracker_step(tr[1], np.array([[5.0, 1.0, 20.0, 30.0]]), np.array([0.9]), np.zeros((1, 512), np.float32)))
cfg = types.SimpleNamespace(model_name="yolov5s", min_conf=0.25, min_iou=0.45, max_det=300)
args = types.SimpleNamespace(weight=None, mapping=None, output_path=None)
track = dict(MAX_DIST=0.2, MIN_CONFIDENCE=0.25, NMS_MAX_OVERLAP=0.5, MAX_IOU_DISTANCE=0.6, MAX_AGE=30, N_INIT=3, NN_BUDGET=10)
for t in tr: eng.tracker_destroy(t)
pipe = CountingPipeline(args, cfg, {"cam": {"cam_04": {"tracking_config": track}}}, engine=eng, class_names=[f"c{i}" for i in range(nc)])
zone = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "cam_04_halfres.json")
case("run_stream(0 frames)", lambda: pipe.run_stream(FrameSource(np.zeros((0, 360, 640, 3), np.uint8)), "cam_04", zone, batch=4, asynchronous=True))
case("run_stream(1 frame)", lambda: pipe.run_stream(FrameSource(synth_frames(1, 360, 640, 3, 1)), "cam_04", zone, batch=4, asynchronous=True))
case("run(0 frames)", lambda: pipe.run(FrameSource(np.zeros((0, 360, 640, 3), np.uint8)), "cam_04", zone))
print("EDGE_DONE")
