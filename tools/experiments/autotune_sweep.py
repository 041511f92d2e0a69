"""Autotuned tile configurations against the untuned implicit-GEMM forms over random geometries / batch sizes / models.
Run twice (VC_AUTOTUNE=0 writes the reference layers to a file, the default run compares): the switch is read once per process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import vehicle_counting_amd.engine as E
from vehicle_counting_amd.weights import synth_yolo
rng = np.random.default_rng(5)
N = int(os.environ.get("VC_SWEEP_N", 8))
model = os.environ.get("VC_MODEL", "yolov5s")
size = int(os.environ.get("VC_SIZE", 640))
out = os.environ["VC_SWEEP_FILE"]
geoms = [(int(rng.integers(100, 1000)), int(rng.integers(100, 1300)), int(rng.integers(1, 9))) for _ in range(N)]
sd = synth_yolo(model, nc=5, seed=1702, det_scale=6.0, obj_shift=4.0)
LAYERS = (2, 4, 6, 9, 13, 17, 20, 23)
res = {}
for gi, (H, W, B) in enumerate(geoms):
    fr = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    eng = E.Engine(sd, None, precision="bf16", model_name=model, img_size=size, num_classes=5, max_batch=B, max_frame_hw=(H, W))
    d = eng.detect(list(fr))
    for l in LAYERS: res[f"{gi}_{l}"] = eng.debug_layer(l, batch=B)
    res[f"{gi}_n"] = np.array([len(x) for x in d])
    eng.close()
if os.environ.get("VC_AUTOTUNE") == "0":
    np.savez(out, **res); print("WROTE", len(res))
else:
    ref = np.load(out)
    worst = 0.0
    for gi, (H, W, B) in enumerate(geoms):
        for l in LAYERS:
            a, b = res[f"{gi}_{l}"], ref[f"{gi}_{l}"]
            assert a.shape == b.shape
            err = np.abs(a - b) / (np.abs(b) + 1.0)
            worst = max(worst, float(err.max()))
            if err.max() > 0.05: print("BAD", (H, W, B), l, float(err.max()), np.argwhere(err > 0.05)[:5].tolist())
        print((H, W, B), "dets", res[f"{gi}_n"].tolist(), ref[f"{gi}_n"].tolist())
    print("worst relative difference", worst)
