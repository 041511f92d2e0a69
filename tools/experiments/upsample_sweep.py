"""Upsample fold-in (conv_igemm_kernel<..., UP>) against upsample2x_kernel + conv over random frame geometries, batch sizes and models:
layers 13 / 17 / 20 / 23 and the detections bit for bit.  VC_SWEEP_N cases (default 24)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import vehicle_counting_amd.engine as E
from vehicle_counting_amd.weights import synth_yolo
rng = np.random.default_rng(int(os.environ.get("VC_SWEEP_SEED", 11)))
N = int(os.environ.get("VC_SWEEP_N", 24))
bad = 0
for gi in range(N):
    model = "yolov5s" if gi % 3 else "yolov5m"
    size = int(rng.choice([320, 640, 1024])) if model == "yolov5s" else int(rng.choice([320, 640]))
    H, W, B = int(rng.integers(90, 1100)), int(rng.integers(90, 1300)), int(rng.integers(1, 6))
    sd = synth_yolo(model, nc=5, seed=1702 + gi, det_scale=6.0, obj_shift=2.0)
    fr = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    eng = E.Engine(sd, None, precision="bf16", model_name=model, img_size=size, num_classes=5, max_batch=B, max_frame_hw=(H, W), max_candidates=8192)
    out = {}
    try:
        for on in (1, 0):
            eng.set_option("fuse_upsample", on)
            d = eng.detect(list(fr))
            out[on] = (d, [eng.debug_layer(l, batch=B) for l in (13, 17, 20, 23)])
    except Exception as ex:                     # (a random head that fires everywhere: not what is swept here)
        print("skip", model, size, (H, W, B), str(ex)[:60], flush=True)
        eng.close()
        continue
    same = all(np.array_equal(a, b) for a, b in zip(out[1][0], out[0][0])) and all(np.array_equal(a, b) for a, b in zip(out[1][1], out[0][1]))
    print(("ok " if same else "BAD"), model, size, (H, W, B), [len(x) for x in out[1][0]], flush=True)
    bad += 0 if same else 1
    eng.close()
print("SWEEP_OK" if bad == 0 else f"SWEEP_BAD {bad}")
