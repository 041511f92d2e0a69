"""conv3x3s2_halo_kernel<..., F2> (YOLOv5s layer 3 + C3.cv1 | cv2 of layer 4 in one launch) against the two launches over random frame geometries, batch
sizes and tensor sizes: layers 4 / 6 / 17, layer 3 on demand and the detections bit for bit.  The stand-alone layer 3 is pinned to tile configuration 49 (the
fused kernel's own tile and K order; see tests/test_gpu_round5.py).  VC_SWEEP_N cases (default 24)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import vehicle_counting_amd.engine as E
from vehicle_counting_amd.weights import synth_yolo
rng = np.random.default_rng(int(os.environ.get("VC_SWEEP_SEED", 41)))
N = int(os.environ.get("VC_SWEEP_N", 24))
bad = ran = 0
for gi in range(N):
    size = int(rng.choice([320, 640, 1024]))
    H, W, B = int(rng.integers(90, 1100)), int(rng.integers(90, 1300)), int(rng.integers(1, 7))
    sd = synth_yolo("yolov5s", nc=5, seed=1702 + gi, det_scale=6.0, obj_shift=2.0)
    fr = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    eng = E.Engine(sd, None, precision="bf16", img_size=size, num_classes=5, max_batch=B, max_frame_hw=(H, W), max_candidates=8192)
    out = {}
    try:
        eng.set_option("fuse_s2_pw", 0)
        eng.detect(list(fr))
        lines = []
        for l in eng.tune_export().strip().splitlines():
            k, c = l.split()
            lines.append(f"{k} {'49' if '_ci64_co128_k3x3_s2_' in k else c}")
        eng.tune_import("\n".join(lines) + "\n")
        for on in (1, 0):
            eng.set_option("fuse_s2_pw", on)
            d = eng.detect(list(fr))
            out[on] = (d, [eng.debug_layer(l, batch=B) for l in (4, 6, 17, 3)])
    except Exception as ex:
        print("skip", size, (H, W, B), str(ex)[:70], flush=True)
        eng.close()
        continue
    same = all(np.array_equal(a, b) for a, b in zip(out[1][0], out[0][0])) and all(np.array_equal(a, b) for a, b in zip(out[1][1], out[0][1]))
    ran += 1
    print(("ok " if same else "BAD"), size, (H, W, B), "layer 3", out[1][1][3].shape, [len(x) for x in out[1][0]], flush=True)
    bad += 0 if same else 1
    eng.close()
print("SWEEP_OK" if bad == 0 and ran >= N // 2 else f"SWEEP_BAD {bad} of {ran}")
