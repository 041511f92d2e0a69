"""The detector queue's round-5 tail kernels over random frame geometries, batch sizes, models and candidate loads: the sparse head with its
sixteen-lanes-per-anchor decode and per-run slot requests against the dense head's decode_kernel (one atomic per candidate), the NMS walk on one
wave, and the register form of the SPPF pools against the LDS-plane form -- detections bit for bit.  Candidate loads from a few per frame to
thousands (obj_shift), so the > 512-candidate path of the NMS walk (mask rows read from global memory) is covered.  VC_SWEEP_N cases (default 30)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import vehicle_counting_amd.engine as E
from vehicle_counting_amd.weights import synth_yolo
rng = np.random.default_rng(int(os.environ.get("VC_SWEEP_SEED", 29)))
N = int(os.environ.get("VC_SWEEP_N", 30))
bad = ran = 0
for gi in range(N):
    model = "yolov5s" if gi % 4 else "yolov5m"
    size = int(rng.choice([320, 640, 1024])) if model == "yolov5s" else int(rng.choice([320, 640]))
    H, W, B = int(rng.integers(90, 1100)), int(rng.integers(90, 1300)), int(rng.integers(1, 7))
    nc = int(rng.choice([1, 5, 17, 80]))
    shift = float(rng.choice([1.0, 2.0, 3.0, 4.0, 5.0, 6.0]))
    sd = synth_yolo(model, nc=nc, seed=1702 + gi, det_scale=6.0, obj_shift=shift)
    fr = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    eng = E.Engine(sd, None, precision="bf16", model_name=model, img_size=size, num_classes=nc, max_batch=B, max_frame_hw=(H, W), max_candidates=8192)
    out = {}
    try:
        for mode, (sparse, sep) in enumerate(((1, 1), (0, 1), (1, 0))):
            eng.set_option("sparse_head", sparse); eng.set_option("sppf_sep", sep)
            out[mode] = eng.detect(list(fr))
    except Exception as ex:                     # more than max_candidates boxes: a random head that fires everywhere is not what is swept here
        print("skip", model, size, (H, W, B), nc, shift, str(ex)[:60], flush=True)
        eng.close()
        continue
    same = all(np.array_equal(a, b) for m in (1, 2) for a, b in zip(out[0], out[m]))
    ran += 1
    print(("ok " if same else "BAD"), model, size, (H, W, B), "nc", nc, "shift", shift, "detections", [len(x) for x in out[0]], flush=True)
    bad += 0 if same else 1
    eng.close()
print("SWEEP_OK" if bad == 0 and ran >= N // 2 else f"SWEEP_BAD {bad} of {ran}")
