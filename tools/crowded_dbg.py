"""Diagnosis tool: the crowded golden scenario (tests/golden/scenarios.py) step by step through vc_tracker_step against the oracle tracker,
printing the first frame / track where ids, states or means diverge (used to find the set-order dependence, DESIGN.md section 2 ii)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np
import scenarios
import vehicle_counting_amd.engine as E
from vehicle_counting_amd.weights import synth_reid
name = sys.argv[1] if len(sys.argv) > 1 else "crowded"
g = np.load(os.path.join("tests", "golden", f"tracker_{name}.npz"))
p, frames = scenarios.build(name)
eng = E.Engine(None, synth_reid(1702), precision="f32", max_crops=64, max_frame_hw=(360, 640), max_tracks=512, nn_budget_cap=60)
tid = eng.tracker_create(max_dist=p["max_dist"], max_iou_distance=p["max_iou_distance"], max_age=p["max_age"], n_init=p["n_init"], nn_budget=p["budget"])
for t, dets in enumerate(frames):
    eng.tracker_step(tid, np.array([d["tlwh"] for d in dets]).reshape(-1, 4), np.array([d["conf"] for d in dets]), np.array([d["feature"] for d in dets], dtype=np.float32).reshape(-1, 512))
    s = eng.tracker_state(tid, with_cov=False)
    ok = np.array_equal(s["ids"], g[f"f{t}_ids"]) and np.array_equal(s["state"], g[f"f{t}_state"]) and np.allclose(s["mean"], g[f"f{t}_mean"], rtol=1e-9, atol=1e-9)
    if not ok:
        print(name, "NO_REG" if os.environ.get("VC_TRACK_NO_REG") else "reg", "first mismatch at frame", t, "tracks", len(s["ids"]), "want", len(g[f"f{t}_ids"]))
        break
else:
    print(name, "NO_REG" if os.environ.get("VC_TRACK_NO_REG") else "reg", "all frames match")
