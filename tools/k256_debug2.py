"""Dense 256-rectangle scene, class by class through vc_tracker_step vs the oracle TrackerState on the SAME detections and features:
first frame whose state differs, with the kernel's own cost rows (vc_tracker_debug_costs) next to the oracle's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import vehicle_counting_amd.engine as E
from oracle import deepsort as od, reid as orr, yolov5 as oy
from vehicle_counting_amd.synth import synth_frames, synth_tracks
from vehicle_counting_amd.weights import synth_reid
import test_gpu_round3 as R
np.set_printoptions(linewidth=200, precision=9)
T, H, W, n_obj, nc = 12, 640, 640, 256, 3
frames = synth_frames(T, H, W, n_obj=n_obj, seed=1702, bounce=True)
det, cnt = R.injected(synth_tracks(T, H, W, n_obj=n_obj, seed=1702, bounce=True))
eng = E.Engine(None, synth_reid(1702), precision="f32", max_crops=512, max_frame_hw=(H, W), max_tracks=4096, nn_budget_cap=60)
c = 0
ref = od.TrackerState(0.2, 60, max_iou_distance=0.6, max_age=30, n_init=3)
tid = eng.tracker_create(**R.TRACK_KW)
for f in range(T):
    m = oy.marshal_like_reference(det[f])
    sel = m["classes"] == c
    xywh, sc = m["bboxes"][sel], m["scores"][sel]
    xyxy = xywh.copy(); xyxy[:, 2:] += xyxy[:, :2]
    cx = od.xyxy_to_cxcywh(xyxy)
    feats = eng.embed(frames[f], cx)
    tlwh = cx.copy(); tlwh[:, 0] -= cx[:, 2] / 2.0; tlwh[:, 1] -= cx[:, 3] / 2.0
    keepc = sc > 0.25
    tl, cf, ft = tlwh[keepc], sc[keepc], feats[keepc]
    keep = od.dsort_nms(tl, 0.5, cf)
    dets = [{"tlwh": tl[i].astype(np.float64), "conf": float(cf[i]), "feature": ft[i]} for i in keep]
    ref.predict()
    conf_idx = [i for i, k in enumerate(ref.tracks) if k.state == od.CONFIRMED]
    cand = [i for i, k in enumerate(ref.tracks) if not (k.state == od.CONFIRMED and k.tsu != 1)]
    cols = list(range(len(dets)))
    app_ref = ref._appearance_cost(dets, conf_idx, cols) if dets and conf_idx else np.zeros((0, len(dets)))
    iou_ref = ref._iou_cost(dets, cand, cols) if dets and cand else np.zeros((0, len(dets)))
    ids_before = [k.tid for k in ref.tracks]
    ref.update(dets)
    eng.tracker_step(tid, np.array([d["tlwh"] for d in dets]), np.array([d["conf"] for d in dets]), np.array([d["feature"] for d in dets], np.float32))
    app, iou = eng.tracker_debug_costs()
    s = eng.tracker_state(tid, with_cov=False)
    same = list(s["ids"]) == [k.tid for k in ref.tracks] and list(s["state"]) == [k.state for k in ref.tracks] and list(s["tsu"]) == [k.tsu for k in ref.tracks]
    dm = np.abs(s["mean"] - np.array([k.mean for k in ref.tracks])).max() if same and len(ref.tracks) else -1
    print(f"frame {f}: dets {len(dets)} tracks {len(ref.tracks)} same {same} mean diff {dm:.2e}", flush=True)
    if len(dets) and conf_idx:
        got = app[conf_idx]
        gate_same = np.array_equal(got == od.GATED_COST, app_ref == od.GATED_COST)
        op = app_ref != od.GATED_COST
        print("   appearance: gates equal", gate_same, "max |diff| on open entries", np.abs(got[op] - app_ref[op]).max() if op.any() else 0,
              "entries within 1e-5 of max_dist:", int((np.abs(app_ref[op] - 0.2) < 1e-5).sum()))
    if len(dets) and cand:
        print("   iou max |diff|", np.abs(iou[cand] - iou_ref).max(), "entries within 1e-9 of 0.6:", int((np.abs(iou_ref - 0.6) < 1e-9).sum()))
    if not same:
        a, b = list(s["ids"]), [k.tid for k in ref.tracks]
        print("   product ids", a[-12:]); print("   oracle  ids", b[-12:])
        pm = {i: tuple(np.round(mm, 3)) for i, mm in zip(s["ids"], s["mean"][:, :4])}
        om = {k.tid: tuple(np.round(k.mean[:4], 3)) for k in ref.tracks}
        for i in sorted(set(pm) | set(om)):
            if pm.get(i) != om.get(i): print("   track", i, "product", pm.get(i), "oracle", om.get(i))
        break
