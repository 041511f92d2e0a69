"""Generate the golden fixtures in tests/golden/ by running the REFERENCE code (read-only tree).

Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Outputs (small, committed): kalman.npz, dsort_nms.npz, iou_cost.npz, cosine.npz, assignment.npz,
tracker_<scenario>.npz, reid_forward.npz, counting.json.  Inputs are either stored next to the
expected outputs or regenerated from seeds (scenarios.py, vehicle-counting_amd/weights.py).
"""
import io
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import _refimport  # noqa: E402
import scenarios  # noqa: E402

sort = _refimport.load_sort()
rng = np.random.default_rng(1702)       # the reference's own seed constant (utilities/random_seed.py:5)


def gen_kalman():
    kf = sort.kalman_filter.KalmanFilter()
    out = {}
    meas = np.stack([rng.uniform(0, 1200, 64), rng.uniform(0, 700, 64), rng.uniform(0.3, 2.5, 64),
                     rng.uniform(20, 300, 64)], 1)
    init_m, init_c, pred_m, pred_c, proj_m, proj_c, upd_m, upd_c, gate = [], [], [], [], [], [], [], [], []
    zs = []
    gate_in = []
    for z in meas:
        m, c = kf.initiate(z)
        init_m.append(m); init_c.append(c)
        for _ in range(int(rng.integers(1, 4))):
            m, c = kf.predict(m, c)
        pred_m.append(m); pred_c.append(c)
        pm, pc = kf.project(m, c)
        proj_m.append(pm); proj_c.append(pc)
        z2 = z + np.array([rng.normal(0, 3), rng.normal(0, 3), rng.normal(0, 0.02), rng.normal(0, 2)])
        zs.append(z2)
        um, uc = kf.update(m, c, z2)
        upd_m.append(um); upd_c.append(uc)
        cand = z2[None, :] + rng.normal(0, 1, (6, 4)) * np.array([30, 30, 0.2, 20])
        gate_in.append(cand)
        gate.append(kf.gating_distance(m, c, cand))
    # a long predict/update chain on one track (accumulated rounding)
    m, c = kf.initiate(meas[0])
    chain_z, chain_m, chain_c = [], [], []
    z = meas[0].copy()
    for t in range(50):
        m, c = kf.predict(m, c)
        z = z + np.array([3.0, -1.5, 0.0, 0.2]) + rng.normal(0, 0.5, 4) * np.array([1, 1, 0.01, 1])
        if t % 7 != 3:
            m, c = kf.update(m, c, z)
        chain_z.append(z.copy()); chain_m.append(m.copy()); chain_c.append(c.copy())
    out.update(meas=meas, init_m=init_m, init_c=init_c, pred_m=pred_m, pred_c=pred_c, proj_m=proj_m,
               proj_c=proj_c, zs=zs, upd_m=upd_m, upd_c=upd_c, gate_in=gate_in, gate=gate,
               chain_z=chain_z, chain_m=chain_m, chain_c=chain_c)
    np.savez_compressed(os.path.join(HERE, "kalman.npz"), **{k: np.asarray(v) for k, v in out.items()})


def gen_nms():
    cases = {}
    for ci, n in enumerate([1, 2, 5, 17, 64, 200]):
        xy = rng.uniform(0, 400, (n, 2))
        wh = rng.uniform(20, 150, (n, 2))
        boxes = np.concatenate([xy, wh], 1)
        if n >= 5:      # near-duplicates and exact duplicates
            boxes[1] = boxes[0] + rng.normal(0, 1.0, 4)
            boxes[3] = boxes[2]
        scores = rng.uniform(0.25, 1.0, n)
        if n == 64:
            scores[5] = scores[6]        # tied scores: np.argsort's (unstable) quicksort decides -> case 4 is "order undefined"
        for ov in (0.5, 0.3, 1.0):
            keep = sort.preprocessing.non_max_suppression(boxes.copy(), ov, scores.copy())
            cases[f"c{ci}_boxes"] = boxes
            cases[f"c{ci}_scores"] = scores
            cases[f"c{ci}_ov{ov}"] = np.asarray(keep, dtype=np.int64)
    keep = sort.preprocessing.non_max_suppression(np.zeros((0, 4)), 0.5, np.zeros((0,)))
    cases["empty"] = np.asarray(keep, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "dsort_nms.npz"), **cases)


def gen_iou():
    out = {}
    for ci, (t, d) in enumerate([(1, 1), (3, 5), (16, 9), (40, 64)]):
        a = np.concatenate([rng.uniform(0, 300, (t, 2)), rng.uniform(10, 120, (t, 2))], 1)
        b = np.concatenate([rng.uniform(0, 300, (d, 2)), rng.uniform(10, 120, (d, 2))], 1)
        if t > 1:
            b[0] = a[1]                  # identical boxes -> iou 1
        m = np.stack([sort.iou_matching.iou(a[i], b) for i in range(t)])
        out[f"c{ci}_a"], out[f"c{ci}_b"], out[f"c{ci}_iou"] = a, b, m
    np.savez_compressed(os.path.join(HERE, "iou_cost.npz"), **out)


def gen_cosine():
    out = {}
    metric = sort.nn_matching.NearestNeighborDistanceMetric("cosine", 0.2, budget=4)
    protos = rng.standard_normal((3, 512)).astype(np.float32)
    protos /= np.linalg.norm(protos, axis=1, keepdims=True)
    step_feats, step_targets, step_active, sample_counts = [], [], [], []
    for step in range(7):
        tg = [1, 2, 3] if step < 4 else [1, 3]
        f = np.stack([protos[t - 1] + 0.05 * rng.standard_normal(512).astype(np.float32) for t in tg]).astype(np.float32)
        metric.partial_fit(f, np.asarray(tg), tg)
        step_feats.append(np.pad(f, ((0, 3 - len(tg)), (0, 0))))
        step_targets.append(np.pad(np.asarray(tg), (0, 3 - len(tg)), constant_values=-1))
        sample_counts.append([len(metric.samples.get(k, [])) for k in (1, 2, 3)])
    q = np.stack([protos[i % 3] + 0.1 * rng.standard_normal(512).astype(np.float32) for i in range(5)]).astype(np.float32)
    q[4] *= 3.7                          # un-normalised query: the reference re-normalises
    cost = metric.distance(q, [1, 3])
    out.update(step_feats=np.asarray(step_feats), step_targets=np.asarray(step_targets),
               sample_counts=np.asarray(sample_counts), query=q, cost=cost,
               gallery1=np.asarray(metric.samples[1]), gallery3=np.asarray(metric.samples[3]))
    np.savez_compressed(os.path.join(HERE, "cosine.npz"), **out)


class _T:       # minimal stand-ins: min_cost_matching only indexes the lists it is given
    pass


def gen_assignment():
    la = sort.linear_assignment
    out = {}
    ci = 0
    for (nr, nc) in [(1, 1), (3, 3), (5, 2), (2, 6), (8, 8), (20, 13), (13, 20), (40, 40)]:
        for mode in ("random", "gated", "ties"):
            c = rng.uniform(0, 0.5, (nr, nc))
            if mode == "gated":
                c[rng.random((nr, nc)) < 0.5] = 1e5
            if mode == "ties":
                c = np.round(c, 1)
            max_d = 0.2 if mode != "ties" else 0.3
            cc = c.copy()
            m, ut, ud = la.min_cost_matching(lambda *a, _c=cc: _c.copy(), max_d, [None] * nr, [None] * nc,
                                             list(range(nr)), list(range(nc)))
            out[f"c{ci}_cost"] = c
            out[f"c{ci}_max"] = np.asarray(max_d)
            out[f"c{ci}_matches"] = np.asarray(m, dtype=np.int64).reshape(-1, 2)
            out[f"c{ci}_ut"] = np.asarray(ut, dtype=np.int64)
            out[f"c{ci}_ud"] = np.asarray(ud, dtype=np.int64)
            ci += 1
    out["n_cases"] = np.asarray(ci)
    np.savez_compressed(os.path.join(HERE, "assignment.npz"), **out)


def gen_tracker_traces():
    for name in scenarios.SCENARIOS:
        p, frames = scenarios.build(name)
        metric = sort.nn_matching.NearestNeighborDistanceMetric("cosine", p["max_dist"], p["budget"])
        trk = sort.tracker.Tracker(metric, max_iou_distance=p["max_iou_distance"], max_age=p["max_age"],
                                   n_init=p["n_init"])
        rec = {"n_frames": np.asarray(len(frames))}
        for t, dets in enumerate(frames):
            dl = [sort.detection.Detection(d["tlwh"], d["conf"], d["feature"]) for d in dets]
            trk.predict()
            trk.update(dl)
            rec[f"f{t}_ids"] = np.asarray([x.track_id for x in trk.tracks], dtype=np.int64)
            rec[f"f{t}_state"] = np.asarray([x.state for x in trk.tracks], dtype=np.int64)
            rec[f"f{t}_hits"] = np.asarray([x.hits for x in trk.tracks], dtype=np.int64)
            rec[f"f{t}_age"] = np.asarray([x.age for x in trk.tracks], dtype=np.int64)
            rec[f"f{t}_tsu"] = np.asarray([x.time_since_update for x in trk.tracks], dtype=np.int64)
            rec[f"f{t}_mean"] = np.asarray([x.mean for x in trk.tracks], dtype=np.float64).reshape(-1, 8)
            rec[f"f{t}_covdiag"] = np.asarray([np.diag(x.covariance) for x in trk.tracks], dtype=np.float64).reshape(-1, 8)
            rec[f"f{t}_gallery"] = np.asarray(sorted((k, len(v)) for k, v in metric.samples.items()),
                                              dtype=np.int64).reshape(-1, 2)
        np.savez_compressed(os.path.join(HERE, f"tracker_{name}.npz"), **rec)


def gen_reid():
    import torch
    from vehicle_counting_amd.weights import synth_reid
    model = _refimport.load_reid_model()
    net = model.Net(reid=True, num_classes=751)
    sd = {k: torch.from_numpy(v) for k, v in synth_reid(1702).items()}
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all(m.startswith("classifier") or m.endswith("num_batches_tracked") for m in missing), missing
    net.eval()
    x = rng.standard_normal((6, 3, 50, 50)).astype(np.float32)
    with torch.no_grad():
        y = net(torch.from_numpy(x)).numpy()
    np.savez_compressed(os.path.join(HERE, "reid_forward.npz"), x=x, y=y, seed=np.asarray(1702))


def gen_counting():
    bbp = _refimport.load_bb_polygon()
    cu = _refimport.load_counting_utils()
    zone_path = os.path.join(_refimport.REF, "demo", "sample", "cam_04.json")
    poly, dirs = cu.load_zone_anno(zone_path)
    out = {"zone": poly, "directions": dirs, "boxes": [], "points": [], "vectors": []}
    for _ in range(200):
        x1, y1 = rng.integers(0, 1200), rng.integers(0, 650)
        b = [int(x1), int(y1), int(x1 + rng.integers(5, 200)), int(y1 + rng.integers(5, 200))]
        out["boxes"].append({"box": b, "inside": bool(bbp.check_bbox_intersect_polygon(poly, b))})
    # points on vertices / edges / collinear extensions
    special = [tuple(poly[0]), tuple(poly[1]), (poly[0][0], poly[0][1] - 10), (poly[0][0], poly[0][1] + 10),
               ((poly[0][0] + poly[1][0]) / 2, (poly[0][1] + poly[1][1]) / 2), (0, 0), (700, 400), (1260, 613)]
    sq = [[0, 0], [10, 0], [10, 10], [0, 10]]
    for pt in special:
        out["points"].append({"poly": "zone", "pt": list(pt), "inside": bool(bbp.is_point_in_polygon(poly, pt))})
    for pt in [(5, 5), (0, 5), (10, 5), (5, 0), (5, 10), (0, 0), (10, 10), (11, 5), (5, -1), (10, 11), (0, -3), (10, -3)]:
        out["points"].append({"poly": "square", "pt": list(pt), "inside": bool(bbp.is_point_in_polygon(sq, pt))})
    two_dirs = {"01": dirs["01"], "02": [[900.0, 300.0], [600.0, 600.0]], "03": [[100.0, 100.0], [500.0, 110.0]]}
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(40):
            v = [[float(rng.uniform(0, 1280)), float(rng.uniform(0, 720))], [float(rng.uniform(0, 1280)), float(rng.uniform(0, 720))]]
            out["vectors"].append({"vec": v, "best": cu.find_best_match_direction(v, two_dirs)})
        v0 = [[300.0, 300.0], [300.0, 300.0]]        # zero-length track vector (Q11)
        out["vectors"].append({"vec": v0, "best": cu.find_best_match_direction(v0, two_dirs)})
        vneg = [[943.0, 274.0], [655.0, 575.0]]      # opposite to the only direction -> falls back to the first
        out["vectors"].append({"vec": vneg, "best": cu.find_best_match_direction(vneg, {"01": dirs["01"]})})
    out["two_dirs"] = two_dirs
    # CSV + counts from a synthetic track_dict (the structure VideoCounting.run builds)
    td = [dict() for _ in range(3)]
    recs = []
    for lab in range(3):
        for tid in (3, 1, 7)[: lab + 1]:
            n = int(rng.integers(1, 5))
            f0 = int(rng.integers(1, 50))
            boxes = [np.array([100 + 10 * k + tid, 300 + 5 * k, 160 + 10 * k + tid, 380 + 5 * k]) for k in range(n)]
            frames = [f0 + k for k in range(n)]
            fb, lb = boxes[0], boxes[-1]
            direction = cu.find_best_match_direction((((fb[2] + fb[0]) / 2, (fb[3] + fb[1]) / 2),
                                                      ((lb[2] + lb[0]) / 2, (lb[3] + lb[1]) / 2)), two_dirs)
            td[lab][tid] = {"boxes": boxes, "frames": frames, "color": "x", "direction": direction}
            recs.append({"label": lab, "track": tid, "boxes": [b.tolist() for b in boxes], "frames": frames})
    tmp = os.path.join(HERE, "_tmp.csv")
    cu.save_tracking_to_csv(td, tmp)
    with open(tmp) as f:
        out["csv_text"] = f.read()
    import pandas as pd
    df = pd.read_csv(tmp)
    os.remove(tmp)
    count = {int(d): {c: 0 for c in range(3)} for d in two_dirs}      # counting/utils.py:301-305 (int keys)
    for fid in sorted(set(df.frame_id)):
        count, _ = cu.count_frame_directions(df[df.frame_id == fid], count)
    out["csv_tracks"] = recs
    out["counts"] = {d: [count[int(d)][c] for c in range(3)] for d in two_dirs}
    with open(os.path.join(HERE, "counting.json"), "w") as f:
        json.dump(out, f, indent=1)
    with open(os.path.join(HERE, "cam_04.json"), "w") as f:     # the one real data file the reference holds
        json.dump(json.load(open(zone_path)), f, indent=1)


def gen_overlay_calls():
    """Visualisation egress: the reference's own drawing code (utilities/counting/utils.py draw_anno /
    visualize_one_frame / draw_text / draw_frame_count / count_frame_directions, driven like the body of visualize_merged :307-331)
    executed against a RECORDING cv2: every cv2 call it makes, in order, with its arguments.  OpenCV is not installed, so the pixels
    cannot be pinned; the call list -- what is drawn, where, in which colour / thickness / order, including the one-frame delay of the
    count text -- can.  cv2.getTextSize answers with the metric of the 5 x 7 substitute font (advance 6, height 7, at scale
    max(1, round(2 * fontScale))): the reference lays its header boxes and text lines out from whatever the font reports."""
    import types
    import pandas as pd
    calls = []

    def _pt(p):
        return [int(p[0]), int(p[1])]

    def _col(c):
        return [int(v) for v in c]

    class _Rec(types.ModuleType):
        FONT_HERSHEY_SIMPLEX, FONT_HERSHEY_PLAIN, LINE_AA = 0, 1, 16

        def line(self, img, p0, p1, color, thickness=1, *a, **k):
            calls.append(["line", _pt(p0), _pt(p1), _col(color), int(thickness)]); return img

        def circle(self, img, c, r, color, thickness=1, *a, **k):
            calls.append(["circle", _pt(c), int(r), _col(color), int(thickness)]); return img

        def rectangle(self, img, c1, c2, color, thickness=1, *a, **k):
            calls.append(["rectangle", _pt(c1), _pt(c2), _col(color), int(thickness)]); return img

        def polylines(self, img, pts, closed, color, thickness=1, *a, **k):
            calls.append(["polylines", [[_pt(q.reshape(-1)) for q in p] for p in pts], bool(closed), _col(color), int(thickness)]); return img

        def putText(self, img, text, org, fontFace, fontScale, color, thickness=1, lineType=8, *a, **k):
            calls.append(["putText", str(text), _pt(org), int(fontFace), float(fontScale), _col(color), int(thickness)]); return img

        def getTextSize(self, text, fontFace, fontScale, thickness):
            s = max(1, int(round(2.0 * float(fontScale))))
            return (max(len(text) * 6 - 1, 0) * s, 7 * s), 0

    saved = sys.modules.get("cv2")
    sys.modules["cv2"] = _Rec("cv2")
    for k in [k for k in sys.modules if k.startswith("ref_counting")]:
        del sys.modules[k]
    try:
        cu = _refimport.load_counting_utils()
        zone, dirs = cu.load_zone_anno(os.path.join(_refimport.REF, "demo", "sample", "cam_04.json"))
        dirs = {"01": dirs["01"], "02": [[900.0, 300.0], [600.0, 600.0]]}           # the sample file has one direction; a second one for the counts
        hw = (720, 1280)
        img = np.zeros(hw + (3,), np.uint8)
        rows = []
        for f in (1, 2, 3):
            rows.append(dict(track_id=1, frame_id=f, box=str([100 + 10 * f, 200, 180 + 10 * f, 300]), color=str((10, 200, 30)), label=0, direction=1,
                             fpoint=str((150.0, 250.0)), lpoint=str((170.0, 250.0)), fframe=1, lframe=3))
        for f in (2, 3):
            rows.append(dict(track_id=2, frame_id=f, box=str([400, 100 + 5 * f, 460, 190 + 5 * f]), color=str((250, 20, 20)), label=1, direction=2,
                             fpoint=str((430.0, 155.0)), lpoint=str((430.0, 160.0)), fframe=2, lframe=3))
        rows.append(dict(track_id=11, frame_id=3, box=str([5, 3, 40.7, 60.2]), color=str((1, 2, 3)), label=1, direction=1,
                         fpoint=str((22.9, 31.6)), lpoint=str((22.9, 31.6)), fframe=3, lframe=3))
        df = pd.DataFrame(rows)
        count = {int(d): {label: 0 for label in range(2)} for d in dirs}          # :301-305
        prev_text = None
        frames = []
        for frame_id in (1, 2, 3, 4):                                              # the body of :312-331
            del calls[:]
            tmp = df[df.frame_id.astype(int) == frame_id]
            count, text = cu.count_frame_directions(tmp, count)
            im = cu.draw_anno(img, zone, dirs)
            if len(tmp) > 0:
                im = cu.visualize_one_frame(im, tmp)
            if prev_text:
                im = cu.draw_text(im, prev_text)
            prev_text = text
            im = cu.draw_frame_count(im, frame_id)
            frames.append({"frame_id": frame_id, "calls": json.loads(json.dumps(calls)), "count_text": text})
        out = {"hw": list(hw), "zone": zone, "directions": dirs, "rows": rows, "frames": frames,
               "counts": {str(d): [count[d][c] for c in range(2)] for d in count}}
    finally:
        if saved is not None:
            sys.modules["cv2"] = saved
        else:
            del sys.modules["cv2"]
        for k in [k for k in sys.modules if k.startswith("ref_counting")]:
            del sys.modules[k]
    with open(os.path.join(HERE, "overlay_calls.json"), "w") as f:
        json.dump(out, f, indent=None, separators=(",", ":"))


if __name__ == "__main__":
    gen_kalman(); gen_nms(); gen_iou(); gen_cosine(); gen_assignment(); gen_tracker_traces(); gen_reid(); gen_counting(); gen_overlay_calls()
    print("golden fixtures written to", HERE)
