"""Diagnosis: dense 256-rectangle scene through the batched stream path vs the oracle tracker (a) on the oracle's own embeddings,
(b) on the ENGINE's embeddings.  (b) equal and (a) not => the difference is a near-tie decided by 3e-5 feature noise, not tracker logic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import vehicle_counting_amd.engine as E
from oracle import deepsort as od, reid as orr, yolov5 as oy
from vehicle_counting_amd.synth import synth_frames, synth_tracks
from vehicle_counting_amd.weights import synth_reid, synth_yolo
import test_gpu_round3 as R

T, B, H, W, n_obj, nc = 24, 32, 640, 640, 256, 3
frames = synth_frames(T, H, W, n_obj=n_obj, seed=1702, bounce=True)
det, cnt = R.injected(synth_tracks(T, H, W, n_obj=n_obj, seed=1702, bounce=True))
eng = E.Engine(synth_yolo("yolov5s", nc=nc, seed=1702, det_scale=4.0, obj_shift=-2.0), synth_reid(1702), precision="f32", num_classes=nc,
               max_batch=32, max_frame_hw=(640, 640), max_crops=32 * 256, max_tracks=8192, nn_budget_cap=60)
tids = [eng.tracker_create(**R.TRACK_KW) for _ in range(nc)]
got = R.stream_rows(eng, tids, torch.from_numpy(frames).cuda(), B, H, W, inject=(det, cnt))
tids2 = [eng.tracker_create(**R.TRACK_KW) for _ in range(nc)]
blocking = []
for f in range(T):
    m = oy.marshal_like_reference(det[f])
    blocking.append(eng.videotracker_run(tids2, frames[f], m["bboxes"], m["classes"], m["scores"]))
oembed = orr.make_embedder(synth_reid(1702))
cur = {}
def eembed(crops):
    return cur["f"][:len(crops)] if False else None
def run_oracle(embed_for_frame):
    ovt = od.VideoTrackerOracle(nc, R.TRACK_CFG, None)
    out = []
    for f in range(T):
        for d in ovt.ds:
            d.embed = embed_for_frame(f)
        m = oy.marshal_like_reference(det[f])
        res = ovt.run(frames[f], m["bboxes"], m["classes"], m["scores"])
        out.append(np.array([list(b) + [tr, lb] for b, tr, lb in zip(res["boxes"], res["tracks"], res["labels"])], dtype=np.int64).reshape(-1, 6))
    return out
ref_a = run_oracle(lambda f: oembed)
def eng_embedder(f):
    def emb(crops):
        # the oracle hands over crops; recompute the boxes it cut them from is awkward -> embed the crops' pixels through vc_embed on a canvas
        feats = []
        for c in crops:
            canvas = np.zeros((H, W, 3), np.uint8); h, w = c.shape[:2]; canvas[:h, :w] = c
            # box (cx, cy, w, h) whose int-truncated, clamped corners are exactly [0, w) x [0, h)
            feats.append(eng.embed(canvas, np.array([[w / 2.0, h / 2.0, float(w), float(h)]]))[0])
        return np.array(feats, np.float32).reshape(-1, 512)
    return emb
ref_b = run_oracle(eng_embedder)
def first_diff(a, b):
    for f in range(T):
        if a[f].shape != b[f].shape or not np.array_equal(a[f][:, 4:], b[f][:, 4:]):
            return f
    return None
print("batched vs blocking product path: first differing frame", first_diff(got, blocking))
print("blocking vs oracle(own features): first differing frame", first_diff(blocking, ref_a))
print("product vs oracle(own features): first differing frame", first_diff(got, ref_a))
print("product vs oracle(engine features): first differing frame", first_diff(got, ref_b))
fa = first_diff(got, ref_a)
if fa is not None:
    g, r = got[fa], ref_a[fa]
    print("rows product", len(g), "oracle", len(r))
    if g.shape == r.shape:
        bad = np.where((g[:, 4:] != r[:, 4:]).any(1))[0]
        print("differing rows:", [(g[i].tolist(), r[i].tolist()) for i in bad])
