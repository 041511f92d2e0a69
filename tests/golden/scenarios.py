"""Seeded synthetic tracker scenarios shared by the golden generator and the parity tests.

The scenario *inputs* (per-frame detections: tlwh, confidence, 512-d feature) are regenerated from the
seed wherever they are needed, so only the reference's *outputs* have to be committed as fixtures.
"""
from __future__ import annotations

import numpy as np

FEAT_DIM = 512

# name -> (seed, frames, objects, tracker params, event script)
PARAMS_CAM04 = dict(max_dist=0.2, budget=60, max_iou_distance=0.6, max_age=30, n_init=3)   # configs/cam_configs.yaml:14-22
PARAMS_SHORT = dict(max_dist=0.2, budget=5, max_iou_distance=0.6, max_age=4, n_init=3)


def _identity_features(rng, n):
    f = rng.standard_normal((n, FEAT_DIM)).astype(np.float32)
    return f / np.linalg.norm(f, axis=1, keepdims=True)


def _observe(rng, proto, noise):
    f = proto + noise * rng.standard_normal(FEAT_DIM).astype(np.float32)
    return (f / np.linalg.norm(f)).astype(np.float32)


def build(name):
    """Return (params, frames) with frames = list of list of dict(tlwh f64[4], conf float, feature f32[512])."""
    spec = SCENARIOS[name]
    rng = np.random.default_rng(spec["seed"])
    n, T = spec["objects"], spec["frames"]
    protos = _identity_features(rng, n)
    lo, hi = spec.get("area", ([50, 50], [1100, 600]))
    pos = rng.uniform(lo, hi, size=(n, 2))
    vel = rng.uniform(-6, 6, size=(n, 2))
    wh = rng.uniform([30, 30], [120, 160], size=(n, 2))
    if spec.get("crossing"):
        pos[0], vel[0], wh[0] = (200.0, 300.0), (8.0, 0.0), (60.0, 80.0)
        pos[1], vel[1], wh[1] = (600.0, 302.0), (-8.0, 0.0), (60.0, 80.0)
    frames = []
    for t in range(T):
        dets = []
        for i in range(n):
            if not spec["visible"](i, t, rng):
                continue
            c = pos[i] + vel[i] * t + rng.normal(0, 0.5, 2)
            s = wh[i] * (1 + rng.normal(0, 0.01, 2))
            dets.append({"tlwh": np.array([c[0] - s[0] / 2, c[1] - s[1] / 2, s[0], s[1]], dtype=np.float64),
                         "conf": float(rng.uniform(0.3, 0.95)),
                         "feature": _observe(rng, protos[i], spec.get("feat_noise", 0.01))})
            if spec.get("glitch") and spec["glitch"](i, t):       # the object is there but looks like nothing seen before
                dets[-1]["feature"] = _identity_features(rng, 1)[0]
        for extra in spec.get("extras", lambda t, rng, protos: [])(t, rng, protos):
            dets.append(extra)
        if spec.get("suppress"):                                   # a detector-side NMS: heavily covered boxes vanish for a frame
            dets = _suppress(dets, spec["suppress"])
        order = rng.permutation(len(dets))
        frames.append([dets[j] for j in order])
    return spec["params"], frames


def _suppress(dets, max_overlap):
    """Greedy, score-descending: a box disappears when a kept box covers more than `max_overlap` of it (what DeepSORT's own
    NMS does to overlapping detections, sort/preprocessing.py -- here it only shapes the scenario's input)."""
    order = sorted(range(len(dets)), key=lambda j: -dets[j]["conf"])
    kept = []
    for j in order:
        x, y, w, h = dets[j]["tlwh"]
        ok = True
        for k in kept:
            a, b, c, d = dets[k]["tlwh"]
            iw, ih = min(x + w, a + c) - max(x, a), min(y + h, b + d) - max(y, b)
            if iw > 0 and ih > 0 and iw * ih / (w * h) > max_overlap:
                ok = False
                break
        if ok:
            kept.append(j)
    return [dets[j] for j in sorted(kept)]


def _always(i, t, rng):
    return True


def _occlusion(i, t, rng):
    return not (i == 0 and 10 <= t < 18) and not (i == 2 and 20 <= t < 23)


def _deletion(i, t, rng):
    return not (i == 1 and t >= 8) and not (i == 0 and 12 <= t < 15)


def _flicker(i, t, rng):
    if i == 3:
        return t in (5, 6, 20)            # tentative tracks that die
    return True


def _random_vis(i, t, rng):
    born = (i * 3) % 17
    return t >= born and rng.random() > 0.15


def _crowded_vis(i, t, rng):
    return not (t >= 6 and i in {(11 * t + 5 * k + 1) % 40 for k in range(3)})        # three objects hidden per frame ...


def _crowded_glitch(i, t):
    return t >= 6 and i in {(7 * t + 3 * k) % 40 for k in range(3)}                     # ... three others change their looks ...


def _crowded_new(t, rng, protos):
    if t < 6:
        return []                                                                       # ... and three strangers show up somewhere
    return [{"tlwh": np.concatenate([rng.uniform([20, 20], [1150, 640]), rng.uniform([30, 30], [90, 110])]), "conf": float(rng.uniform(0.3, 0.95)),
             "feature": _identity_features(rng, 1)[0]} for _ in range(3)]


def _far_twin(t, rng, protos):
    """A detection carrying object 0's appearance but far away -> must be gated by Mahalanobis."""
    if t < 6:
        return []
    return [{"tlwh": np.array([1150.0, 650.0, 50.0, 60.0]), "conf": 0.9,
             "feature": _observe(rng, protos[0], 0.01)}]


SCENARIOS = {
    "steady": dict(seed=11, frames=25, objects=5, params=PARAMS_CAM04, visible=_always),
    "occlusion": dict(seed=12, frames=40, objects=4, params=PARAMS_CAM04, visible=_occlusion),
    "deletion": dict(seed=13, frames=30, objects=3, params=PARAMS_SHORT, visible=_deletion),
    "crossing": dict(seed=14, frames=60, objects=2, params=PARAMS_CAM04, visible=_always, crossing=True),
    "flicker": dict(seed=15, frames=30, objects=4, params=PARAMS_CAM04, visible=_flicker),
    "gated_twin": dict(seed=16, frames=20, objects=3, params=PARAMS_CAM04, visible=_always, extras=_far_twin),
    "stress": dict(seed=17, frames=80, objects=14, params=PARAMS_SHORT, visible=_random_vis, feat_noise=0.03),
    "budget": dict(seed=18, frames=30, objects=3, params=PARAMS_SHORT, visible=_always, feat_noise=0.02),
    # 45 objects packed into a 420 x 300 px area, 15 % of them missing in any frame: dozens of overlapping IoU candidates, many
    # rejected pairs per step, confirmed tracks with large list positions missing single frames -- the step where the reference's
    # `list(set(track_indices) - set(matched))` (linear_assignment.py:144) is NOT ascending and decides the ids of new tracks
    "crowded": dict(seed=28, frames=28, objects=40, params=PARAMS_CAM04, visible=_crowded_vis, glitch=_crowded_glitch, extras=_crowded_new,
                    area=([60, 60], [1100, 620])),
    # the same events among 90 objects: more than 64 live tracks, i.e. the general (LDS-list) matching path of the device kernel
    "crowded90": dict(seed=28, frames=20, objects=90, params=PARAMS_CAM04, visible=_crowded_vis, glitch=_crowded_glitch, extras=_crowded_new,
                      area=([40, 40], [1180, 660])),
}
