"""Seeded synthetic video: low-frequency background + textured rectangles moving at constant
velocity with N(0, 0.5 px) jitter.  No real media exists in this environment (demo/sample/cam_04.mp4 is absent)."""
from __future__ import annotations

import numpy as np

SEED = 1702          # the reference's own seed constant (utilities/random_seed.py:5)


def _objects(n_obj, H, W, seed):
    rng = np.random.default_rng(seed)
    pos = np.stack([rng.uniform(0.1 * W, 0.9 * W, n_obj), rng.uniform(0.35 * H, 0.85 * H, n_obj)], 1)
    vel = rng.uniform(-4, 4, (n_obj, 2))
    wh = np.stack([rng.uniform(30, 120, n_obj), rng.uniform(30, 160, n_obj)], 1) * min(1.0, H / 720 + 0.3)
    tex = rng.integers(0, 256, (n_obj, 16, 16, 3), dtype=np.uint8)
    label = rng.integers(0, 3, n_obj)
    return rng, pos, vel, wh, tex, label


def _reflect(x, lo, hi):
    """Fold a coordinate back into [lo, hi] (objects bounce off the frame border instead of leaving a long clip)."""
    span = hi - lo
    y = np.mod(x - lo, 2 * span)
    return lo + np.where(y > span, 2 * span - y, y)


def synth_tracks(n_frames, H, W, n_obj=8, seed=SEED, bounce=False):
    """Ground-truth boxes per frame: (xywh top-left float64 (n,4), labels int64 (n,), scores float64 (n,)).
    bounce=True keeps the objects inside the frame for clips of any length (bench.py's 512-frame streams)."""
    rng, pos, vel, wh, tex, label = _objects(n_obj, H, W, seed)
    jrng = np.random.default_rng(seed + 1)
    out = []
    for t in range(n_frames):
        c = pos + vel * t
        if bounce:
            c = np.stack([_reflect(c[:, 0], wh[:, 0] / 2 + 1, W - wh[:, 0] / 2 - 3), _reflect(c[:, 1], wh[:, 1] / 2 + 1, H - wh[:, 1] / 2 - 3)], 1)
        c = c + jrng.normal(0, 0.5, pos.shape)
        tl = c - wh / 2
        boxes = np.concatenate([tl, wh], 1)
        boxes[:, 0] = np.clip(boxes[:, 0], 0, W - wh[:, 0] - 2)
        boxes[:, 1] = np.clip(boxes[:, 1], 0, H - wh[:, 1] - 2)
        out.append((boxes.astype(np.float64), label.astype(np.int64), jrng.uniform(0.4, 0.95, n_obj)))
    return out


def synth_frames(n_frames, H, W, n_obj=8, seed=SEED, bounce=False):
    """(n_frames, H, W, 3) uint8 BGR frames with the rectangles of synth_tracks() drawn in."""
    rng, pos, vel, wh, tex, label = _objects(n_obj, H, W, seed)
    yy, xx = np.mgrid[0:H, 0:W]
    bg = (96 + 40 * np.sin(xx / 97.0) * np.cos(yy / 61.0))[..., None] + np.array([0, 8, 16])
    frames = np.empty((n_frames, H, W, 3), np.uint8)
    tracks = synth_tracks(n_frames, H, W, n_obj, seed, bounce)
    for t in range(n_frames):
        f = bg.copy()
        for i, b in enumerate(tracks[t][0]):
            x1, y1, w, h = (int(round(v)) for v in b)
            x2, y2 = min(x1 + w, W), min(y1 + h, H)
            if x2 <= x1 or y2 <= y1:
                continue
            ty = (np.arange(y1, y2) - y1) * 16 // max(h, 1)
            tx = (np.arange(x1, x2) - x1) * 16 // max(w, 1)
            f[y1:y2, x1:x2] = tex[i][np.clip(ty, 0, 15)][:, np.clip(tx, 0, 15)]
        frames[t] = np.clip(f, 0, 255).astype(np.uint8)
    return frames
