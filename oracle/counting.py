"""CPU ORACLE (test infrastructure, never a product path) -- count / CSV side, rows C1-C6 of SURVEY.md section 8.

Restates utilities/counting/bb_polygon.py and utilities/counting/utils.py (reference paths
relative to /root/reference) with plain float arithmetic; pinned by tests/golden/counting.json
generated from the reference (tests/golden/make_golden.py).
"""
from __future__ import annotations

import json
import math


# ---- bb_polygon.py:14-66 ---------------------------------------------------------------
def _turn(p, q, r):
    """0 collinear, 1 clockwise, 2 counter-clockwise (bb_polygon.py:26-38)."""
    v = (q[1] - p[1]) * (r[0] - q[0]) - (q[0] - p[0]) * (r[1] - q[1])
    return 0 if v == 0 else (1 if v > 0 else 2)


def _within_box(p, q, r):
    """q inside the axis-aligned box of segment pr (bb_polygon.py:14-17)."""
    return min(p[0], r[0]) <= q[0] <= max(p[0], r[0]) and min(p[1], r[1]) <= q[1] <= max(p[1], r[1])


def segments_cross(p1, q1, p2, q2):
    """bb_polygon.py:40-66."""
    o1, o2, o3, o4 = _turn(p1, q1, p2), _turn(p1, q1, q2), _turn(p2, q2, p1), _turn(p2, q2, q1)
    if o1 != o2 and o3 != o4:
        return True
    return ((o1 == 0 and _within_box(p1, p2, q1)) or (o2 == 0 and _within_box(p1, q2, q1)) or
            (o3 == 0 and _within_box(p2, p1, q2)) or (o4 == 0 and _within_box(p2, q1, q2)))


def point_in_polygon(poly, pt):
    """Ray to (x, 1e9); bb_polygon.py:68-93."""
    far = [pt[0], 1e9]
    n = len(poly)
    hits = 0
    for i in range(n):
        a, b = poly[i], poly[(i + 1) % n]
        if segments_cross(a, b, pt, far):
            if _turn(a, pt, b) == 0:
                return _within_box(a, pt, b)
            hits += 1
    return hits % 2 == 1


def bbox_touches_zone(poly, box):
    """Any of the 4 corners inside the polygon; bb_polygon.py:96-114 (Q12)."""
    x1, y1, x2, y2 = box
    return any(point_in_polygon(poly, c) for c in ((x1, y1), (x2, y1), (x2, y2), (x1, y2)))


def cosine_2d(u, v):
    """bb_polygon.py:117-124 (0/0 -> nan like numpy)."""
    ax, ay = float(u[1][0] - u[0][0]), float(u[1][1] - u[0][1])
    bx, by = float(v[1][0] - v[0][0]), float(v[1][1] - v[0][1])
    num = ax * bx + ay * by
    den = math.sqrt(ax * ax + ay * ay) * math.sqrt(bx * bx + by * by)
    if den == 0.0:
        return float("nan") if num == 0.0 else math.copysign(float("inf"), num)
    return num / den


# ---- counting/utils.py ------------------------------------------------------------------
def load_zone(path):
    """utils.py:128-137: zone = shapes[0], directions keyed by last two label chars."""
    with open(path) as f:
        anno = json.load(f)
    dirs = {s["label"][-2:]: s["points"] for s in anno["shapes"] if s["label"].startswith("direction")}
    return anno["shapes"][0]["points"], dirs


def best_direction(vec, dirs):
    """utils.py:139-152: strict '>' from best_score = 0, falls back to the first key (Q11)."""
    keys = list(dirs.keys())
    best, score = keys[0], 0
    for k in keys:
        s = cosine_2d(vec, dirs[k])
        if s > score:
            best, score = k, s
    return best


def build_track_dict(num_classes, polygon, dirs, frames, tracks, labels, boxes):
    """modules/track.py:102-133 minus the random colour (Q10)."""
    td = [dict() for _ in range(num_classes)]
    for f, t, l, b in zip(frames, tracks, labels, boxes):
        if bbox_touches_zone(polygon, b):
            rec = td[l].setdefault(t, {"boxes": [], "frames": []})
            rec["boxes"].append(b)
            rec["frames"].append(f)
    for l in range(num_classes):
        for rec in td[l].values():
            fb, lb = rec["boxes"][0], rec["boxes"][-1]
            fp = ((fb[2] + fb[0]) / 2, (fb[3] + fb[1]) / 2)
            lp = ((lb[2] + lb[0]) / 2, (lb[3] + lb[1]) / 2)
            rec["direction"] = best_direction((fp, lp), dirs)
    return td


def csv_rows(td):
    """utils.py:154-198: one row per (track, frame); label-major, dict insertion order."""
    rows = []
    for l in range(len(td)):
        for t, rec in td[l].items():
            fb, lb = rec["boxes"][0], rec["boxes"][-1]
            fp = ((fb[2] + fb[0]) / 2, (fb[3] + fb[1]) / 2)
            lp = ((lb[2] + lb[0]) / 2, (lb[3] + lb[1]) / 2)
            for b, f in zip(rec["boxes"], rec["frames"]):
                rows.append({"track_id": int(t), "frame_id": int(f), "box": [int(v) for v in b], "label": l,
                             "direction": rec["direction"], "fpoint": fp, "lpoint": lp,
                             "fframe": int(rec["frames"][0]), "lframe": int(rec["frames"][-1])})
    return rows


def direction_counts(rows, dir_keys, num_classes):
    """utils.py:276-287 reduced to its end state: one count per track at its last frame."""
    counts = {d: [0] * num_classes for d in dir_keys}
    for r in rows:
        if r["lframe"] == r["frame_id"]:
            counts[r["direction"]][r["label"]] += 1
    return counts
