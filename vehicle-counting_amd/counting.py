"""Zone filter, direction assignment, CSV rows and per-direction counts.

Host-side post-pass, run once per video on the rows the HIP tracker produced; it mirrors the reference's
functions one to one (names, argument meaning, quirks):

  load_zone_anno               utilities/counting/utils.py:128-137
  check_bbox_intersect_polygon utilities/counting/bb_polygon.py:96-114 (any bbox corner inside shapes[0], Q12)
  find_best_match_direction    utilities/counting/utils.py:139-152     (strict '>' from 0, first key fallback, Q11)
  save_tracking_to_csv         utilities/counting/utils.py:154-198
  count_frame_directions       utilities/counting/utils.py:276-297     (end state: one count per track)
"""
from __future__ import annotations

import csv

import numpy as np
import json
import math


def load_zone_anno(zone_path):
    with open(zone_path, "r") as f:
        anno = json.load(f)
    directions = {}
    for shape in anno["shapes"]:
        if shape["label"].startswith("direction"):
            directions[shape["label"][-2:]] = shape["points"]
    return anno["shapes"][0]["points"], directions


def _orient(p, q, r):
    v = (q[1] - p[1]) * (r[0] - q[0]) - (q[0] - p[0]) * (r[1] - q[1])
    if v == 0:
        return 0
    return 1 if v > 0 else 2


def _on_segment(p, q, r):
    return (q[0] <= max(p[0], r[0]) and q[0] >= min(p[0], r[0]) and q[1] <= max(p[1], r[1]) and q[1] >= min(p[1], r[1]))


def _intersect(p1, q1, p2, q2):
    o1, o2 = _orient(p1, q1, p2), _orient(p1, q1, q2)
    o3, o4 = _orient(p2, q2, p1), _orient(p2, q2, q1)
    if o1 != o2 and o3 != o4:
        return True
    if o1 == 0 and _on_segment(p1, p2, q1):
        return True
    if o2 == 0 and _on_segment(p1, q2, q1):
        return True
    if o3 == 0 and _on_segment(p2, p1, q2):
        return True
    return o4 == 0 and _on_segment(p2, q1, q2)


def is_point_in_polygon(polygon, point):
    extreme = (point[0], 1e9)
    count = 0
    n = len(polygon)
    for i in range(n):
        a, b = polygon[i], polygon[(i + 1) % n]
        if _intersect(a, b, point, extreme):
            if _orient(a, point, b) == 0:
                return _on_segment(a, point, b)
            count += 1
    return count % 2 == 1


def check_bbox_intersect_polygon(polygon, bbox):
    x1, y1, x2, y2 = bbox
    for corner in ((x1, y1), (x2, y1), (x2, y2), (x1, y2)):
        if is_point_in_polygon(polygon, corner):
            return True
    return False


def zone_mask(polygon, boxes):
    """check_bbox_intersect_polygon for many integer boxes at once through the library's host routine (same arithmetic)."""
    import ctypes as C

    import numpy as np

    from . import _lib as L
    b = np.ascontiguousarray(np.asarray(boxes, dtype=np.int64).reshape(-1, 4))
    poly = np.ascontiguousarray(np.asarray(polygon, dtype=np.float64).reshape(-1, 2))
    out = np.zeros(len(b), np.uint8)
    L.check(L.lib().vc_zone_filter_host(L.ptr(poly, C.c_double), len(poly), L.ptr(b, C.c_int64), len(b), L.ptr(out, C.c_uint8)))
    return out.astype(bool)


def cosin_similarity(a2d, b2d):
    ax, ay = float(a2d[1][0] - a2d[0][0]), float(a2d[1][1] - a2d[0][1])
    bx, by = float(b2d[1][0] - b2d[0][0]), float(b2d[1][1] - b2d[0][1])
    den = math.sqrt(ax * ax + ay * ay) * math.sqrt(bx * bx + by * by)      # np.linalg.norm == sqrt(dot(x, x))
    num = ax * bx + ay * by
    if den == 0.0:
        return float("nan") if num == 0.0 else math.copysign(float("inf"), num)
    return num / den


def find_best_match_direction(obj_vector, paths):
    keys = list(paths.keys())
    best_score, best_match = 0, keys[0]
    for k in keys:
        score = cosin_similarity(obj_vector, paths[k])
        if score > best_score:
            best_score, best_match = score, k
    return best_match


def _centre(box):
    return ((box[2] + box[0]) / 2, (box[3] + box[1]) / 2)


def csv_records(track_dict):
    """Rows of save_tracking_to_csv (colour omitted: the reference draws it from an unseeded RNG, Q10)."""
    rows = []
    for label_id in range(len(track_dict)):
        for track_id, rec in track_dict[label_id].items():
            boxes = np.asarray(rec["boxes"]).reshape(-1, 4).tolist()         # one conversion per track, not per value
            frames = np.asarray(rec["frames"]).reshape(-1).tolist()
            fpoint, lpoint = _centre(boxes[0]), _centre(boxes[-1])
            tid, colour, direction = int(track_id), rec.get("color", ""), rec["direction"]
            fp, lp = (float(fpoint[0]), float(fpoint[1])), (float(lpoint[0]), float(lpoint[1]))
            ff, lf = int(frames[0]), int(frames[-1])
            for box, frame in zip(boxes, frames):
                rows.append({"track_id": tid, "frame_id": int(frame), "box": [int(v) for v in box], "color": colour, "label": label_id,
                             "direction": direction, "fpoint": fp, "lpoint": lp, "fframe": ff, "lframe": lf})
    return rows


COLUMNS = ["track_id", "frame_id", "box", "color", "label", "direction", "fpoint", "lpoint", "fframe", "lframe"]


def save_tracking_to_csv(track_dict, filename):
    rows = csv_records(track_dict)
    with open(filename, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(COLUMNS)
        for r in rows:
            w.writerow([r["track_id"], r["frame_id"], str(r["box"]), r["color"], r["label"], r["direction"],
                        str(r["fpoint"]), str(r["lpoint"]), r["fframe"], r["lframe"]])
    return rows


def count_directions(rows, direction_keys, num_classes):
    """End state of count_frame_directions over all frames: count[direction][label] += 1 at each track's last frame."""
    counts = {d: [0] * num_classes for d in direction_keys}
    for r in rows:
        if r["lframe"] == r["frame_id"]:
            counts[r["direction"]][r["label"]] += 1
    return counts


class NativeCounter:
    """VideoCounting's end state behind the C ABI (vc_counter_* / vc_counts): zone filter, per-track first / last box, direction
    assignment, the per-(direction, class) counts (the tensor the all-gather merges) and the CSV table, fed batch by batch.  Same
    arithmetic and row order as VideoCounting + csv_records above, which the tests hold it equal to."""

    def __init__(self, zone_path, num_classes):
        import ctypes as C

        from . import _lib as L
        polygon, directions = load_zone_anno(zone_path)
        self.direction_keys = list(directions.keys())
        self.num_classes = num_classes
        poly = np.ascontiguousarray(np.asarray(polygon, np.float64).reshape(-1, 2))
        lines = np.ascontiguousarray(np.asarray([directions[k] for k in self.direction_keys], np.float64).reshape(-1, 4))
        self._h = C.c_void_p()
        L.check(L.lib().vc_counter_create(L.ptr(poly, C.c_double), len(poly), L.ptr(lines, C.c_double), len(lines), num_classes, C.byref(self._h)))

    def add(self, frames, tracks, labels, boxes):
        import ctypes as C

        from . import _lib as L
        f = np.ascontiguousarray(frames, dtype=np.int64).reshape(-1)
        t = np.ascontiguousarray(tracks, dtype=np.int64).reshape(-1)
        lab = np.ascontiguousarray(labels, dtype=np.int64).reshape(-1)
        b = np.ascontiguousarray(boxes, dtype=np.int64).reshape(-1, 4)
        L.check(L.lib().vc_counter_add(self._h, L.ptr(f, C.c_int64), L.ptr(t, C.c_int64), L.ptr(lab, C.c_int64), L.ptr(b, C.c_int64), len(f)))

    def counts(self):
        """int32 (n_dir, n_cls), rows in the order of the zone file's direction shapes."""
        import ctypes as C

        from . import _lib as L
        out = np.zeros((len(self.direction_keys), self.num_classes), np.int32)
        L.check(L.lib().vc_counts(self._h, L.ptr(out, C.c_int)))
        return out

    def table(self):
        """save_tracking_to_csv's table as columns (numpy arrays, CSV row order); `direction` holds the reference's keys ('01', ...)."""
        import ctypes as C

        from . import _lib as L
        n = C.c_int64()
        L.check(L.lib().vc_counter_rows_count(self._h, C.byref(n)))
        n = n.value
        t = {"track_id": np.zeros(n, np.int64), "frame_id": np.zeros(n, np.int64), "box": np.zeros((n, 4), np.int64), "label": np.zeros(n, np.int64),
             "fpoint": np.zeros((n, 2), np.float64), "lpoint": np.zeros((n, 2), np.float64), "fframe": np.zeros(n, np.int64), "lframe": np.zeros(n, np.int64)}
        d = np.zeros(n, np.int32)
        L.check(L.lib().vc_counter_rows(self._h, n, L.ptr(t["track_id"], C.c_int64), L.ptr(t["frame_id"], C.c_int64), L.ptr(t["box"], C.c_int64),
                                        L.ptr(t["label"], C.c_int64), L.ptr(d, C.c_int), L.ptr(t["fpoint"], C.c_double), L.ptr(t["lpoint"], C.c_double),
                                        L.ptr(t["fframe"], C.c_int64), L.ptr(t["lframe"], C.c_int64)))
        t["direction_index"] = d
        t["direction"] = np.asarray(self.direction_keys, dtype=object)[d] if n else np.zeros(0, dtype=object)
        return t

    def records(self):
        """The same rows as csv_records(VideoCounting(...).run(...)) (list of dicts; colour omitted)."""
        t = self.table()
        box, fp, lp = t["box"].tolist(), t["fpoint"].tolist(), t["lpoint"].tolist()
        return [{"track_id": int(t["track_id"][i]), "frame_id": int(t["frame_id"][i]), "box": box[i], "color": "", "label": int(t["label"][i]),
                 "direction": t["direction"][i], "fpoint": (fp[i][0], fp[i][1]), "lpoint": (lp[i][0], lp[i][1]),
                 "fframe": int(t["fframe"][i]), "lframe": int(t["lframe"][i])} for i in range(len(box))]

    def save_csv(self, filename):
        rows = self.records()
        with open(filename, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(COLUMNS)
            for r in rows:
                w.writerow([r["track_id"], r["frame_id"], str(r["box"]), r["color"], r["label"], r["direction"],
                            str(r["fpoint"]), str(r["lpoint"]), r["fframe"], r["lframe"]])
        return rows

    def close(self):
        from . import _lib as L
        if self._h:
            L.lib().vc_counter_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
