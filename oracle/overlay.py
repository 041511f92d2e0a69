"""TEST INFRASTRUCTURE ONLY (tests/ may import this; the product never does).

CPU restatement of the reference's visualisation egress as a DISPLAY LIST: every cv2 call that
/root/reference/utilities/counting/utils.py makes for a frame, in order, with its arguments -- `draw_arrow` (:7-12), `draw_one_box`
(:17-34), `draw_text` (:36-102), `draw_anno` (:104-121), `draw_frame_count` (:123-126), `visualize_one_frame` (:250-274),
`count_frame_directions` (:276-297) and the per-frame body of `visualize_merged` (:312-331).

Pinned: tests/golden/overlay_calls.json holds the calls the reference's own code made when it was run against a recording cv2
(tests/golden/make_golden.py::gen_overlay_calls); tests/test_overlay.py checks this file against it call for call.  PIXELS are
parity-unpinned: OpenCV is not installed in this image, its line rasteriser and Hershey fonts cannot be run or restated from
/root/reference.  Text metrics: `text_size` below is what the recording cv2's getTextSize answered (the 5 x 7 substitute font)."""
import numpy as np

FONT_HERSHEY_SIMPLEX, FONT_HERSHEY_PLAIN = 0, 1


def text_size(text, font_scale):
    """(w, h) of one line in the substitute font: advance 6, height 7, scale max(1, round(2 * fontScale))."""
    s = max(1, int(round(2.0 * float(font_scale))))
    return max(len(text) * 6 - 1, 0) * s, 7 * s


def _pt(p):
    return [int(p[0]), int(p[1])]


def draw_arrow(calls, start, end, color):
    """utils.py:7-12."""
    calls.append(["line", _pt(start), _pt(end), list(color), 3])
    calls.append(["circle", _pt(end), 8, list(color), -1])


def draw_one_box(calls, img_hw, box, key=None, value=None, color=None, line_thickness=None):
    """utils.py:17-34."""
    tl = line_thickness or int(round(0.001 * max(img_hw)))
    c1, c2 = (int(box[0]), int(box[1])), (int(box[2]), int(box[3]))
    calls.append(["rectangle", list(c1), list(c2), list(color), tl * 2])
    if key is not None and value is not None:
        header = f"{key} || {value}"
        tf = max(tl - 2, 1)
        s_size = text_size(f"| {value}", float(tl) / 3)
        t_size = text_size(f"{key} |", float(tl) / 3)
        c2 = c1[0] + t_size[0] + s_size[0] + 15, c1[1] - t_size[1] - 3
        calls.append(["rectangle", list(c1), list(c2), list(color), -1])
        calls.append(["putText", header, [c1[0], c1[1] - 2], 0, float(tl) / 3, [0, 0, 0], tf])


def draw_text(calls, img_h, text, uv_top_left=None, color=(255, 255, 255), font_scale=0.75, thickness=1, outline_color=(0, 0, 0), line_spacing=1.5):
    """utils.py:36-102."""
    lines = text.splitlines()
    if uv_top_left is None:
        _, h = text_size(lines[0], font_scale)
        uv_top_left = (10, img_h - h * (len(lines) + 3))
    uv = np.array(uv_top_left, dtype=float)
    for line in lines:
        _, h = text_size(line, font_scale)
        org = tuple((uv + [0, h]).astype(int))
        if outline_color is not None:
            calls.append(["putText", line, _pt(org), FONT_HERSHEY_SIMPLEX, float(font_scale), list(outline_color), thickness * 3])
        calls.append(["putText", line, _pt(org), FONT_HERSHEY_SIMPLEX, float(font_scale), list(color), thickness])
        uv += [0, h * line_spacing]


def draw_anno(calls, polygon=None, paths=None):
    """utils.py:104-121."""
    if polygon:
        pts = np.array(polygon, np.int32).reshape(-1, 2)
        calls.append(["polylines", [[_pt(p) for p in pts]], True, [0, 0, 255], 5])
    if paths:
        for path, points in paths.items():
            points = np.array(points, np.int32)
            draw_arrow(calls, points[0], points[1], (0, 0, 0))
            calls.append(["putText", path, _pt(points[1]), FONT_HERSHEY_PLAIN, 1.5, [0, 0, 0], 3])


def draw_frame_count(calls, img_h, frame_id):
    """utils.py:123-126."""
    draw_text(calls, img_h, f"Frame:{frame_id}", (10, 25), color=(0, 255, 0))


def visualize_one_frame(calls, img_hw, rows):
    """utils.py:250-274.  rows: dicts with track_id, box (xyxy), color, label, fpoint (the CSV's columns, already parsed)."""
    for r in rows:
        box = r["box"]
        fpoint = np.array(r["fpoint"]).astype(int)
        cpoint = np.array([(box[2] + box[0]) / 2, (box[3] + box[1]) / 2]).astype(int)
        draw_arrow(calls, fpoint, cpoint, r["color"])
        draw_one_box(calls, img_hw, box, key=f"id: {r['track_id']}", value=f"cls: {r['label']}", color=r["color"])


def count_frame_directions(rows, count_dict):
    """utils.py:276-297."""
    for r in rows:
        if r["lframe"] == r["frame_id"]:
            count_dict[r["direction"]][r["label"]] += 1
    count_text = []
    for d in count_dict.keys():
        tmp = f"direction:{d} || "
        for cls_id in count_dict[d].keys():
            tmp += f"{cls_id}:{count_dict[d][cls_id]} | "
        count_text.append(tmp)
    return count_dict, "\n".join(count_text)


class VisualizeMerged:
    """utils.py:299-331 without the video I/O: feed frame ids in order, get each frame's cv2 calls."""

    def __init__(self, rows, directions, zones, num_classes):
        self.rows, self.directions, self.zones = rows, directions, zones
        self.count = {int(d): {label: 0 for label in range(num_classes)} for d in directions}       # :301-305
        self.prev_text = None                                                                        # :307, "delay direction text by one frame"

    def frame_calls(self, frame_id, img_hw):
        calls = []
        tmp = [r for r in self.rows if int(r["frame_id"]) == frame_id]
        self.count, text = count_frame_directions(tmp, self.count)
        draw_anno(calls, self.zones, self.directions)
        if len(tmp) > 0:
            visualize_one_frame(calls, img_hw, tmp)
        if self.prev_text:
            draw_text(calls, img_hw[0], self.prev_text)
        self.prev_text = text
        draw_frame_count(calls, img_hw[0], frame_id)
        return calls
