"""Visualisation egress: utilities/counting/utils.py:299-331 `visualize_merged` on frames that stay in HBM.

The host side follows the reference's draw_* helpers call by call and turns each call into primitives (line, disc, rectangle, filled
box, glyph); `vc_overlay` rasterises them on the device, one workgroup per frame, in order.  The reference rasterises with OpenCV
(anti-aliased Hershey fonts); OpenCV is not available here, so the PIXELS are not pinned against it -- the primitive list is the
parity surface: tests/test_overlay.py compares it with the display list of the reference's own drawing code (recorded cv2 calls,
tests/golden/overlay_calls.json, through oracle/overlay.py), and the device rasteriser is pinned against the NumPy rasteriser.  Text uses one
5 x 7 bitmap alphabet (lower case is drawn with the capital glyphs); a text scale s maps OpenCV's fontScale as max(1, round(2 * fontScale))."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L

LINE, DISC, RECT, FILL, GLYPH = 0, 1, 2, 3, 4

_F = {
    " ": ".....|.....|.....|.....|.....|.....|.....", "0": ".###.|#...#|#..##|#.#.#|##..#|#...#|.###.", "1": "..#..|.##..|..#..|..#..|..#..|..#..|.###.",
    "2": ".###.|#...#|....#|...#.|..#..|.#...|#####", "3": "#####|...#.|..#..|...#.|....#|#...#|.###.", "4": "...#.|..##.|.#.#.|#..#.|#####|...#.|...#.",
    "5": "#####|#....|####.|....#|....#|#...#|.###.", "6": "..##.|.#...|#....|####.|#...#|#...#|.###.", "7": "#####|....#|...#.|..#..|.#...|.#...|.#...",
    "8": ".###.|#...#|#...#|.###.|#...#|#...#|.###.", "9": ".###.|#...#|#...#|.####|....#|...#.|.##..", "A": ".###.|#...#|#...#|#####|#...#|#...#|#...#",
    "B": "####.|#...#|#...#|####.|#...#|#...#|####.", "C": ".###.|#...#|#....|#....|#....|#...#|.###.", "D": "###..|#..#.|#...#|#...#|#...#|#..#.|###..",
    "E": "#####|#....|#....|####.|#....|#....|#####", "F": "#####|#....|#....|####.|#....|#....|#....", "G": ".###.|#...#|#....|#.###|#...#|#...#|.####",
    "H": "#...#|#...#|#...#|#####|#...#|#...#|#...#", "I": ".###.|..#..|..#..|..#..|..#..|..#..|.###.", "J": "..###|...#.|...#.|...#.|...#.|#..#.|.##..",
    "K": "#...#|#..#.|#.#..|##...|#.#..|#..#.|#...#", "L": "#....|#....|#....|#....|#....|#....|#####", "M": "#...#|##.##|#.#.#|#.#.#|#...#|#...#|#...#",
    "N": "#...#|#...#|##..#|#.#.#|#..##|#...#|#...#", "O": ".###.|#...#|#...#|#...#|#...#|#...#|.###.", "P": "####.|#...#|#...#|####.|#....|#....|#....",
    "Q": ".###.|#...#|#...#|#...#|#.#.#|#..#.|.##.#", "R": "####.|#...#|#...#|####.|#.#..|#..#.|#...#", "S": ".####|#....|#....|.###.|....#|....#|####.",
    "T": "#####|..#..|..#..|..#..|..#..|..#..|..#..", "U": "#...#|#...#|#...#|#...#|#...#|#...#|.###.", "V": "#...#|#...#|#...#|#...#|#...#|.#.#.|..#..",
    "W": "#...#|#...#|#...#|#.#.#|#.#.#|#.#.#|.#.#.", "X": "#...#|#...#|.#.#.|..#..|.#.#.|#...#|#...#", "Y": "#...#|#...#|#...#|.#.#.|..#..|..#..|..#..",
    "Z": "#####|....#|...#.|..#..|.#...|#....|#####", ":": ".....|..#..|..#..|.....|..#..|..#..|.....", "|": "..#..|..#..|..#..|..#..|..#..|..#..|..#..",
    ".": ".....|.....|.....|.....|.....|.##..|.##..", "-": ".....|.....|.....|#####|.....|.....|.....", "_": ".....|.....|.....|.....|.....|.....|#####",
    ",": ".....|.....|.....|.....|.##..|..#..|.#...", "/": "....#|....#|...#.|..#..|.#...|#....|#....", "?": ".###.|#...#|....#|...#.|..#..|.....|..#..",
}


def glyph_bits(ch):
    """35-bit bitmap of a character, bit 5 * row + column (rows top to bottom); unknown characters draw as '?'."""
    rows = _F.get(ch.upper(), _F["?"]).split("|")
    bits = 0
    for r, row in enumerate(rows):
        for c, px in enumerate(row):
            if px == "#":
                bits |= 1 << (5 * r + c)
    return bits


GLYPH_W, GLYPH_H, GLYPH_ADV = 5, 7, 6                     # cell width / height / advance, in units of the text scale


def _bgr(color):
    b, g, r = (int(v) & 255 for v in color)
    return b | (g << 8) | (r << 16)


def text_scale(font_scale):
    return max(1, int(round(2.0 * float(font_scale))))


def text_size(text, scale):
    """(width, height) in pixels of one line of text, the counterpart of cv2.getTextSize."""
    return (max(len(text) * GLYPH_ADV - 1, 0) * scale, GLYPH_H * scale)


class PrimList:
    """The primitives of ONE frame, in painting order."""

    def __init__(self):
        self.rows = []

    def _add(self, kind, x0, y0, x1, y1, t, color, bits=0):
        self.rows.append((kind, int(x0), int(y0), int(x1), int(y1), int(t), _bgr(color), bits & 0xFFFFFFFF, bits >> 32, 0, 0, 0))

    def line(self, p0, p1, color, thickness):
        self._add(LINE, p0[0], p0[1], p1[0], p1[1], thickness, color)

    def disc(self, c, radius, color):
        self._add(DISC, c[0], c[1], 0, 0, radius, color)

    def rect(self, c1, c2, color, thickness):
        self._add(RECT, c1[0], c1[1], c2[0], c2[1], thickness, color)

    def fill(self, c1, c2, color):
        self._add(FILL, c1[0], c1[1], c2[0], c2[1], 0, color)

    def text(self, s, org_bottom_left, scale, color, bold=0):
        """One line of text whose bottom-left corner is org (cv2.putText's convention).  bold > 0 paints every glyph at the offsets
        -bold .. bold in x and y first (draw_text's outline pass uses thickness * 3)."""
        x, y = int(org_bottom_left[0]), int(org_bottom_left[1]) - GLYPH_H * scale
        for i, ch in enumerate(s):
            bits = glyph_bits(ch)
            if bits == 0:
                continue
            gx = x + i * GLYPH_ADV * scale
            for oy in range(-bold, bold + 1):
                for ox in range(-bold, bold + 1):
                    self._add(GLYPH, gx + ox, y + oy, 0, 0, scale, color, bits)

    # ---- the reference's helpers, call by call ------------------------------------------------------------------------------
    def draw_arrow(self, start, end, color):
        """counting/utils.py:7-12: cv2.line thickness 3 + filled circle radius 8 at the end point."""
        self.line(start, end, color, 3)
        self.disc(end, 8, color)

    def draw_one_box(self, img_hw, box, key, value, color):
        """counting/utils.py:17-34."""
        tl = int(round(0.001 * max(img_hw))) or 1                    # the reference's `line_thickness or ...` (0 -> falls through in cv2 as 1 px)
        c1, c2 = (int(box[0]), int(box[1])), (int(box[2]), int(box[3]))
        self.rect(c1, c2, color, tl * 2)
        if key is not None and value is not None:
            header = f"{key} || {value}"
            scale = text_scale(float(tl) / 3)
            s_size = text_size(f"| {value}", scale)
            t_size = text_size(f"{key} |", scale)
            c2 = (c1[0] + t_size[0] + s_size[0] + 15, c1[1] - t_size[1] - 3)
            self.fill(c1, c2, color)
            self.text(header, (c1[0], c1[1] - 2), scale, (0, 0, 0))

    def draw_text(self, img_h, text, uv_top_left=None, color=(255, 255, 255), font_scale=0.75, outline_color=(0, 0, 0), line_spacing=1.5):
        """counting/utils.py:36-102: multi-line text with an outline; default position = bottom left of the frame."""
        lines = text.splitlines()
        scale = text_scale(font_scale)
        if uv_top_left is None:
            _, h = text_size(lines[0], scale)
            uv_top_left = (10, img_h - h * (len(lines) + 3))
        u, v = float(uv_top_left[0]), float(uv_top_left[1])
        for line in lines:
            _, h = text_size(line, scale)
            org = (int(u), int(v + h))
            if outline_color is not None:
                self.text(line, org, scale, outline_color, bold=1)
            self.text(line, org, scale, color)
            v += h * line_spacing

    def draw_anno(self, polygon=None, paths=None):
        """counting/utils.py:104-121: zone polygon in red (thickness 5), one black arrow + its name per direction."""
        if polygon:
            pts = [(int(p[0]), int(p[1])) for p in polygon]
            for a, b in zip(pts, pts[1:] + pts[:1]):
                self.line(a, b, (0, 0, 255), 5)
        if paths:
            for path, points in paths.items():
                p0, p1 = (int(points[0][0]), int(points[0][1])), (int(points[1][0]), int(points[1][1]))
                self.draw_arrow(p0, p1, (0, 0, 0))
                self.text(str(path), p1, text_scale(1.5), (0, 0, 0), bold=1)

    def draw_frame_count(self, img_h, frame_id):
        """counting/utils.py:123-126."""
        self.draw_text(img_h, f"Frame:{frame_id}", (10, 25), color=(0, 255, 0))

    def visualize_one_frame(self, img_hw, rows):
        """counting/utils.py:250-274.  rows: dicts with track_id, box (xyxy), color, label, fpoint."""
        for r in rows:
            box = r["box"]
            fpoint = (int(r["fpoint"][0]), int(r["fpoint"][1]))
            cpoint = (int((box[2] + box[0]) / 2), int((box[3] + box[1]) / 2))
            self.draw_arrow(fpoint, cpoint, r["color"])
            self.draw_one_box(img_hw, box, f"id: {r['track_id']}", f"cls: {r['label']}", r["color"])


def count_frame_directions(rows, count_dict):
    """counting/utils.py:276-297: count[direction][label] += 1 at each track's last frame; the text of the running totals."""
    for r in rows:
        if r["lframe"] == r["frame_id"]:
            count_dict[r["direction"]][r["label"]] += 1
    lines = []
    for d in count_dict.keys():
        t = f"direction:{d} || "
        for cls_id in count_dict[d].keys():
            t += f"{cls_id}:{count_dict[d][cls_id]} | "
        lines.append(t)
    return count_dict, "\n".join(lines)


class MergedVisualizer:
    """visualize_merged (counting/utils.py:299-331) as a stateful object: feed batches of frames in order; the running counts and
    the one-frame delay of the count text carry over between batches."""

    def __init__(self, csv_rows, directions, zones, num_classes):
        self.by_frame = {}
        for r in csv_rows:
            self.by_frame.setdefault(int(r["frame_id"]), []).append(r)
        self.directions, self.zones = directions, zones
        self.count_dict = {int(d): {label: 0 for label in range(num_classes)} for d in directions}
        self.prev_text = None

    def frame_prims(self, frame_id, hw):
        pl = PrimList()
        rows = self.by_frame.get(int(frame_id), [])
        self.count_dict, text = count_frame_directions(rows, self.count_dict)
        pl.draw_anno(self.zones, self.directions)
        if rows:
            pl.visualize_one_frame(hw, rows)
        if self.prev_text:
            pl.draw_text(hw[0], self.prev_text)
        self.prev_text = text
        pl.draw_frame_count(hw[0], frame_id)
        return pl

    def batch_prims(self, frame_ids, hw):
        lists = [self.frame_prims(f, hw) for f in frame_ids]
        first = np.zeros(len(lists) + 1, np.int32)
        for i, pl in enumerate(lists):
            first[i + 1] = first[i] + len(pl.rows)
        prims = np.array([r for pl in lists for r in pl.rows], dtype=np.int64).astype(np.uint32).view(np.int32).reshape(-1, 12) if first[-1] else np.zeros((0, 12), np.int32)
        return np.ascontiguousarray(prims), first

    def draw(self, engine, frames_dev_ptr, frame_ids, h, w):
        """Paint the overlay of these frames (device pointer to len(frame_ids) x h x w x 3 BGR u8) in place."""
        prims, first = self.batch_prims(frame_ids, (h, w))
        overlay(engine, frames_dev_ptr, len(frame_ids), h, w, prims, first)


def overlay(engine, frames_dev_ptr, b, h, w, prims, first):
    prims = np.ascontiguousarray(prims, dtype=np.int32)
    first = np.ascontiguousarray(first, dtype=np.int32)
    L.check(L.lib().vc_overlay(engine._h, C.c_void_p(frames_dev_ptr), b, h, w, L.ptr(prims.reshape(-1), C.c_int), L.ptr(first, C.c_int)))
