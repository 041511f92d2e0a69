"""Visualisation egress, host side (CPU): the primitive list follows utilities/counting/utils.py:299-331 call by call -- what is
drawn per frame, in which order, the running counts and the one-frame delay of the count text."""
import os

import numpy as np

def _mod():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import vehicle_counting_amd.overlay as m
    return m


ZONE = os.path.join(os.path.dirname(__file__), "golden", "cam_04.json")


def _rows():
    # two tracks: track 1 (label 0) over frames 1-3 towards direction 1, track 2 (label 1) over frames 2-3 towards direction 2
    rows = []
    for f in (1, 2, 3):
        rows.append({"track_id": 1, "frame_id": f, "box": [100 + 10 * f, 200, 180 + 10 * f, 300], "color": (10, 200, 30), "label": 0,
                     "direction": 1, "fpoint": (150.0, 250.0), "lpoint": (170.0, 250.0), "fframe": 1, "lframe": 3})
    for f in (2, 3):
        rows.append({"track_id": 2, "frame_id": f, "box": [400, 100 + 5 * f, 460, 190 + 5 * f], "color": (250, 20, 20), "label": 1,
                     "direction": 2, "fpoint": (430.0, 155.0), "lpoint": (430.0, 160.0), "fframe": 2, "lframe": 3})
    return rows


def test_glyphs_are_5x7_and_distinct():
    m = _mod()
    seen = {}
    for ch in "0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ:|.-_,/?":
        b = m.glyph_bits(ch)
        assert 0 < b < (1 << 35)
        assert b not in seen, f"{ch!r} draws like {seen.get(b)!r}"
        seen[b] = ch
    assert m.glyph_bits("a") == m.glyph_bits("A") and m.glyph_bits(" ") == 0
    assert m.glyph_bits("#") == m.glyph_bits("?")                      # unknown characters
    assert m.text_size("id: 7 |", 2) == ((7 * 6 - 1) * 2, 14)


def test_frame_primitives_follow_the_reference_order():
    m = _mod()
    zones = [[100, 100], [900, 100], [900, 600], [100, 600]]
    directions = {"1": [[200, 500], [800, 500]], "2": [[500, 550], [500, 150]]}
    viz = m.MergedVisualizer(_rows(), directions, zones, num_classes=2)
    hw = (720, 1280)
    p1 = viz.frame_prims(1, hw).rows
    kinds = [r[0] for r in p1]
    # draw_anno: 4 polygon edges (red, thickness 5), then per direction: line 3 + disc 8 + its name (outlined glyphs)
    assert kinds[:4] == [m.LINE] * 4 and all(r[5] == 5 and r[6] == 0xFF0000 for r in p1[:4])     # red = (0, 0, 255) BGR -> R << 16
    assert p1[3][1:5] == (100, 600, 100, 100)                                                      # the polygon is closed
    assert kinds[4:6] == [m.LINE, m.DISC] and p1[4][5] == 3 and p1[5][5] == 8 and p1[5][1:3] == (800, 500)
    i = 6
    while kinds[i] == m.GLYPH: i += 1                                  # "1" with its outline: 9 glyph prims
    assert i - 6 == 9
    # frame 1: one row -> arrow (line + disc) in the track's colour, box outline of thickness 2 * round(0.001 * 1280) = 2, header
    j = i
    while not (kinds[j] == m.LINE and p1[j][6] == (10 | 200 << 8 | 30 << 16)): j += 1
    assert kinds[j:j + 4] == [m.LINE, m.DISC, m.RECT, m.FILL]
    assert p1[j][1:5] == (150, 250, 150, 250) and p1[j + 2][1:6] == (110, 200, 190, 300, 2)
    header = [r for r in p1[j + 4:] if r[0] == m.GLYPH and r[6] == 0][:len("id: 1 || cls: 0") - 4]
    assert len(header) == len("id:1||cls:0")                           # spaces paint nothing
    # no count text on the first frame (prev_text is None), "Frame:1" in green last
    tail = [r for r in p1 if r[0] == m.GLYPH and r[6] == (255 << 8)]
    assert len(tail) == len("Frame:1") and p1[-1] == tail[-1]
    # counts: track 1 and track 2 both end in frame 3; the text of frame f is drawn on frame f + 1
    p2 = viz.frame_prims(2, hw).rows
    p3 = viz.frame_prims(3, hw).rows
    assert viz.count_dict == {1: {0: 1, 1: 0}, 2: {0: 0, 1: 1}}
    white = lambda rows: [r for r in rows if r[0] == m.GLYPH and r[6] == 0xFFFFFF]
    assert len(white(p2)) == len(white(p3)) == len("direction:1||0:0|1:0|") * 2
    p4 = viz.frame_prims(4, hw)
    assert len(white(p4.rows)) == len(white(p3))                      # same glyph count, digits differ
    prims, first = viz.batch_prims([5, 6], hw)
    assert prims.dtype == np.int32 and prims.shape[1] == 12 and first[0] == 0 and first[-1] == len(prims)


def test_numpy_rasteriser_basics():
    from overlay_raster import paint
    m = _mod()
    img = np.zeros((40, 60, 3), np.uint8)
    pl = m.PrimList()
    pl.line((5, 5), (25, 15), (1, 2, 3), 1)
    pl.fill((30, 30), (35, 33), (9, 9, 9))
    pl.text("1", (40, 20), 1, (0, 255, 0))
    paint(img, np.array(pl.rows, dtype=np.int64).astype(np.uint32).view(np.int32).reshape(-1, 12))
    assert tuple(img[5, 5]) == (1, 2, 3) and tuple(img[15, 25]) == (1, 2, 3) and tuple(img[10, 15]) == (1, 2, 3)
    assert (img[30:34, 30:36] == 9).all() and img[34, 30].sum() == 0
    assert tuple(img[13, 42]) == (0, 255, 0)                           # top of the '1' glyph: row 0 = "..#.."


# ---- the reference's own call list (tests/golden/overlay_calls.json) -> oracle/overlay.py -> the product's primitive list -------------
def _golden_calls():
    import ast
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "overlay_calls.json")) as f:
        g = json.load(f)
    rows = []
    for r in g["rows"]:                                                # CSV cells as visualize_one_frame evals them (utils.py:261-263)
        r = dict(r)
        for k in ("box", "color", "fpoint", "lpoint"):
            r[k] = ast.literal_eval(r[k])
        rows.append(r)
    return g, rows


def test_oracle_display_list_equals_the_reference_calls():
    """oracle/overlay.py against the cv2 calls the reference's own drawing code made (recording cv2, make_golden.py::gen_overlay_calls):
    4 frames of the body of visualize_merged -- zone polygon, direction arrows + names, per-track arrow / box / header, the count text
    one frame late, the frame counter -- call for call, argument for argument."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import oracle.overlay as O
    g, rows = _golden_calls()
    viz = O.VisualizeMerged(rows, g["directions"], g["zone"], num_classes=2)
    for fr in g["frames"]:
        calls = viz.frame_calls(fr["frame_id"], tuple(g["hw"]))
        assert len(calls) == len(fr["calls"]), fr["frame_id"]
        for a, b in zip(calls, fr["calls"]):
            assert a == b, (fr["frame_id"], a, b)
    assert {str(d): [viz.count[d][c] for c in range(2)] for d in viz.count} == g["counts"]
    assert sum(len(fr["calls"]) for fr in g["frames"]) == 78 and any(c[0] == "polylines" for c in g["frames"][0]["calls"])


def _translate(m, calls):
    """cv2 calls -> the product's primitive rows (the mapping overlay.py's docstring states: line, filled circle -> disc, rectangle
    outline / filled, closed polyline -> its edges, putText -> one glyph primitive per painted character and outline offset)."""
    def bgr(c):
        return (int(c[0]) & 255) | ((int(c[1]) & 255) << 8) | ((int(c[2]) & 255) << 16)
    out = []
    for c in calls:
        if c[0] == "line":
            out.append((m.LINE, c[1][0], c[1][1], c[2][0], c[2][1], c[4], bgr(c[3]), 0, 0, 0, 0, 0))
        elif c[0] == "circle":
            assert c[4] == -1
            out.append((m.DISC, c[1][0], c[1][1], 0, 0, c[2], bgr(c[3]), 0, 0, 0, 0, 0))
        elif c[0] == "rectangle":
            out.append((m.FILL if c[4] < 0 else m.RECT, c[1][0], c[1][1], c[2][0], c[2][1], max(c[4], 0), bgr(c[3]), 0, 0, 0, 0, 0))
        elif c[0] == "polylines":
            for pts in c[1]:
                for a, b in zip(pts, pts[1:] + (pts[:1] if c[2] else [])):
                    out.append((m.LINE, a[0], a[1], b[0], b[1], c[4], bgr(c[3]), 0, 0, 0, 0, 0))
        elif c[0] == "putText":
            text, org, scale, bold = c[1], c[2], max(1, int(round(2.0 * c[4]))), (c[6] - 1) // 2
            for i, ch in enumerate(text):
                bits = m.glyph_bits(ch)
                if bits:
                    for oy in range(-bold, bold + 1):
                        for ox in range(-bold, bold + 1):
                            out.append((m.GLYPH, org[0] + i * 6 * scale + ox, org[1] - 7 * scale + oy, 0, 0, scale, bgr(c[5]), bits & 0xFFFFFFFF, bits >> 32, 0, 0, 0))
        else:
            raise AssertionError(c)
    return out


def test_product_primitives_equal_the_oracle_display_list():
    """vehicle_counting_amd.overlay.MergedVisualizer against oracle/overlay.py, primitive for primitive: the golden scene (whose calls
    are the reference's own) and 20 random scenes -- tracks entering and leaving, fractional boxes, boxes at the frame border (header
    boxes with negative corners), zones of 3 to 7 points, 1 to 4 directions, frames without rows, two frame sizes (tl = 1 and 2)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import oracle.overlay as O
    m = _mod()
    g, rows = _golden_calls()
    scenes = [(rows, g["directions"], g["zone"], 2, tuple(g["hw"]), [1, 2, 3, 4])]
    rng = np.random.default_rng(7)
    for k in range(20):
        hw = (720, 1280) if k % 2 == 0 else (1520, 2704)
        ncls = int(rng.integers(1, 4))
        dirs = {f"{d + 1:02d}": [[float(rng.integers(0, hw[1])), float(rng.integers(0, hw[0]))], [float(rng.integers(0, hw[1])), float(rng.integers(0, hw[0]))]]
                for d in range(int(rng.integers(1, 5)))}
        zone = [[int(rng.integers(0, hw[1])), int(rng.integers(0, hw[0]))] for _ in range(int(rng.integers(3, 8)))]
        rws = []
        for tid in range(int(rng.integers(0, 6))):
            f0, n = int(rng.integers(1, 5)), int(rng.integers(1, 5))
            x, y = float(rng.uniform(-5, hw[1] - 50)), float(rng.uniform(-5, hw[0] - 50))
            boxes = [[x + 7.3 * j, y + 2.1 * j, x + 7.3 * j + 61.5, y + 2.1 * j + 80.25] for j in range(n)]
            fp = ((boxes[0][0] + boxes[0][2]) / 2, (boxes[0][1] + boxes[0][3]) / 2)
            lp = ((boxes[-1][0] + boxes[-1][2]) / 2, (boxes[-1][1] + boxes[-1][3]) / 2)
            for j in range(n):
                rws.append(dict(track_id=tid + 1, frame_id=f0 + j, box=boxes[j], color=tuple(int(v) for v in rng.integers(0, 256, 3)), label=int(rng.integers(0, ncls)),
                                direction=int(rng.integers(1, len(dirs) + 1)), fpoint=fp, lpoint=lp, fframe=f0, lframe=f0 + n - 1))
        scenes.append((rws, dirs, zone, ncls, hw, list(range(1, 10))))
    for rws, dirs, zone, ncls, hw, fids in scenes:
        ora = O.VisualizeMerged(rws, dirs, zone, ncls)
        viz = m.MergedVisualizer(rws, dirs, zone, ncls)
        for fid in fids:
            want = _translate(m, ora.frame_calls(fid, hw))
            got = viz.frame_prims(fid, hw).rows
            assert len(got) == len(want), (fid, len(got), len(want))
            for a, b in zip(got, want):
                assert tuple(a) == tuple(b), (fid, a, b)
        assert viz.count_dict == ora.count
