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
