"""NumPy restatement of the overlay rasteriser's integer rules (csrc/overlay.hip header): test infrastructure only."""
import numpy as np

LINE, DISC, RECT, FILL, GLYPH = 0, 1, 2, 3, 4


def _put(img, x, y, c):
    h, w = img.shape[:2]
    if 0 <= x < w and 0 <= y < h:
        img[y, x] = (c & 255, (c >> 8) & 255, (c >> 16) & 255)


def _line(img, x0, y0, x1, y1, t, c):
    dx, dy = x1 - x0, y1 - y0
    n = max(abs(dx), abs(dy))
    for i in range(n + 1):
        x = x0 + (2 * dx * i + n) // (2 * n) if n else x0            # Python's // is floor division
        y = y0 + (2 * dy * i + n) // (2 * n) if n else y0
        for s in range(t * t):
            _put(img, x - t // 2 + s % t, y - t // 2 + s // t, c)


def paint(img, prims):
    """img: (H, W, 3) uint8 BGR, modified in place; prims: (n, 12) int32 of ONE frame, painted in order."""
    h, w = img.shape[:2]
    for p in np.asarray(prims).astype(np.int64):
        kind, x0, y0, x1, y1, t, c = (int(v) for v in p[:7])
        c &= 0xFFFFFF
        if kind == LINE:
            _line(img, x0, y0, x1, y1, max(t, 1), c)
        elif kind == RECT:
            t = max(t, 1)
            _line(img, x0, y0, x1, y0, t, c); _line(img, x1, y0, x1, y1, t, c); _line(img, x1, y1, x0, y1, t, c); _line(img, x0, y1, x0, y0, t, c)
        elif kind == DISC:
            r = max(t, 0)
            for oy in range(-r, r + 1):
                for ox in range(-r, r + 1):
                    if ox * ox + oy * oy <= r * r:
                        _put(img, x0 + ox, y0 + oy, c)
        elif kind == FILL:
            xa, xb, ya, yb = max(min(x0, x1), 0), min(max(x0, x1), w - 1), max(min(y0, y1), 0), min(max(y0, y1), h - 1)
            if xb >= xa and yb >= ya:
                img[ya:yb + 1, xa:xb + 1] = (c & 255, (c >> 8) & 255, (c >> 16) & 255)
        elif kind == GLYPH:
            s = max(t, 1)
            bits = (int(p[7]) & 0xFFFFFFFF) | ((int(p[8]) & 0xFFFFFFFF) << 32)
            for bit in range(35):
                if (bits >> bit) & 1:
                    for q in range(s * s):
                        _put(img, x0 + (bit % 5) * s + q % s, y0 + (bit // 5) * s + q // s, c)
    return img
