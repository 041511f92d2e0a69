"""GPU: the overlay rasteriser (csrc/overlay.hip, vc_overlay) against the NumPy restatement of its integer rules, pixel for pixel:
random primitive lists (overlapping, clipped at every border, painter's order) and the lists MergedVisualizer builds for a batch."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402

import vehicle_counting_amd.engine as E  # noqa: E402
import vehicle_counting_amd.overlay as ov  # noqa: E402
from overlay_raster import paint  # noqa: E402
from vehicle_counting_amd.weights import synth_reid  # noqa: E402


@pytest.fixture(scope="module")
def eng():
    return E.Engine(None, synth_reid(1702), precision="bf16", max_crops=8, max_frame_hw=(64, 64), max_tracks=16, nn_budget_cap=4)


def _run(eng, frames, prims, first):
    d = torch.from_numpy(frames.copy()).cuda()
    ov.overlay(eng, d.data_ptr(), frames.shape[0], frames.shape[1], frames.shape[2], prims, first)
    want = frames.copy()
    for f in range(frames.shape[0]):
        paint(want[f], prims[first[f]:first[f + 1]])
    return d.cpu().numpy(), want


def test_random_primitives_match_numpy_exactly(eng):
    rng = np.random.default_rng(1702)
    B, H, W = 5, 96, 128
    frames = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    lists = []
    for f in range(B):
        pl = ov.PrimList()
        for _ in range(40):
            k = int(rng.integers(0, 5))
            p0 = rng.integers(-20, [W + 20, H + 20]); p1 = rng.integers(-20, [W + 20, H + 20])
            col = tuple(int(v) for v in rng.integers(0, 256, 3))
            if k == ov.LINE: pl.line(p0, p1, col, int(rng.integers(1, 6)))
            elif k == ov.DISC: pl.disc(p0, int(rng.integers(0, 12)), col)
            elif k == ov.RECT: pl.rect(p0, p1, col, int(rng.integers(1, 5)))
            elif k == ov.FILL: pl.fill(p0, p1, col)
            else: pl.text("Id:7|x", p0, int(rng.integers(1, 4)), col, bold=int(rng.integers(0, 2)))
        lists.append(pl)
    lists[2] = ov.PrimList()                                          # a frame with nothing to draw
    first = np.zeros(B + 1, np.int32)
    for i, pl in enumerate(lists): first[i + 1] = first[i] + len(pl.rows)
    prims = np.array([r for pl in lists for r in pl.rows], dtype=np.int64).astype(np.uint32).view(np.int32).reshape(-1, 12)
    got, want = _run(eng, frames, prims, first)
    assert np.array_equal(got, want)
    assert np.array_equal(got[2], frames[2]) and not np.array_equal(got[0], frames[0])


def test_merged_visualizer_batch(eng):
    zone = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "cam_04_halfres.json")))
    polygon = next(s["points"] for s in zone["shapes"] if s["label"] == "zone")
    directions = {s["label"][-2:]: s["points"] for s in zone["shapes"] if s["label"].startswith("direction")}
    rows = []
    for f in range(1, 7):
        rows.append({"track_id": 3, "frame_id": f, "box": [50 + 8 * f, 60, 120 + 8 * f, 140], "color": (30, 144, 255), "label": 2,
                     "direction": int(sorted(directions)[0]), "fpoint": (85.0, 100.0), "lpoint": (133.0, 100.0), "fframe": 1, "lframe": 6})
    H, W = 360, 640
    frames = np.full((6, H, W, 3), 90, np.uint8)
    viz = ov.MergedVisualizer(rows, directions, polygon, num_classes=4)
    prims, first = viz.batch_prims(list(range(1, 7)), (H, W))
    got, want = _run(eng, frames, prims, first)
    assert np.array_equal(got, want)
    assert (got[0] != 90).any() and viz.count_dict[int(sorted(directions)[0])][2] == 1
