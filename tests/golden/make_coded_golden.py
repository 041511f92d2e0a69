"""Generates tests/golden/coded_{s640,s720p,l1280}.json: the fp32 CPU oracle's CSV (oracle/pipeline.py::run_video) on the
well-conditioned clips of tests/coded_case.py.  Run from the repo root:  python tests/golden/make_coded_golden.py [case ...]

These are outputs of THIS repo's oracle (a cache of a deterministic computation: the GPU suite would otherwise spend minutes of host
time re-deriving them on every run; YOLOv5l at 1280 x 1280 takes ~10 s per frame on the CPU), not of the reference -- the detector
half of the oracle is parity-unpinned either way (oracle/yolov5.py).  tests/test_oracle_coded.py re-runs the oracle on the CPU and
checks the committed rows against it."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import coded_case as cc  # noqa: E402
from oracle import pipeline as op  # noqa: E402


def main(names):
    for name in names:
        c = cc.CASES[name]
        ysd, rsd, frames, truth = cc.build(name)
        with tempfile.TemporaryDirectory() as d:
            t0 = time.time()
            rows, counts, n_det = op.run_video(frames, ysd, rsd, cc.TRACK_CFG, cc.zone_file(name, d), variant=c["variant"], nc=cc.NC, size=c["size"])
        out = {"case": dict(c, nc=cc.NC, track=cc.TRACK_CFG), "plates_per_frame": [len(t) for t in truth], "n_det": n_det, "counts": counts,
               "rows": [dict(r, fpoint=list(r["fpoint"]), lpoint=list(r["lpoint"])) for r in rows]}
        with open(cc.golden_path(name), "w") as f:
            json.dump(out, f, separators=(",", ":"))
        print(name, "rows", len(rows), "detections", sum(n_det), "plates", sum(out["plates_per_frame"]), f"{time.time() - t0:.0f} s")


if __name__ == "__main__":
    main(sys.argv[1:] or list(cc.CASES))
