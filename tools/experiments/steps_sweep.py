"""Frames/s of one extra operating point of bench.py against the number of timed steps (fill / drain of the three-stage pipeline, clip wrap-arounds)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
wl = B.WORKLOADS["s640-bf16"]
which = sys.argv[1] if len(sys.argv) > 1 else "k12"
for steps in (12, 24, 48, 96):
    if which == "k12":
        r = B.quick_point(wl, 0, 0, dev, 1, frame_hw=(720, 1280), zone=B.ZONE_720P, n_obj=12, inject=12, clip=256, steps=steps, warmup=3, full=True)
    elif which == "s720p":
        r = B.quick_point(B.WL_720P, 0, 0, dev, 1, frame_hw=(720, 1280), zone=B.ZONE_720P, clip=256, steps=steps, warmup=3, full=True)
    else:
        r = B.quick_point(wl, 0, 0, dev, 1, clip=512, steps=steps, warmup=3, full=True)
    print(which, steps, round(r["value"]), round(r["ms_per_step"], 3), r["stage_ms_per_step"], flush=True)
