"""The bench's host-frames point (frames in pinned host memory, PCIe inside the timed region): copies staged one batch ahead of the
detector (vc_stream_stage_host) against copies in front of their own detector (VC_BENCH_HOST_INLINE=1), alternating, and the
device-resident rate on the same box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
wl = bench.WORKLOADS["s640-bf16"]
dev = torch.device("cuda:0")
for rep in range(3):
    for inline in (False, True):
        if inline: os.environ["VC_BENCH_HOST_INLINE"] = "1"
        else: os.environ.pop("VC_BENCH_HOST_INLINE", None)
        p = bench.quick_point(wl, 0, 0, dev, 1, host=True, steps=40, warmup=6)
        print("host frames,", "copy before its own detector:" if inline else "copy staged one batch ahead: ", round(p["value"]), "frames/s", flush=True)
os.environ.pop("VC_BENCH_HOST_INLINE", None)
p = bench.quick_point(wl, 0, 0, dev, 1, host=False, steps=40, warmup=6)
print("device frames:", round(p["value"]), "frames/s")
