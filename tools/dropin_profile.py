"""The drop-in loop alone (bench.py's dropin_bs1 point: CountingPipeline.run, batch 1, host frames) -- run under
`rocprofv3 --kernel-trace --stats` by tools/dropin_trace.sh to see kernel time per frame against wall time per frame."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

hw = tuple(int(v) for v in os.environ.get("VC_FRAME_HW", "640,640").split(","))
wl = bench.WL_720P if hw != (640, 640) else bench.WORKLOADS["s640-bf16"]
print(json.dumps(bench.dropin_point_(wl, 0, hw, int(os.environ.get("VC_FRAMES", 256)), bench.ZONE_720P if hw != (640, 640) else bench.ZONE)))
