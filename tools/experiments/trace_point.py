"""One of bench.py's extra operating points alone (for rocprofv3 --kernel-trace): python tools/experiments/trace_point.py s720p|m1024|K256"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
which = sys.argv[1] if len(sys.argv) > 1 else "s720p"
wl = bench.WORKLOADS["s640-bf16"]
if which == "s720p":
    out = bench.quick_point_(bench.WL_720P, 0, 0, 0, 1, frame_hw=(720, 1280), zone=bench.ZONE_720P, clip=256, steps=6, warmup=3, full=True)
elif which == "m1024":
    out = bench.quick_point_(bench.WORKLOADS["m1024-bf16"], 0, 0, 0, 1, steps=4, warmup=2, full=True)
elif which == "K256":
    out = bench.quick_point_(wl, 0, 0, 0, 1, n_obj=256, inject=256, B=32, clip=128, steps=6, full=True)
else:
    raise SystemExit("unknown point")
print(json.dumps({k: out[k] for k in ("value", "det_per_frame", "stage_ms_per_step", "ms_per_step") if k in out}))
