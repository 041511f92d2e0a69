"""Run one conv shape a few times through vc_conv2d_host (for rocprofv3 PMC passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vehicle_counting_amd.engine as E
B, H, W, Ci, Co, k, s, p = [int(v) for v in os.environ.get("VC_SHAPE", "16,80,80,64,64,3,1,1").split(",")]
rng = np.random.default_rng(0)
x = rng.standard_normal((B, H, W, Ci), dtype=np.float32)
w = (rng.standard_normal((Co, Ci, k, k), dtype=np.float32) / np.sqrt(Ci * k * k)).astype(np.float32)
b = np.zeros(Co, np.float32)
for _ in range(int(os.environ.get("VC_REPS", 3))):
    y = E.conv2d(x, w, b, stride=s, pad=p, act=int(os.environ.get("VC_ACT", 1)), precision="bf16")
print("ok", y.shape, float(np.abs(y).mean()))
