"""Isolated launch time of the small-M layers of a batch-1 pass under several tile configurations (VC_CONV_TIME through vc_conv2d_host):
what split-K (56 - 59) buys against the tiles the autotuner picks today.  usage (GPU box): python tools/sk_time.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {
    "det 3x3 256->256 20x20 (M 400, K 2304)": (1, 20, 20, 256, 256, 3, 1, 1),
    "det 3x3 128->128 40x40 (M 1600, K 1152)": (1, 40, 40, 128, 128, 3, 1, 1),
    "det 3x3/s2 256->512 40->20 (M 400, K 2304)": (1, 40, 40, 256, 512, 3, 2, 1),
    "det 1x1 512->512 20x20 (M 400, K 512)": (1, 20, 20, 512, 512, 1, 1, 0),
    "det 1x1 1024->512 20x20 (M 400, K 1024)": (1, 20, 20, 1024, 512, 1, 1, 0),
    "reid 3x3 512->512 4x4 x 18 crops (M 288, K 4608)": (18, 4, 4, 512, 512, 3, 1, 1),
    "reid 3x3 256->256 7x7 x 18 crops (M 882, K 2304)": (18, 7, 7, 256, 256, 3, 1, 1),
    "reid 3x3 128->128 13x13 x 18 crops (M 3042, K 1152)": (18, 13, 13, 128, 128, 3, 1, 1),
    "det 3x3 256->256 20x20 x 8 frames (M 3200)": (8, 20, 20, 256, 256, 3, 1, 1),
}
CFGS = [6, 15, 16, 36, 37, 56, 57, 58, 59, 60, 61, 62, 63]
CHILD = """
import sys, numpy as np
sys.path.insert(0, %r)
import vehicle_counting_amd.engine as E
B, H, W, Ci, Co, k, s, p = %r
rng = np.random.default_rng(1)
x = rng.standard_normal((B, H, W, Ci), dtype=np.float32)
w = (rng.standard_normal((Co, Ci, k, k), dtype=np.float32) / np.sqrt(Ci * k * k)).astype(np.float32)
E.conv2d(x, w, np.zeros(Co, np.float32), stride=s, pad=p, act=1, precision="bf16")
"""
for name, case in CASES.items():
    row = []
    for cfg in CFGS:
        env = dict(os.environ, VC_CONV_CFG=str(cfg), VC_CONV_TIME="30", VC_CONV_STRICT="1")
        r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, case)], env=env, capture_output=True, text=True)
        t = [l for l in r.stderr.splitlines() if "[vc conv time]" in l]
        row.append(f"{cfg}: {t[-1].split('best ')[1].split(' ms')[0]}" if t else f"{cfg}: -")
    print(name, "|", "  ".join(row), flush=True)
