"""conv3x3_halo_v2_kernel (tile configuration 55) against the halo configurations the autotuner picked before it (39, 30, 38, 31): bit-identity and
isolated time on the 3x3 / s1 shapes of a 128-frame step (Bottleneck / BasicBlock form: residual + activation).
Usage: [CFGS=39,30,55] python tools/experiments/halo_compare.py [shape ...]"""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

SHAPES = {  # name: (B, H, W, Cin, Cout, act, res_mode)
    "det40_128": (128, 40, 40, 128, 128, 1, 2),
    "det20_256": (128, 20, 20, 256, 256, 1, 2),
    "reid13_128": (1536, 13, 13, 128, 128, 2, 1),
    "reid7_256": (1536, 7, 7, 256, 256, 2, 1),
    "reid4_512": (1536, 4, 4, 512, 512, 2, 1),
}
CFGS = [int(c) for c in os.environ.get("CFGS", "39,38,30,31,55").split(",")]

def child(name, cfg):
    import vehicle_counting_amd.engine as E
    B, H, W, Ci, Co, act, rm = SHAPES[name]
    rng = np.random.default_rng(1)
    x = rng.standard_normal((B, H, W, Ci), dtype=np.float32)
    w = (rng.standard_normal((Co, Ci, 3, 3), dtype=np.float32) / np.sqrt(Ci * 9)).astype(np.float32)
    b = rng.standard_normal(Co, dtype=np.float32) * 0.1
    res = rng.standard_normal((B, H, W, Co), dtype=np.float32)
    os.environ["VC_CONV_CFG"] = str(cfg)
    os.environ["VC_CONV_TIME"] = "20"
    y = E.conv2d(x, w, b, stride=1, pad=1, act=act, res=res, res_mode=rm, precision="bf16")
    np.save(f"/tmp/halo_{name}_{cfg}.npy", y)

if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]))
        sys.exit(0)
    for name in (sys.argv[1:] or SHAPES):
        ref = None
        for cfg in CFGS:
            r = subprocess.run([sys.executable, __file__, "--child", name, str(cfg)], capture_output=True, text=True)
            m = re.search(r"best ([\d.]+) ms mean ([\d.]+) ms, ([\d.]+) TFLOP", r.stderr)
            if r.returncode != 0 or not m:
                print(f"{name} cfg {cfg}: FAILED {r.stderr[-300:]!r}")
                continue
            y = np.load(f"/tmp/halo_{name}_{cfg}.npy")
            if ref is None:
                ref = y
            same = np.array_equal(ref, y)
            print(f"{name} cfg {cfg}: best {m.group(1)} ms mean {m.group(2)} ms {m.group(3)} TFLOP/s  identical_to_first={same} maxdiff={np.abs(ref - y).max():.3g}", flush=True)
