"""front_fused_kernel / c3_fused_kernel timing under ablation masks (engine options "ff_ablate" / "c3_ablate" (diagnostics, wrong results): which component bounds the kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vehicle_counting_amd.engine as E
from vehicle_counting_amd.synth import synth_frames
from vehicle_counting_amd.weights import synth_yolo
B, NC, H, W = 128, 80, 640, 640
eng = E.Engine(synth_yolo("yolov5s", nc=NC, det_scale=4.0, obj_shift=-8.0), None, precision="bf16", num_classes=NC, max_batch=B, max_frame_hw=(H, W))
fr = torch.from_numpy(synth_frames(B, H, W, 12, 1702)).cuda()
names = {0: "full", 1: "no transcendentals", 2: "no stem MFMAs", 3: "no trans + no stem MFMAs", 4: "no stem LDS stores", 8: "no output stores", 16: "no stem phase", 32: "no conv phase",
         48: "patch staging only", 64: "no patch writes", 17: "conv phase, no trans", 33: "stem phase, no trans", 35: "stem phase, no trans, no MFMA", 39: "stem: no trans/MFMA/LDS store"}
for _ in range(2):
    eng.stream_submit(fr.data_ptr(), B, H, W); eng.sync(); eng.stream_reset()
names.update({128: "no global fetch", 192: "no fetch, no patch writes", 256: "no patch LDS reads (stem)", 257: "no patch reads, no trans", 384: "no fetch, no patch reads"})
for abl in (0, 128, 192, 256, 257, 384, 1, 16, 32, 0):
    eng.set_option("ff_ablate", abl)
    eng.profile(True); eng.profile_reset()
    eng.stream_submit(fr.data_ptr(), B, H, W); eng.sync()
    l = [x for x in eng.profile_ops().strip().split("\n") if "cfg=102" in x]
    eng.profile(False)
    eng.stream_reset()
    print(f"abl {abl:3d} {names.get(abl, ''):32s} {l[0].split('ms=')[1].split()[0] if l else '?'} ms", flush=True)

eng.set_option("ff_ablate", 0)
cn = {0: "full", 1: "no transcendentals", 2: "no global fetch", 4: "no output stores", 8: "no 3x3 pass", 16: "no cv12 pass", 32: "no m.cv1 pass", 64: "no cv3 pass",
      6: "no fetch, no stores", 120: "staging only (no passes)", 7: "no trans/fetch/stores"}
for abl in (0, 1, 2, 4, 6, 8, 16, 32, 64, 120, 7, 0):
    eng.set_option("c3_ablate", abl)
    eng.profile(True); eng.profile_reset()
    eng.stream_submit(fr.data_ptr(), B, H, W); eng.sync()
    l = [x for x in eng.profile_ops().strip().split("\n") if "cfg=103" in x]
    eng.profile(False)
    eng.stream_reset()
    print(f"c3 abl {abl:3d} {cn.get(abl, ''):32s} {l[0].split('ms=')[1].split()[0] if l else '?'} ms", flush=True)
