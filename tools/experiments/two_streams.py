"""Experiment: two camera streams on one GPU (two engines, B frames each) interleaved from one host thread vs one stream with 2B frames.
Do the detector kernels of one stream fill the other's memory-bound / tail phases?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gc
import torch
import bench

B = int(os.environ.get("VC_B", 64))
steps = int(os.environ.get("VC_STEPS", 40))
wl = dict(bench.WORKLOADS["s640-bf16"])
dev = torch.device("cuda:0")

def timed(streams, n):
    for s in streams: s.sync(1)
    gc.disable()
    t0 = time.perf_counter()
    first = 0
    for s in streams: s.submit(first)
    for i in range(first, first + n):
        for s in streams:
            if i + 1 < first + n: s.submit(i + 1)
        for s in streams: s.run_async(i)
        if i > first:
            for s in streams: s.collect(i - 1, False)
    for s in streams: s.collect(first + n - 1, False)
    for s in streams: s.sync(1)
    dt = time.perf_counter() - t0
    gc.enable()
    return dt

one = bench.Stream(wl, 0, 0, dev, B=2 * B)
timed([one], 5)
dt = timed([one], steps)
print(f"1 stream  x {2*B} frames: {steps * 2 * B / dt:9.0f} frames/s  ({dt / steps * 1e3:.3f} ms/step)")
del one
torch.cuda.empty_cache()
two = [bench.Stream(wl, r, 0, dev, B=B) for r in range(2)]
timed(two, 5)
dt = timed(two, steps)
print(f"2 streams x {B} frames: {steps * 2 * B / dt:9.0f} frames/s  ({dt / steps * 1e3:.3f} ms/step pair)")
