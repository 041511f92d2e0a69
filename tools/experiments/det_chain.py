"""Detector chain alone: time per 128-frame batch of submit + run with zero injected detections (no ReID crops, empty tracker steps),
against the sum of the detector kernels' isolated durations -- the difference is what the gaps between dependent launches cost."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
import vehicle_counting_amd._lib as L

st = bench.Stream(dict(bench.WORKLOADS["s640-bf16"]), 0, 0, torch.device("cuda:0"))
B, H, W = st.B, st.H, st.W
zero = (np.zeros((B, 1, 6), np.float32), np.zeros(B, np.int32))
st.eng.stream_inject(*zero)
for i in range(5):
    st.eng.stream_submit(st.batch_ptr(i), B, H, W); st.eng.stream_run_async(st.trackers, st.batch_ptr(i), B, H, W); st.eng.stream_collect()
st.eng.sync()
n = 40
t0 = time.perf_counter()
st.eng.stream_submit(st.batch_ptr(0), B, H, W)
for i in range(n):
    if i + 1 < n: st.eng.stream_submit(st.batch_ptr(i + 1), B, H, W)
    st.eng.stream_run_async(st.trackers, st.batch_ptr(i), B, H, W)
    st.eng.stream_collect()
st.eng.sync()
dt = (time.perf_counter() - t0) / n * 1e3
st.eng.profile(True); st.eng.profile_reset()
st.eng.stream_submit(st.batch_ptr(0), B, H, W); st.eng.stream_run_async(st.trackers, st.batch_ptr(0), B, H, W); st.eng.stream_collect(); st.eng.sync()
c = st.eng.profile_read(L.PROF_CONV); a = st.eng.profile_read(L.PROF_DETECT_AUX)
print(f"detector chain: {dt:.3f} ms per batch back to back; isolated kernels: conv {c['ms']:.3f} ms in {c['launches']} launches + aux {a['ms']:.3f} ms in {a['launches']} launches")
