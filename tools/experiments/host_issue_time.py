"""Host time of the stream path's issue calls (vc_stream_submit, vc_stream_run_async) against the step time, per batch size:
is a small-batch step bound by the host's launches (a hipGraph of the detector chain would help) or by the GPU's dependent kernels?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench

for B in (1, 8, 32):
    st = bench.Stream(bench.WORKLOADS["s640-bf16"], 0, 0, torch.device("cuda:0"), B=B, clip=256)
    st.run_steps(0, 8, False)
    st.sync(1)
    n = 64
    t_sub = t_run = t_col = 0.0
    t0 = time.perf_counter()
    st.submit(100)
    for i in range(100, 100 + n):
        a = time.perf_counter(); st.submit(i + 1); b = time.perf_counter(); st.run_async(i); c = time.perf_counter()
        if i > 100: st.collect(i - 1, False)
        d = time.perf_counter()
        t_sub += b - a; t_run += c - b; t_col += d - c
    st.collect(100 + n - 1, False)
    st.eng.stream_reset()
    dt = time.perf_counter() - t0
    print(f"B={B}: step {dt / n * 1e3:.3f} ms; host: submit {t_sub / n * 1e3:.3f} ms, run_async {t_run / n * 1e3:.3f} ms (includes waiting for the detector), collect {t_col / n * 1e3:.3f} ms (waits for the tracker)", flush=True)
    st.eng.close()
