"""Tracker time per batch of the bench stream with one kernel on the GPU at a time (blocking events): mean over VC_STEPS steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import vehicle_counting_amd._lib as L

steps = int(os.environ.get("VC_STEPS", 16))
st = bench.Stream(dict(bench.WORKLOADS["s640-bf16"]), 0, 0, torch.device("cuda:0"))
st.run_steps(0, 6, False)
st.eng.sync()
st.eng.profile(True); st.eng.profile_reset()
for i in range(steps):
    st.run_steps(6 + i, 1, False)
st.eng.sync()
tr = st.eng.profile_read(L.PROF_TRACK)
st.eng.profile(False)
print(f"tracker: {tr['ms'] / steps:.3f} ms per {st.B}-frame batch over {steps} batches ({tr['launches']} timed scopes)")
