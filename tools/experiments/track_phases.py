"""Phase split of the tracker kernel (VC_TRACK_DBG=1 prints it per batch) on a bench stream: VC_K detections injected per frame."""
import os, sys
os.environ["VC_TRACK_DBG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
K, B = int(os.environ.get("VC_K", 256)), int(os.environ.get("VC_B", 32))
wl = bench.WORKLOADS[os.environ.get("VC_WL", "s640-bf16")]
st = bench.Stream(wl, 0, 0, torch.device("cuda:0"), n_obj=K or None, inject=K, B=B, clip=4 * B)
st.run_steps(0, 6, False)
