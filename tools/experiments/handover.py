"""Hand-over from the detector to the ReID pass in a rocprofv3 kernel trace: nms_scan_kernel end -> the batch's crop_resize start (D2H of the
detections, host marshal, H2D of the crop list), and crop start -> track_batch_kernel start.  usage: python tools/experiments/handover.py <db>"""
import sqlite3, sys
import numpy as np
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select start, end, name from kernels order by start"))
nms = [r for r in rows if "nms_scan_kernel" in r[2]]
crop = [r for r in rows if "crop_resize" in r[2]]
trk = [r for r in rows if "track_batch_kernel" in r[2]]
front = [r for r in rows if "front_fused" in r[2]]
out = []
for n in nms[3:-2]:
    cr = next((x for x in crop if x[0] > n[1]), None)
    tb = next((x for x in trk if cr and x[0] > cr[0]), None)
    fr = next((x for x in front if x[0] > n[1]), None)
    if cr and tb and fr:
        out.append(((cr[0] - n[1]) / 1e3, (tb[0] - cr[0]) / 1e3, (tb[1] - tb[0]) / 1e3, (fr[0] - n[1]) / 1e3))
a = np.array(out)
print(f"{len(a)} passes: nms end -> crop start {np.median(a[:,0]):.0f} us (p10 {np.percentile(a[:,0],10):.0f}, p90 {np.percentile(a[:,0],90):.0f}); crop start -> tracker start "
      f"{np.median(a[:,1]):.0f} us; tracker kernel {np.median(a[:,2]):.0f} us; nms end -> next front kernel {np.median(a[:,3]):.0f} us")
try:
    mc = list(c.execute("select start, end, bytes from memory_copies order by start"))
    big = [m for m in mc if m[2] > 500000]
    if big: print(f"copies > 0.5 MB: {len(big)}, median {np.median([(m[1]-m[0])/1e3 for m in big]):.0f} us for {np.median([m[2] for m in big])/1e3:.0f} KB")
except Exception as ex:
    print("no memory copy table:", ex)
