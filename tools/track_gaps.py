"""Tracker kernels of a rocprofv3 kernel trace (rocpd db): per batch the device-resident tracker is one track_batch_kernel launch
preceded by the hoisted appearance-dot kernels (plan / norm / dots) and followed by the free-list merge; no host round trips."""
import sqlite3, sys
import numpy as np
c = sqlite3.connect(sys.argv[1])
for k in ("track_batch_kernel", "track_dots_kernel", "track_norm_kernel", "track_plan_kernel", "merge_free_kernel"):
    rows = list(c.execute(f"select start, end from kernels where name like '%{k}%' order by start"))
    if not rows:
        continue
    a = np.array(rows, dtype=np.int64)
    dur = (a[:, 1] - a[:, 0]) / 1e3
    print(f"{k}: {len(a)} launches, duration avg {dur.mean():.1f} us (p50 {np.median(dur):.1f}, p90 {np.percentile(dur, 90):.1f}, max {dur.max():.1f})")
rows = list(c.execute("select start, end from kernels where name like '%track_%kernel%' or name like '%merge_free%' order by start"))
a = np.array(rows, dtype=np.int64)
b = list(c.execute("select start from kernels where name like '%track_plan_kernel%' order by start"))
e = list(c.execute("select end from kernels where name like '%merge_free_kernel%' order by start"))
n = min(len(b), len(e))
if n:
    per = np.array([e[i][0] - b[i][0] for i in range(n)]) / 1e6
    print(f"tracker time per batch (plan start -> merge end): avg {per.mean():.2f} ms, p50 {np.median(per):.2f}, max {per.max():.2f} over {n} batches")
