"""Tracker kernel durations and the gaps between consecutive tracker launches, from a rocprofv3 kernel trace (rocpd db)."""
import sqlite3, sys
import numpy as np
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select start, end from kernels where name like '%track_step_kernel%' order by start"))
a = np.array(rows, dtype=np.int64)
dur = (a[:, 1] - a[:, 0]) / 1e3
gap = (a[1:, 0] - a[:-1, 1]) / 1e3
print(f"track_step_kernel: {len(a)} launches, duration avg {dur.mean():.1f} us (p50 {np.median(dur):.1f}, p90 {np.percentile(dur, 90):.1f})")
g = gap[gap < 500]
print(f"end->next start gap: p50 {np.median(g):.1f} us, mean {g.mean():.1f} us, p90 {np.percentile(g, 90):.1f} us  (gaps > 500 us = batch boundaries dropped: {len(gap) - len(g)})")
