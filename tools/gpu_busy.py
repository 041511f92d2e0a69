"""GPU occupancy over time from a rocprofv3 kernel trace (rocpd db): union-busy fraction and per-queue busy fractions over
the steady-state half of the run."""
import sqlite3, sys
import numpy as np
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(c.execute(f"select start, end, name, {qcol or '0'} from kernels order by start"))
t0, t1 = rows[0][0], rows[-1][1]
# steady state = between the 30th and 70th percentile of the tracker launches (the timed steps of bench.py)
tk = [r[0] for r in rows if "track_batch_kernel" in r[2]]
lo, hi = tk[int(len(tk) * 0.3)], tk[int(len(tk) * 0.7)]
sel = [r for r in rows if r[0] >= lo and r[1] <= hi]
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
span = hi - lo
print(f"window {span/1e6:.2f} ms, {len(sel)} kernels; union busy {100*union([(r[0], r[1]) for r in sel])/span:.1f} %; sum of durations {100*sum(r[1]-r[0] for r in sel)/span:.1f} %")
for q in sorted(set(r[3] for r in sel)):
    s = [r for r in sel if r[3] == q]
    names = {}
    for r in s: names[r[2][:40]] = names.get(r[2][:40], 0) + (r[1] - r[0])
    top = sorted(names.items(), key=lambda kv: -kv[1])[:2]
    print(f"  queue {q}: {len(s)} kernels, busy {100*union([(r[0], r[1]) for r in s])/span:.1f} %  top: {[(n, round(100*v/span,1)) for n, v in top]}")
