"""Idle gaps of one queue in a rocprofv3 kernel trace (rocpd db): for the detector queue of bench.py's steady state, the largest gaps
between consecutive kernels, with the kernels on either side -- launch gaps inside a pass are a few us, a gap between the last kernel
of a pass and the first of the next is the host not having submitted the next batch yet.
usage: python tools/queue_gaps.py <db> [kernel-name substring that identifies the queue, default front_fused]"""
import sqlite3, sys
import numpy as np
c = sqlite3.connect(sys.argv[1])
key = sys.argv[2] if len(sys.argv) > 2 else "front_fused"
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else "stream_id"
rows = list(c.execute(f"select start, end, name, {qcol} from kernels order by start"))
tk = [r[0] for r in rows if "track_batch_kernel" in r[2]]
lo, hi = tk[int(len(tk) * 0.3)], tk[int(len(tk) * 0.7)]
q = next(r[3] for r in rows if key in r[2])
s = [r for r in rows if r[3] == q and r[0] >= lo and r[1] <= hi]
gaps = [(s[i + 1][0] - s[i][1], s[i][2][:50], s[i + 1][2][:50]) for i in range(len(s) - 1)]
g = np.array([x[0] for x in gaps], float) / 1e3
span = (hi - lo) / 1e3
steps = sum(1 for r in s if key in r[2])
print(f"queue {q}: {len(s)} kernels over {span/1e3:.2f} ms ({steps} passes); idle {g.clip(0).sum()/span*100:.1f} % of the window; gaps: median {np.median(g):.1f} us, "
      f"> 20 us: {int((g > 20).sum())} totalling {g[g > 20].sum()/steps:.0f} us per pass, <= 20 us: {g[(g > 0) & (g <= 20)].sum()/steps:.0f} us per pass")
for d, a, b in sorted(gaps, key=lambda x: -x[0])[:12]:
    print(f"  {d/1e3:8.1f} us  after {a}  before {b}")
if len(sys.argv) > 3:                                   # one whole pass, in order: start offset, duration, gap before, kernel
    starts = [i for i, r in enumerate(s) if key in r[2]]
    a, b = starts[len(starts) // 2], starts[len(starts) // 2 + 1]
    t0 = s[a][0]
    print(f"one pass ({b - a} kernels, {(s[b][0] - t0)/1e3:.0f} us from its first kernel to the next pass's first):")
    for i in range(a, b):
        print(f"  +{(s[i][0]-t0)/1e3:8.1f} us  dur {(s[i][1]-s[i][0])/1e3:7.1f}  gap {((s[i][0]-s[i-1][1])/1e3 if i > a else 0):6.1f}  {s[i][2][:90]}")
