"""From a VC_TUNE_LOG=1 stderr log: per layer key, the best configuration, its time, and where given configurations rank.  usage: tune_rank.py <log> 56 57"""
import sys, collections
want = [int(x) for x in sys.argv[2:]]
t = collections.defaultdict(dict)
for l in open(sys.argv[1], errors="ignore"):
    if not l.startswith("[vc tune]"): continue
    p = l.split()
    t[p[2]][int(p[4])] = float(p[5])
for k, d in t.items():
    best = min(d, key=d.get)
    s = " ".join(f"cfg{c}={d[c]:.4f}({d[c]/d[best]:.2f}x)" for c in want if c in d)
    if s: print(f"{k}: best cfg{best} {d[best]:.4f}  {s}")
