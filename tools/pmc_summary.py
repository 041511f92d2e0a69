"""Print per-kernel PMC counter sums from a rocprofv3 --pmc result database."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
filt = sys.argv[2] if len(sys.argv) > 2 else "conv_igemm"
cols = [d[1] for d in c.execute("pragma table_info(counters_collection)")]
rows = list(c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection where kernel_name like ? group by kernel_name, counter_name", (f"%{filt}%",)))
for r in rows:
    print(r[0][:70], r[1], f"{r[2]:.4g}", "n=", r[3])
