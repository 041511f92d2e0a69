#!/bin/bash
# kernel duration of one conv shape per tile config, and its launch floor (VC_CONV_ABLATE=6: the kernel returns at once);
# the per-phase split (prologue / first tile / K loop / epilogue) comes from tools/convdbg.sh
export TMPDIR=/tmp
shape=${1:-16,80,80,64,64,3,1,1}
for cfg in ${CFGS:-3 2 9}; do
  for ab in ${ABL:-0 6}; do
    rm -rf /tmp/ab; VC_SHAPE=$shape VC_CONV_CFG=$cfg VC_CONV_ABLATE=$ab VC_REPS=5 timeout 60 rocprofv3 --kernel-trace -d /tmp/ab -o a -- python tools/conv_one.py > /dev/null 2>&1
    python - <<PY
import sqlite3
c = sqlite3.connect("/tmp/ab/a_results.db")
r = list(c.execute("select min(duration), avg(duration) from kernels where name like '%conv%_kernel%'"))[0]
print("shape $shape cfg $cfg ablate $ab  min %.1f us avg %.1f us" % (r[0]/1e3, r[1]/1e3))
PY
  done
done
