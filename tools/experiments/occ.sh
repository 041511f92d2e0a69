#!/bin/bash
# kernel duration of one conv shape vs workgroups per CU (dynamic LDS padding) -- is the K loop bound by bytes in flight?
export TMPDIR=/tmp
shape=${1:-16,80,80,64,64,3,1,1}
for cfg in ${CFGS:-3 6}; do
  for lds in ${LDS:-0 16384 36864 61440 122880}; do
    rm -rf /tmp/ab; VC_SHAPE=$shape VC_CONV_CFG=$cfg VC_CONV_DYN_LDS=$lds VC_REPS=5 timeout 60 rocprofv3 --kernel-trace -d /tmp/ab -o a -- python tools/conv_one.py > /dev/null 2>&1
    python - <<PY
import sqlite3
c = sqlite3.connect("/tmp/ab/a_results.db")
r = list(c.execute("select min(duration), avg(duration) from kernels where name like '%conv%_kernel%'"))[0]
print("shape $shape cfg $cfg dyn_lds $lds  min %.1f us avg %.1f us" % (r[0]/1e3, r[1]/1e3))
PY
  done
done
