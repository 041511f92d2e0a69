#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of one conv shape under given tile configurations (each counter in its own pass).
# usage (on the GPU box): VC_SHAPE=128,160,160,64,128,3,2,1 bash tools/pmc_one.sh 44 5
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_one
mkdir -p $OUT
cd /tmp
for cfg in "$@"; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    VC_CONV_CFG=$cfg VC_REPS=2 timeout -s KILL 300 rocprofv3 --pmc $ctr -d $OUT/c${cfg}_$ctr -o r -- python $GRAFT_REPO_ROOT/tools/conv_one.py > $OUT/log_${cfg}_$ctr.txt 2>&1
    python - <<PY
import sqlite3, glob
for db in glob.glob("$OUT/c${cfg}_$ctr/**/*.db", recursive=True):
    c = sqlite3.connect(db)
    for r in c.execute("select kernel_name, sum(value), count(distinct dispatch_id) from counters_collection where counter_name='$ctr' group by kernel_name"):
        if "conv" in r[0]: print("cfg $cfg $ctr", r[0][:70], "KiB per launch %.0f" % (r[1] / r[2]), "launches", r[2])
PY
  done
done
rm -rf $OUT/c*_FETCH_SIZE $OUT/c*_WRITE_SIZE
