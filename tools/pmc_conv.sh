#!/bin/bash
# SQ counters + per-workgroup phase times of one conv shape under given tile configurations (each counter set its own pass).
# usage (on the GPU box): VC_SHAPE=128,40,40,128,128,3,1,1 bash tools/pmc_conv.sh 39 55
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_conv
mkdir -p $OUT
cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
for cfg in "$@"; do
  echo "== cfg $cfg shape $VC_SHAPE"
  VC_CONV_CFG=$cfg VC_CONV_DBG=1 VC_REPS=2 timeout 120 python $ROOT/tools/conv_one.py 2>&1 | grep "conv dbg" | tail -2
  VC_CONV_CFG=$cfg VC_CONV_TIME=10 VC_REPS=1 timeout 120 python $ROOT/tools/conv_one.py 2>&1 | grep "conv time"
  n=0
  for set in "$P1" "$P2"; do
    n=$((n+1))
    rm -rf $OUT/p$n
    VC_CONV_CFG=$cfg VC_REPS=2 timeout -s KILL 300 rocprofv3 --pmc $set -d $OUT/p$n -o r -- python $ROOT/tools/conv_one.py > $OUT/log_${cfg}_$n.txt 2>&1
    python - <<PY
import sqlite3, glob
for db in glob.glob("$OUT/p$n/**/*.db", recursive=True):
    c = sqlite3.connect(db)
    for r in c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"):
        if "conv" in r[0]: print("  %-28s %14.0f per launch  (%s, %d launches)" % (r[1], r[2] / r[3], r[0][:60], r[3]))
PY
    rm -rf $OUT/p$n
  done
done
