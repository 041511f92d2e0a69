#!/bin/bash
# Per-kernel totals of one extra operating point: bash tools/experiments/trace_point.sh s720p
export TMPDIR=/tmp
P=${1:-s720p}
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$P
rm -rf /tmp/tp; mkdir -p $OUT
export VC_TUNE_CACHE=$OUT/tune.txt
python $GRAFT_REPO_ROOT/tools/experiments/trace_point.py $P > $OUT/untraced.txt 2>&1; tail -1 $OUT/untraced.txt
cd /tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d /tmp/tp -o t -- python $GRAFT_REPO_ROOT/tools/experiments/trace_point.py $P > $OUT/log.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(ls /tmp/tp/*.db | head -1) "trace_point.py $P" > $OUT/kernel_stats.md
head -45 $OUT/kernel_stats.md | sed -e "s/(vc::[^|]*|/ |/" -e "s/(HIP_vector[^|]*|/ |/" -e "s/(unsigned[^|]*|/ |/" -e "s/(int[^|]*|/ |/" | cut -c1-150
python tools/gpu_busy.py $(ls /tmp/tp/*.db | head -1) > $OUT/gpu_busy.txt 2>&1; head -8 $OUT/gpu_busy.txt | cut -c1-220
