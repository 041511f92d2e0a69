#!/bin/bash
# Per-kernel totals of one bench workload (rocprofv3 --kernel-trace --stats): bash tools/trace_workload.sh m1024-bf16 [steps]
export TMPDIR=/tmp
WL=${1:-m1024-bf16}; STEPS=${2:-6}
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$WL
rm -rf /tmp/tw; mkdir -p $OUT
cd /tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d /tmp/tw -o t -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps $STEPS --warmup 2 --no-cpu-baseline --no-extras > $OUT/log.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(ls /tmp/tw/*.db | head -1) "bench.py --workload $WL --steps $STEPS" > $OUT/kernel_stats.md
head -40 $OUT/kernel_stats.md | cut -c1-170
python tools/gpu_busy.py $(ls /tmp/tw/*.db | head -1) > $OUT/gpu_busy.txt 2>&1; cat $OUT/gpu_busy.txt | cut -c1-220
grep -o '"ms_per_step": [0-9.]*\|"stage_ms_per_step": {[^}]*}' $OUT/log.txt | head -3
