#!/bin/bash
# kernel trace of the 1280 x 720 operating point (s720p_bf16 of bench.py): which kernels make up its detector tail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/t720
mkdir -p $OUT; cd /tmp; rm -rf /tmp/t720
rocprofv3 --kernel-trace --stats -d /tmp/t720 -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --extras s720p_bf16 > $OUT/log.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(ls /tmp/t720/*.db | head -1) "bench.py --steps 3 --extras s720p_bf16" > $OUT/kernel_stats.md 2>&1
grep -v "conv_igemm\|halo\|fused\|reid\|track_" $OUT/kernel_stats.md | head -40
