#!/bin/bash
# Where a frame of the reference's own loop (batch 1 through the drop-in classes) spends its time: rocprofv3 kernel trace of
# tools/dropin_profile.py -> per-kernel totals (sum of kernel durations per frame against the wall clock per frame).
# usage: bash tools/dropin_trace.sh [out_dir]
export TMPDIR=/tmp
OUT=${1:-gpurun_out/dropin}
mkdir -p $OUT
ROOT=$(pwd)
export VC_TUNE_CACHE=$ROOT/$OUT/tune.txt      # the untraced run picks the tile configurations, the traced run launches no tuning candidates
python tools/dropin_profile.py > /dev/null 2>&1
python tools/dropin_profile.py > $OUT/untraced.json 2> $OUT/untraced.err
cd /tmp && rm -rf /tmp/dt && rocprofv3 --kernel-trace --stats -d /tmp/dt -o d -- python $ROOT/tools/dropin_profile.py > $ROOT/$OUT/traced.json 2> $ROOT/$OUT/traced.err
cd $ROOT
python tools/prof_summary.py $(ls /tmp/dt/*.db | head -1) "tools/dropin_profile.py (24 warm-up + 256 timed + 64 split frames = 344 frames, batch 1)" > $OUT/kernel_stats.md 2>&1
head -70 $OUT/kernel_stats.md; cat $OUT/untraced.json
