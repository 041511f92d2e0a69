#!/bin/bash
# One round of committed profiles (run on the GPU box through gpurun, from the repo root):
#   kernel trace + stats, FETCH_SIZE / WRITE_SIZE / MFMA-busy PMC passes (each in its own run), bench JSON with cpu baseline.
# usage: bash tools/profile_round.sh r01
R=${1:-r02}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export VC_TUNE_CACHE=$OUT/tune.txt          # tile configs picked once, so the traced runs contain no autotune launches
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
cp $OUT/tune.txt $OUT/conv_tune.txt
cd /tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras"
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
CMD2="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- $CMD2 > $OUT/fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write -- $CMD2 > $OUT/write.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d $OUT/mfma -o mfma -- $CMD2 > $OUT/mfma.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $OUT/trace/trace_results.db "VC_TUNE_CACHE=tune.txt rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras" > $OUT/kernel_stats.md
python - >> $OUT/kernel_stats.md <<PY
import json, re
line = [l for l in open("$OUT/trace.log", errors="replace") if l.startswith("{") and "avg_launch_us" in l]
if line:
    d = json.loads(line[-1])
    r = d["roofline"]
    print("\nthe traced command's own bench line (HIP start/stop events of the conv launches of its timed steps): average %.2f us per "
          "launch, %.1f TFLOP/s over the wall clock of its timed steps; end to end %.0f frames/s under the tracer" % (r["avg_launch_us"], r["mfma_tflops"], d["value"]))
PY
python tools/pmc_traffic.py $OUT/fetch/fetch_results.db $OUT/write/write_results.db > $OUT/pmc_traffic.json
python tools/pmc_summary.py $OUT/mfma/mfma_results.db vc:: > $OUT/pmc_mfma.txt 2>&1
python tools/gpu_busy.py $OUT/trace/trace_results.db > $OUT/gpu_busy.txt 2>&1
python tools/track_gaps.py $OUT/trace/trace_results.db >> $OUT/gpu_busy.txt 2>&1
unset VC_TUNE_CACHE
cp $OUT/pmc_traffic.json profiles/${R}_pmc_traffic.json      # the bench line quotes the traffic of THESE passes
timeout 600 python bench.py > $OUT/bench_n1.json 2> $OUT/bench.log
rm -rf $OUT/trace $OUT/fetch $OUT/write $OUT/mfma
ls -la $OUT
tail -c 600 $OUT/bench_n1.json
