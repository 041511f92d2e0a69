#!/bin/bash
# One round of committed profiles (run on the GPU box through gpurun, from the repo root):
#   kernel trace + stats, FETCH_SIZE / WRITE_SIZE / MFMA-busy PMC passes (each in its own run), bench JSON with cpu baseline.
# usage: bash tools/profile_round.sh r01
R=${1:-r04}
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
timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench.log
python tools/mfma_util.py $OUT/pmc_mfma.txt $OUT/kernel_stats.md > $OUT/mfma_util.md 2>&1
(echo "# tools/conv_breakdown.py: every conv launch of one step, one kernel on the GPU at a time (blocking events); cfg = tile configuration the autotuner picked (100 stem_direct, 101 reid stem + pool, 102 front_fused, 103 c3_fused, 104 bneck_fused)"; VC_B=256 timeout 300 python tools/conv_breakdown.py 2>/dev/null | grep -v amdgpu.ids) > $OUT/conv_layers_s640.txt
python tools/layer_bounds.py $OUT/conv_layers_s640.txt > $OUT/layer_bounds.md
for f in kernel_stats.md mfma_util.md gpu_busy.txt conv_layers_s640.txt conv_tune.txt layer_bounds.md; do cp $OUT/$f profiles/${R}_$f; done
cp $OUT/bench_n1.json profiles/${R}_bench_n1.json
mkdir -p gpurun_out/profiles_$R && cp profiles/${R}_* gpurun_out/profiles_$R/
rm -rf $OUT/trace $OUT/fetch $OUT/write $OUT/mfma
ls -la $OUT
tail -c 600 $OUT/bench_n1.json
