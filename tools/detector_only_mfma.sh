#!/bin/bash
# MFMA busy of the DETECTOR's conv kernels alone (no ReID, no co-running queue): the figure north_star's ">= 40 % MFMA utilisation on the
# backbone convs" refers to.  tools/conv_breakdown.py with a head that finds nothing (no crops -> the ReID net never runs), 256 frames of
# 640 x 640; kernel trace and PMC pass are separate runs (never combined).  usage (on the GPU box): bash tools/detector_only_mfma.sh <out dir>
OUT=${1:-$GRAFT_REPO_ROOT/gpurun_out/detonly}
export TMPDIR=/tmp
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
export VC_TUNE_CACHE=$OUT/tune.txt VC_B=256 VC_INJECT=0 VC_OBJ_SHIFT=-8
cd /tmp
timeout 300 python $R/tools/conv_breakdown.py > /dev/null 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/tools/conv_breakdown.py > $OUT/trace.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d $OUT/mfma -o mfma -- python $R/tools/conv_breakdown.py > $OUT/mfma.log 2>&1
cd $R
python tools/prof_summary.py $(find $OUT/trace -name "*.db" | head -1) "VC_B=256 VC_INJECT=0 VC_OBJ_SHIFT=-8 rocprofv3 --kernel-trace --stats -- python tools/conv_breakdown.py (detector only: the head finds nothing, the ReID net never runs)" > $OUT/kernel_stats.md
python tools/pmc_summary.py $(find $OUT/mfma -name "*.db" | head -1) vc:: > $OUT/pmc_mfma.txt 2>&1
GHZ=""
if [ -x tools/ubench/clock_probe ]; then tools/ubench/clock_probe 5000 > $OUT/clock_probe.txt 2>&1; GHZ=$(grep "MFMA + exp + rcp" $OUT/clock_probe.txt | sed -E 's/.*shader clock ([0-9.]+) GHz.*/\1/'); fi
python tools/mfma_util.py $OUT/pmc_mfma.txt $OUT/kernel_stats.md $GHZ > $OUT/mfma_util.md 2>&1
rm -rf $OUT/trace $OUT/mfma
tail -25 $OUT/mfma_util.md
