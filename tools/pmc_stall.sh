#!/bin/bash
# Where the conv kernels' wave-cycles go: two SQ counter passes over a short bench run, summed per kernel template.
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/stall
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
export VC_TUNE_CACHE=$OUT/tune.txt
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd /tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $OUT/p1 -o p1 -- $CMD > $OUT/p1.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/p2 -o p2 -- $CMD > $OUT/p2.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT/p1/p1_results.db vc::conv > $OUT/p1.txt 2>&1
python tools/pmc_summary.py $OUT/p2/p2_results.db vc::conv > $OUT/p2.txt 2>&1
python tools/pmc_summary.py $OUT/p1/p1_results.db stem_ >> $OUT/p1.txt 2>&1
python tools/pmc_summary.py $OUT/p2/p2_results.db stem_ >> $OUT/p2.txt 2>&1
rm -rf $OUT/p1 $OUT/p2
wc -l $OUT/p1.txt $OUT/p2.txt; tail -3 $OUT/p1.log
