#!/bin/bash
# Idle gaps of the detector queue in the bench's steady state (kernel trace of a 30-step run): bash tools/experiments/queue_gaps.sh
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export VC_TUNE_CACHE=/tmp/tune_qg.txt
cd /tmp
python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
rm -rf /tmp/qg
timeout -s KILL 400 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/qg -o t -- python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras > /tmp/qg.log 2>&1
python $R/tools/queue_gaps.py $(ls /tmp/qg/*.db | head -1) front_fused pass; python $R/tools/experiments/handover.py $(ls /tmp/qg/*.db | head -1)
