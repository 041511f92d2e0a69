#!/bin/bash
# SPPF pooling kernel alone under the detector-only workload: kernel trace with the register form and with the LDS-plane form.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/sppf; mkdir -p $OUT
export VC_TUNE_CACHE=$OUT/tune.txt VC_B=128 VC_INJECT=0 VC_OBJ_SHIFT=-8
cd /tmp
timeout 300 python $R/tools/conv_breakdown.py > /dev/null 2>&1
for mode in "VC_SPPF_SEP=0" "VC_SPPF_SEP=1 VC_SPPF_WS=16" "VC_SPPF_SEP=1 VC_SPPF_WS=32" "VC_SPPF_SEP=1 VC_SPPF_WS=8"; do
  rm -rf /tmp/sp
  env $mode timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d /tmp/sp -o t -- python $R/tools/conv_breakdown.py > $OUT/log.txt 2>&1
  echo "== $mode"
  python $R/tools/prof_summary.py $(find /tmp/sp -name "*.db" | head -1) "$mode" | grep "sppf\|head_compact\|nms_\|rank_sort\|decode_sparse" | sed -e "s/(vc::[^|]*|/ |/" | cut -c1-150
done
