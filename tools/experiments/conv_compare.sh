#!/bin/bash
# per-layer conv times under several kernel variants (env knobs of conv_igemm.hip's launcher)
mkdir -p gpurun_out/cmp
run() { name=$1; shift; env "$@" timeout 100 python tools/conv_breakdown.py 2>/dev/null | grep "^conv" | sed 's/.*ms=\([0-9.]*\).*/\1/' > gpurun_out/cmp/$name.txt; }
run glds VC_CONV_GLDS=1
run reg_pd1 VC_CONV_GLDS=0 VC_CONV_PD=1
run reg_pd2 VC_CONV_GLDS=0 VC_CONV_PD=2
run reg_pd3 VC_CONV_GLDS=0 VC_CONV_PD=3
run kc8_pd1 VC_CONV_GLDS=0 VC_CONV_KC=8 VC_CONV_PD=1
VC_CONV_GLDS=1 timeout 100 python tools/conv_breakdown.py 2>/dev/null | grep "^conv" | awk '{print $2,$3,$4,$5}' > gpurun_out/cmp/shapes.txt
paste gpurun_out/cmp/shapes.txt gpurun_out/cmp/glds.txt gpurun_out/cmp/reg_pd1.txt gpurun_out/cmp/reg_pd2.txt gpurun_out/cmp/reg_pd3.txt gpurun_out/cmp/kc8_pd1.txt
