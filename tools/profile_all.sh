#!/bin/bash
# A round's whole committed profile set, in the order the bench line needs it (run on the GPU box through gpurun, from the repo root):
#   1. detector-only MFMA busy (north_star's quantity) -> profiles/<rNN>_detector_only_{mfma_util,kernel_stats}.md, quoted by bench.py as
#      roofline.backbone_mfma_busy;  2. tools/profile_round.sh <rNN> (trace, PMC traffic, MFMA busy of the whole step, per-layer table, bench line).
# usage: bash tools/profile_all.sh r06      (results are also collected under gpurun_out/profiles_<rNN>/ for the way back)
R=${1:?round tag, e.g. r06}
cd $GRAFT_REPO_ROOT
bash tools/detector_only_mfma.sh $GRAFT_REPO_ROOT/gpurun_out/detonly_$R > gpurun_out/detonly_$R.log 2>&1
cp gpurun_out/detonly_$R/mfma_util.md profiles/${R}_detector_only_mfma_util.md
cp gpurun_out/detonly_$R/kernel_stats.md profiles/${R}_detector_only_kernel_stats.md
[ -f gpurun_out/detonly_$R/clock_probe.txt ] && cp gpurun_out/detonly_$R/clock_probe.txt profiles/${R}_clock_probe.txt
bash tools/profile_round.sh $R > gpurun_out/profile_round_$R.log 2>&1
mkdir -p gpurun_out/profiles_$R && cp profiles/${R}_* gpurun_out/profiles_$R/
tail -5 profiles/${R}_detector_only_mfma_util.md
tail -3 profiles/${R}_mfma_util.md
tail -c 1500 profiles/${R}_bench_n1.json
