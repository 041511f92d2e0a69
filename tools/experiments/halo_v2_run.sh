#!/bin/bash
# conv3x3_halo_v2_kernel (cfg 55): identity + time against cfg 39 / 30 in Bottleneck form, then plain (no residual) through tools/conv_one.py
CFGS=39,30,55 python tools/experiments/halo_compare.py
for shape in 128,40,40,128,128,3,1,1 128,20,20,256,256,3,1,1 1536,13,13,128,128,3,1,1; do for cfg in 39 55; do
echo -n "shape $shape cfg $cfg: "; VC_SHAPE=$shape VC_CONV_CFG=$cfg VC_CONV_TIME=10 VC_REPS=1 timeout 120 python tools/conv_one.py 2>&1 | grep "conv time" | sed 's/.*best/best/'
done; done
