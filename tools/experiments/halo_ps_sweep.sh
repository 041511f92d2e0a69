#!/bin/bash
# conv3x3_halo_ps_kernel: identity + time against the non-persistent halo kernels, then start-delay / grid sweeps on two shapes
CFGS=39,30,61,62,63,64 python tools/experiments/halo_pf_compare.py
for shape in 128,40,40,128,128,3,1,1 128,20,20,256,256,3,1,1 1536,13,13,128,128,3,1,1; do
for cfg in 61 62; do for grid in 512 448 384; do for delay in 0 200 400 700 1000; do
echo -n "shape $shape cfg $cfg grid $grid delay $delay: "; VC_SHAPE=$shape VC_CONV_CFG=$cfg VC_HALO_PS_GRID=$grid VC_HALO_PS_DELAY=$delay VC_CONV_TIME=10 VC_REPS=1 timeout 120 python tools/conv_one.py 2>&1 | grep "conv time" | sed 's/.*best/best/'
done; done; done; done
