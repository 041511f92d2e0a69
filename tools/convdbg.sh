#!/bin/bash
# per-workgroup phase times of one conv shape (VC_CONV_DBG) vs tile config / workgroups per CU
shape=${1:-16,80,80,64,64,3,1,1}
for cfg in ${CFGS:-3 14}; do for lds in ${LDS:-0 122880}; do
  echo -n "cfg $cfg dyn_lds $lds: "; VC_SHAPE=$shape VC_CONV_CFG=$cfg VC_CONV_DYN_LDS=$lds VC_CONV_DBG=1 VC_REPS=3 timeout 60 python tools/conv_one.py 2>&1 | grep "conv dbg" | tail -2
done; done
