for cfg in 39 61 62; do for grid in 512; do for delay in 0 400; do
echo "== cfg $cfg delay $delay"; VC_SHAPE=128,40,40,128,128,3,1,1 VC_CONV_CFG=$cfg VC_HALO_PS_GRID=$grid VC_HALO_PS_DELAY=$delay VC_CONV_DBG=1 VC_REPS=2 timeout 120 python tools/conv_one.py 2>&1 | grep "conv dbg" | tail -3
done; done; done
