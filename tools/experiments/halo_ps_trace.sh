for abl in 0 16; do
echo "== ablate $abl"; VC_SHAPE=128,40,40,128,128,3,1,1 VC_CONV_CFG=61 VC_HALO_PS_DELAY=0 VC_CONV_ABLATE=$abl VC_CONV_DBG=1 VC_REPS=2 timeout 120 python tools/conv_one.py 2>&1 | grep "conv dbg" | tail -11
done
