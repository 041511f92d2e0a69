for cfg in 61; do for abl in 0 32 33 48 49; do
echo -n "cfg $cfg ablate $abl: "; VC_SHAPE=128,40,40,128,128,3,1,1 VC_CONV_CFG=$cfg VC_HALO_PS_DELAY=0 VC_CONV_ABLATE=$abl VC_CONV_TIME=10 VC_REPS=1 timeout 120 python tools/conv_one.py 2>&1 | grep "conv time" | sed 's/.*best/best/'
done; done
VC_SHAPE=128,40,40,128,128,3,1,1 VC_CONV_CFG=61 VC_HALO_PS_DELAY=0 VC_CONV_ABLATE=32 VC_CONV_DBG=1 VC_REPS=2 timeout 120 python tools/conv_one.py 2>&1 | grep "conv dbg" | tail -3
