for cfg in 55 59; do for abl in 0 1 2 3 4 8 16 19 20 23 27 31; do
echo -n "cfg $cfg ablate $abl: "; VC_SHAPE=128,40,40,128,128,3,1,1 VC_CONV_CFG=$cfg VC_CONV_ABLATE=$abl VC_CONV_TIME=10 VC_REPS=1 timeout 120 python tools/conv_one.py 2>&1 | grep "conv time" | sed 's/.*best/best/'
done; done
