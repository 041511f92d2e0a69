for shape in 128,80,80,128,128,1,1,0 128,80,80,256,128,1,1,0 128,40,40,256,256,1,1,0; do for cfg in 2 5 10 12 4 33 34 35 41 42 50 51 52 53 54; do
echo -n "shape $shape cfg $cfg: "; VC_SHAPE=$shape VC_CONV_CFG=$cfg VC_CONV_TIME=10 VC_REPS=1 timeout 120 python tools/conv_one.py 2>&1 | grep "conv time" | sed 's/.*best/best/' || echo
echo; done; done
