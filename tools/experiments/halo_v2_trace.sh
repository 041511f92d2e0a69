for shape in 128,40,40,128,128,3,1,1 1536,13,13,128,128,3,1,1; do
echo "== $shape"; VC_SHAPE=$shape VC_CONV_CFG=55 VC_CONV_DBG=1 VC_REPS=2 timeout 120 python tools/conv_one.py 2>&1 | grep "conv dbg" | tail -11
done
