#!/bin/bash
# What a wide pointwise layer's time is made of: kernel duration per tile configuration with the staging loads removed (VC_CONV_ABLATE=1),
# the stores removed (2), both (3), the launch floor (6), and without the SiLU (VC_ACT=0).
export TMPDIR=/tmp
for shape in ${SHAPES:-128,40,40,256,256,1,1,0 128,80,80,256,128,1,1,0}; do
for act in 1 0; do
  export VC_ACT=$act
  echo "== act $act"
  CFGS="${CFGS:-42 65 5}" ABL="${ABL:-0 1 2 3 6}" bash tools/ablate.sh $shape
done
done
