#!/bin/bash
# A/B kernel variants: tools/ab.sh <frames> <reps> v1 v2 ...   (libraries var/libfxcorr_<v>.so: build/ is not shipped to the GPU box)
frames=$1; reps=$2; shift 2
for round in 1 2 3; do
for v in "$@"; do
  FXCORR_LIB=$PWD/var/libfxcorr_$v.so python tools/kbench.py --frames $frames --reps $reps --tag $v
done
done
