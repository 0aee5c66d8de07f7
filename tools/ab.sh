#!/bin/bash
# A/B kernel variants: tools/ab.sh <frames> <reps> lib1 lib2 ...   (libs under var/: build/ is not shipped to the GPU box)
frames=$1; reps=$2; shift 2
for round in 1 2; do
for v in "$@"; do
  FXCORR_LIB=$PWD/var/$v python tools/kbench.py --frames $frames --reps $reps --tag $v
done
done
