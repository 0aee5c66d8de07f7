#!/bin/bash
# A/B kernel variants: tools/ab.sh <frames> <reps> lib1 lib2 ...   (libs under build/variants/)
frames=$1; reps=$2; shift 2
for round in 1 2; do
for v in "$@"; do
  FXCORR_LIB=$PWD/build/variants/$v python tools/kbench.py --frames $frames --reps $reps --tag $v
done
done
