#!/bin/bash
# A/B the nchan = 1 streaming kernel variants under build/variants/
for round in 1 2; do
for v in "$@"; do
  FXCORR_LIB=$PWD/build/variants/$v python tools/kbench.py --nchan 1 --num-samp 1048576 --frames 2048 --reps 10 --rows --tag $v
done
done
