#!/bin/bash
# sample package power and sclk while each micro-benchmark mode runs
for mode in add mul fma mov pk_add pk_mul pk_fma; do
  ./tools/ubench/power_modes $mode 9 > /tmp/pm_$mode.log 2>&1 &
  P=$!
  sleep 5
  for i in 1 2; do
    rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk" | sed -e 's/.*: //' | tr '\n' ' '; echo
    sleep 1
  done
  wait $P
  cat /tmp/pm_$mode.log
done
