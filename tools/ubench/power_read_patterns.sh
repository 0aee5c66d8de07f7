#!/bin/bash
# package power and sclk while a read-only kernel streams 8 GiB over and over: linear walks against the fused kernel's pattern
for mode in read read8 read_lin8 read_fx; do
  ./tools/ubench/power_modes $mode 9 > /tmp/pm_$mode.log 2>&1 &
  P=$!
  sleep 5
  for i in 1 2 3; do
    rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk" | sed -e 's/.*: //' | tr '\n' ' '; echo
    sleep 1
  done
  wait $P
  cat /tmp/pm_$mode.log
done
