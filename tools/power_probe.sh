#!/bin/bash
# Developer probe: sample rocm-smi power / clocks while the headline kernel runs (kbench loop in the background).
mkdir -p gpurun_out
python tools/kbench.py --frames 10000 --reps ${1:-4000} > gpurun_out/power_kbench.log 2>&1 &
KB=$!
while kill -0 $KB 2>/dev/null; do
  rocm-smi --showpower --showclocks --showuse 2>&1 | grep -E "Power|sclk|mclk|GPU use" | tr '\n' ' '
  echo
  sleep 3
done
wait $KB
tail -3 gpurun_out/power_kbench.log
echo "=== idle"
rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk"
rocm-smi --showmaxpower 2>&1 | grep -i -E "power" | head
