#!/bin/bash
# Streaming read with K v_fma_f32 per 16-byte load under the package power cap: the rate a read-once kernel can hold as its
# arithmetic per byte grows (what a 32-tap FIR in the F+X kernel would cost).  Run on the GPU box:
#   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/power_modes tools/ubench/power_modes.hip && bash tools/ubench/power_stream_valu.sh
for k in 0 12 24 48 75 94 140 178 222; do
  ./tools/ubench/power_modes fma_read 7 $k > /tmp/psv_$k.log 2>&1 &
  P=$!
  sleep 4
  S=$(rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk" | sed -e 's/.*: //' | tr '\n' ' ')
  wait $P
  echo "K=$k $(cat /tmp/psv_$k.log | tail -1) | $S"
done
