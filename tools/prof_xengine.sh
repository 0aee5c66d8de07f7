#!/bin/bash
# Run ON THE GPU BOX (inside gpurun): kernel-trace statistics of tools/bench_xengine.py for the matrix-core X-engine and
# for the vector kernel it replaced (FXC_XENGINE=block), the shapes given as arguments (default: 16 32 64 antennas).
#   gpurun -- 'bash tools/prof_xengine.sh 16 32 64'
set -u
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/xengine
mkdir -p "$out"
ants=${*:-16 32 64}
cd /tmp && export TMPDIR=/tmp
for mode in ${XMODES:-mfma block}; do
  if [ $mode = block ]; then export FXC_XENGINE=block; else unset FXC_XENGINE; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$mode" -o t -- python3 "$root/tools/bench_xengine.py" --ants $ants --reps 3 > "$out/$mode.log" 2>&1
  f=$(find "$out/$mode" -name '*kernel_stats.csv' | head -1)
  echo "== $mode"; head -12 "$f"
done
