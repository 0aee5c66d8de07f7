#!/bin/bash
# Run ON THE GPU BOX: per-kernel times of the generic path (channel counts off the tuned kernels) from a rocprofv3 kernel
# trace of tools/bench_taps.py.  usage: tools/prof_generic.sh "1000:4 96:4 5000:4"
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/prof_generic"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for c in ${1:-1000:4}; do
  tag=${c/:/_}
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$tag" -o t -- python3 "$root/tools/bench_taps.py" --reps 3 --cases "$c" > "$out/$tag.log" 2>&1 < /dev/null
  f=$(find "$out/$tag" -name "*kernel_stats.csv" | head -1)
  echo "== $c"
  if [ -n "$f" ]; then head -7 "$f" | cut -d, -f1-5 | cut -c1-160; else echo "no stats file"; tail -3 "$out/$tag.log"; fi
done
