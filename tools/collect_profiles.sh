#!/bin/bash
# Run ON THE GPU BOX (inside gpurun): every rocprofv3 pass behind profiles/rNN/ — kernel trace + stats, and FETCH_SIZE /
# WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md: they do not fit one pass), GRBM_GUI_ACTIVE (effective clock) in a
# third — for bench.py and the named
# workloads of tools/prof_workload.py.  Raw output under gpurun_out/$1/raw; tools/summarize_profiles.py makes the summaries.
#   gpurun -- 'bash tools/collect_profiles.sh r03 ["8ant stream1 ..."]'
set -u
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag/raw
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
run() {   # name, program args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${name}_trace" -o t -- python3 "$@" > "$out/${name}_trace.log" 2>&1
  for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/${name}_$c" -o t -- python3 "$@" > "$out/${name}_$c.log" 2>&1
  done
}
# the driver's own command line (20 timed steps after 5 warm-up steps), minus the CPU leg and the untimed extras
run bench "$root/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-power --no-other-configs --no-verify
for w in ${2:-8ant stream1 nfft2048 taps32 nfft256 nfft16 16ant 32ant res1000}; do run $w "$root/tools/prof_workload.py" $w 10; done
python3 "$root/tools/summarize_profiles.py" "$root/gpurun_out/$tag"
