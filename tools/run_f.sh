#!/bin/bash
# round 6, GPU run F: the two-pass route from the first call on; the suite against the ceilings (for the bounds); bench; res6000 under rocprofv3
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r06
mkdir -p $out
cd $root
FXC_TOL_MEASURE=1 timeout 3000 python3 -m pytest tests -q -m gpu > $out/suite_measure3.log 2>&1; echo "suite rc=$?" >> $out/suite_measure3.log
cp gpurun_out/observed_errors.json $out/observed_errors3.json 2>/dev/null
timeout 900 python3 bench.py > $out/bench_f.json 2> $out/bench_f.err; echo "bench rc=$?" >> $out/bench_f.err
FXC_RTC=1 python3 tools/bench_spec.py --child --check --cases 5000,6000,4500,7000,6561 > $out/xm_final.jsonl 2> $out/xm_final.err
bash tools/collect_spec.sh r06b res6000 > $out/collect_res6000_c.log 2>&1
rm -rf $root/gpurun_out/r06b/raw/*/*/*.db 2>/dev/null
