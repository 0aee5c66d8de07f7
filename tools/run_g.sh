#!/bin/bash
# round 6, GPU run G (final kernels and stage-list table): the suite against the ceilings (for the bounds), smoke, bench, rocprofv3 of
# bench + the judged workloads
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r06g
mkdir -p $out
cd $root
FXC_TOL_MEASURE=1 timeout 3000 python3 -m pytest tests -q -m gpu > $out/suite_measure.log 2>&1; echo "suite rc=$?" >> $out/suite_measure.log
cp gpurun_out/observed_errors.json $out/observed_errors.json 2>/dev/null
python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/smoke.log
timeout 900 python3 bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/bench.err
bash tools/collect_spec.sh r06g res1000 > $out/collect_res1000.log 2>&1
bash tools/collect_spec.sh r06g res3000 > $out/collect_res3000.log 2>&1
bash tools/collect_profiles.sh r06g "stream1" > $out/collect_profiles.log 2>&1
cd $root
timeout 400 python3 tools/soak_spec.py --seconds 200 --seed 68 > $out/soak_spec.json 2> $out/soak_spec.err
timeout 300 python3 tools/soak.py --seconds 150 --seed 6007 > $out/soak.json 2> $out/soak.err
rm -rf $out/raw/*/*/*.db 2>/dev/null
