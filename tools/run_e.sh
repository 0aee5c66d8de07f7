#!/bin/bash
# round 6, GPU run E: last look at the stage lists of the two judged shapes under the final layouts, the suite against the ceilings (for the
# bounds), the bench line
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r06
mkdir -p $out
cd $root
python3 tools/tune_spec.py --picks 0 --lists "4,10,5,5;4,5,10,5;4,5,5,10;4,25,10;4,10,25;5,10,20;6,25,4,5;6,10,10,5;6,5,10,10;3,10,10,10;6,4,25,5;6,5,25,4;6,25,5,4;6,20,5,5" --cases 1000,3000 > $out/lists_final.jsonl 2> $out/lists_final.err
FXC_TOL_MEASURE=1 timeout 3000 python3 -m pytest tests -q -m gpu > $out/suite_measure2.log 2>&1; echo "suite rc=$?" >> $out/suite_measure2.log
cp gpurun_out/observed_errors.json $out/observed_errors2.json 2>/dev/null
timeout 900 python3 bench.py > $out/bench_e.json 2> $out/bench_e.err; echo "bench rc=$?" >> $out/bench_e.err
