#!/bin/bash
# round 6, GPU run B: the suite against the ceilings, FXM_PIPE A/B, the F-only builds against round 5's orders
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r06
mkdir -p $out
cd $root
FXC_TOL_MEASURE=1 timeout 3000 python3 -m pytest tests -q -m gpu > $out/suite_measure.log 2>&1; echo "suite rc=$?" >> $out/suite_measure.log
cp gpurun_out/observed_errors.json $out/observed_errors.json 2>/dev/null
python3 tools/tune_spec.py --env-arms "FXC_RTC_PIPE=0;FXC_RTC_PIPE=1" --cases 1000,1200,2000,500,720,96,1536,250,1440,400 > $out/pipe_ab.jsonl 2> $out/pipe_ab.err
cases="1000:4,720:4,1536:4,2000:4,3000:4,3600:4,5000:4,6000:4,8000:4,96:4,250:4"
FXC_DEV=1 FXC_TAG=composite python3 tools/bench_channelize.py $cases > $out/fonly_ab.jsonl 2> $out/fonly_ab.err
FXC_DEV=1 FXC_TAG=prime_order FXC_RTC_COMPOSITE=0 python3 tools/bench_channelize.py $cases >> $out/fonly_ab.jsonl 2>> $out/fonly_ab.err
FXC_DEV=1 FXC_TAG=any_shape FXC_RTC=0 python3 tools/bench_channelize.py 1000:4,3000:4,6000:4 >> $out/fonly_ab.jsonl 2>> $out/fonly_ab.err
