#!/bin/bash
# round 6, GPU run C: rocprofv3 passes behind profiles/r06/, soaks, first-use latency, the suite under the regenerated bounds
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r06
mkdir -p $out
cd $root
export FXC_RTC_CACHE=$root/gpurun_out/r06/rtc_cache_tmp     # (a cache the probes below start cold from)
for n in 1000 3000 1080 4096; do python3 tools/probe_first_use.py $n >> $out/first_use.jsonl 2>> $out/first_use.err; done
python3 tools/probe_first_use.py 1080 >> $out/first_use.jsonl 2>> $out/first_use.err      # (second process: the run-time cache)
rm -rf $root/gpurun_out/r06/rtc_cache_tmp
unset FXC_RTC_CACHE
bash tools/collect_spec.sh r06 res1000 > $out/collect_res1000.log 2>&1
bash tools/collect_spec.sh r06 res3000 > $out/collect_res3000.log 2>&1
bash tools/collect_spec.sh r06 nfft8192 f8192_ring_kernel > $out/collect_nfft8192.log 2>&1
bash tools/collect_profiles.sh r06 "stream1 8ant" > $out/collect_profiles.log 2>&1
cd $root
timeout 600 python3 tools/soak.py --seconds 300 --seed 6006 > $out/soak.json 2> $out/soak.err
timeout 500 python3 tools/soak_spec.py --seconds 240 --seed 66 > $out/soak_spec.json 2> $out/soak_spec.err
timeout 400 python3 tools/soak_spec.py --seconds 180 --seed 67 --min-nchan 2049 --max-nchan 4097 > $out/soak_spec_lean.json 2> $out/soak_spec_lean.err
timeout 2400 python3 -m pytest tests -q -m gpu > $out/suite_bounded.log 2>&1; echo "suite rc=$?" >> $out/suite_bounded.log
rm -rf $out/raw/*/*/*.db 2>/dev/null
du -sh $out
