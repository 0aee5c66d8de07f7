#!/bin/bash
# round 6, GPU run D: LDS layout and early-table A/B, SQ counters under the new layouts, the two-pass route above 4096 channels, first use
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r06
mkdir -p $out
cd $root
python3 -m pytest tests/test_gpu_parity.py -q -x -k "above_4096 or any_channel_count or new_routes" > $out/suite_xm.log 2>&1; echo "rc=$?" >> $out/suite_xm.log
python3 tools/tune_spec.py --env-arms "FXC_RTC_LAYOUT=0;FXC_RTC_LAYOUT=1" --cases 1000,3000,1200,2000,720,1536,500,2400 > $out/layout_ab.jsonl 2> $out/layout_ab.err
python3 tools/tune_spec.py --env-arms "FXC_RTC_TW_EARLY=0;FXC_RTC_TW_EARLY=1" --cases 3000,4000,2400,3600,2560 > $out/tw_early_ab.jsonl 2> $out/tw_early_ab.err
FXC_RTC=1 python3 tools/bench_spec.py --child --check --cases 5000,6000,4500,7000,6561,8000 > $out/xm_ab.jsonl 2> $out/xm_ab.err
FXC_RTC=1 FXC_XM=0 python3 tools/bench_spec.py --child --dev --check --cases 5000,6000,4500,7000,6561,8000 >> $out/xm_ab.jsonl 2>> $out/xm_ab.err
export FXC_RTC_CACHE=$out/rtc_cache_tmp
for n in 1000 3000 6000 1080; do python3 tools/probe_first_use.py $n >> $out/first_use2.jsonl 2>> $out/first_use2.err; done
python3 tools/probe_first_use.py 1080 >> $out/first_use2.jsonl 2>> $out/first_use2.err
rm -rf $out/rtc_cache_tmp; unset FXC_RTC_CACHE
mkdir -p $root/gpurun_out/r06b
bash tools/collect_spec.sh r06b res1000 > $out/collect_res1000_b.log 2>&1
bash tools/collect_spec.sh r06b res3000 > $out/collect_res3000_b.log 2>&1
bash tools/collect_spec.sh r06b res6000 > $out/collect_res6000_b.log 2>&1
rm -rf $root/gpurun_out/r06b/raw/*/*/*.db 2>/dev/null
