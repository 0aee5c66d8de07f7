#!/bin/bash
# Run ON THE GPU BOX (developer aid): for each channel count the k-th ranked stage list of h_rtc.h::spec_stage_lists (FXC_RTC_PICK, developer
# library) for k = 0 .. $3, and the prime-factor order of round 5 (FXC_RTC_COMPOSITE=0), each checked against the oracle and timed.
#   gpurun -- 'bash tools/sweep_spec_rank.sh gpurun_out/r06/rank.log "1000 3000 1536" 3'
log=${1:-gpurun_out/sweep_rank.log}
mkdir -p "$(dirname "$log")"
for n in ${2:-1000}; do
  echo "== $n legacy" >> "$log"
  FXC_RTC=1 FXC_RTC_COMPOSITE=0 FXC_RTC_VERBOSE=1 python3 tools/bench_spec.py --child --dev --check --cases $n ${4:-} >> "$log" 2>&1
  for k in $(seq 0 ${3:-3}); do
    echo "== $n pick$k" >> "$log"
    FXC_RTC=1 FXC_RTC_PICK=$k FXC_RTC_VERBOSE=1 python3 tools/bench_spec.py --child --dev --check --cases $n ${4:-} >> "$log" 2>&1
  done
done
