#!/bin/bash
# Run ON THE GPU BOX (developer aid): stage lists / work-item groups / LDS layouts of fx_spec.h against each other -- every line one
# child of tools/bench_spec.py on the developer library (the FXC_RTC_* knobs exist there only), checked against the oracle.
#   gpurun -- 'bash tools/sweep_spec_lists.sh gpurun_out/sweep.log "1000:4,2,5,5,5 1000:4,25,10 3000:6,10,10,5" [extra env ...]'
log=${1:-gpurun_out/sweep_lists.log}; shift
cases=${1:-"1000:4,2,5,5,5"}; shift
mkdir -p "$(dirname "$log")"
for c in $cases; do
  n=${c%%:*}; r=${c#*:}
  echo "== $n [$r] $*" >> "$log"
  if [ "$r" = "auto" ]; then env "$@" FXC_RTC=1 python3 tools/bench_spec.py --child --dev --check --cases $n >> "$log" 2>&1
  else env "$@" FXC_RTC=1 FXC_RTC_RADICES=$r python3 tools/bench_spec.py --child --dev --check --cases $n >> "$log" 2>&1; fi
done
