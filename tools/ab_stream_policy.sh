#!/bin/bash
# usage: ab.sh  (runs base then variant)
R=$GRAFT_REPO_ROOT
for L in base st base st; do
  if [ $L = base ]; then unset FXCORR_LIB; else export FXCORR_LIB=$R/build/libfxcorr_$L.so; fi
  echo "== $L"
  python $R/tools/kbench.py --frames 2048 --reps 6 --nchan 2048 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kbench2048', d['median_ms'])"
  python $R/tools/bench_spec.py --cases 1000,1000,3000,8192,8192 2>&1 | grep '"rtc"' | python -c "
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); print('spec', d['nchan'], d['median_ms'])"
  python $R/tools/bench_channelize.py 8192:4,8192:4,4096:4,2048:4,1000:4,3000:4 2>&1 | grep nchan | python -c "
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); print('chan', d['nchan'], d['median_ms'])"
  python $R/tools/bench_ants.py 2048 8 2>&1 | grep n_ant | python -c "
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); print('ants2048', d['n_ant'], d['median_ms'])"
  python $R/tools/bench_ants.py 1000 8 2>&1 | grep n_ant | python -c "
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); print('ants1000', d['n_ant'], d['median_ms'])"
  python $R/tools/bench_ants.py 4096 8 2>&1 | grep n_ant | python -c "
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); print('ants4096', d['n_ant'], d['median_ms'])"
done
