#!/bin/bash
# round 6, GPU run A: suite against the ceilings (records the errors), smoke, bench, the configs[4] / configs[2] experiments
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r06
mkdir -p $out
cd $root
FXC_TOL_MEASURE=1 timeout 2400 python3 -m pytest tests -q -m gpu -x > $out/suite_measure.log 2>&1; echo "suite rc=$?" >> $out/suite_measure.log
cp gpurun_out/observed_errors.json $out/observed_errors.json 2>/dev/null
python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/smoke.log
timeout 900 python3 bench.py > $out/bench_a.json 2> $out/bench_a.err; echo "bench rc=$?" >> $out/bench_a.err
var/copy_rate three > $out/copy_three.json 2>&1
var/copy_rate > $out/copy_rate.log 2>&1
bash tools/exp_8ant_cache.sh r06 128 > $out/exp_8ant_cache.log 2>&1
cd /tmp && export TMPDIR=/tmp
for m in default nt; do
  if [ $m = nt ]; then arg=nt; else arg=; fi
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/read16_$m -o t -- $root/var/copy_rate read16 $arg > $out/read16_$m.log 2>&1
done
python3 - $out <<'PY'
import csv, glob, sys, json
out = sys.argv[1]; res = {}
for m in ("default", "nt"):
    f = glob.glob("%s/read16_%s/**/t_counter_collection.csv" % (out, m), recursive=True)
    if f:
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == "FETCH_SIZE" and "read_kernel" in r["Kernel_Name"]]
        res[m] = {"launches": len(v), "FETCH_SIZE_raw_per_launch": sum(v) / max(1, len(v)), "bytes_read_per_launch": 4 << 30,
                  "raw_KiB_x1024_over_bytes": sum(v) / max(1, len(v)) * 1024 / (4 << 30)}
json.dump(res, open(out + "/fetch_calibration_read16.json", "w"), indent=1); print(res)
PY
