#!/bin/bash
# A/B of 8-antenna route variants ON THE GPU BOX: per-kernel times from a rocprofv3 kernel trace of
# tools/prof_workload.py 8ant for every library under var/ named on the command line (FXCORR_LIB selects the build).
#   gpurun -- 'bash tools/ab_8ant.sh r03x base old_layout w2u1 ...'
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
for v in "$@"; do
  FXCORR_LIB=$root/var/libfxcorr_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${v}_$round" -o t -- python3 "$root/tools/prof_workload.py" 8ant 12 > "$out/${v}_$round.log" 2>&1
  python3 - "$out/${v}_$round/t_kernel_trace.csv" "$v" "$round" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"]
    key = "F" if "fx_fused4096" in name else ("X" if "xengine" in name else ("fold" if "fold_" in name else None))
    if key:
        d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
def tail_mean(v, per_call):   # per call (a call = per_call launches), last 6 calls
    calls = [sum(v[i:i + per_call]) for i in range(0, len(v) - per_call + 1, per_call)]
    calls = calls[-6:]
    return sum(calls) / len(calls)
nF = len(d["F"]) // 12; nX = len(d["X"]) // 12; nf = max(1, len(d["fold"]) // 12)
F = tail_mean(d["F"], nF); X = tail_mean(d["X"], nX); fo = tail_mean(d["fold"], nf)
gb = 512 * 8 * 262144 * 8 / 1e9
print("%-12s round %s: F-only %.1f us  X-engine %.1f us  fold %.1f us  sum %.1f us -> %.3f TB/s algorithmic = %.4f of 8 TB/s" %
      (sys.argv[2], sys.argv[3], F, X, fo, F + X + fo, gb / (F + X + fo) * 1e3, gb / (F + X + fo) * 1e3 / 8))
PY
done
done
