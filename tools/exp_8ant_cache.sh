#!/bin/bash
# Run ON THE GPU BOX: tools/exp_8ant_cache.py over workspace bounds x {shipped library (nontemporal spectra stores / loads), var/libfxcorr_defpol.so
# (default policy on the spectra)}, then FETCH_SIZE / WRITE_SIZE of the best batched arm.   gpurun -- 'bash tools/exp_8ant_cache.sh r06'
tag=${1:-r06}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p "$out"
log=$out/exp_8ant_cache.jsonl
for lib in "" "$root/var/libfxcorr_defpol.so"; do
  for ws in default 1024 384 256 192 128 96; do
    if [ "$ws" = default ]; then unset FXC_WS_MB; else export FXC_WS_MB=$ws; fi
    if [ -n "$lib" ]; then FXCORR_LIB=$lib python3 "$root/tools/exp_8ant_cache.py" >> "$log" 2>> "$out/exp_8ant_cache.err"
    else python3 "$root/tools/exp_8ant_cache.py" >> "$log" 2>> "$out/exp_8ant_cache.err"; fi
  done
done
cat "$log"
cd /tmp && export TMPDIR=/tmp
export FXC_WS_MB=${2:-128}
for c in FETCH_SIZE WRITE_SIZE; do
  FXCORR_LIB=$root/var/libfxcorr_defpol.so rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/8ant_ws${FXC_WS_MB}_$c" -o t -- python3 "$root/tools/prof_workload.py" 8ant 6 > "$out/8ant_ws${FXC_WS_MB}_$c.log" 2>&1
done
python3 - "$out" "$FXC_WS_MB" <<'PY'
import csv, sys, glob, json, collections
out, ws = sys.argv[1], sys.argv[2]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("%s/8ant_ws%s_%s/**/t_counter_collection.csv" % (out, ws, c), recursive=True)
    if not f: continue
    tot = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == c:
            tot[r["Kernel_Name"].split("(")[0][:60]] += float(r["Counter_Value"]); n[r["Kernel_Name"].split("(")[0][:60]] += 1
    res[c] = {k: {"sum": v, "dispatches": n[k]} for k, v in tot.items() if v > 1e6}
json.dump(res, open("%s/8ant_ws%s_pmc.json" % (out, ws), "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
