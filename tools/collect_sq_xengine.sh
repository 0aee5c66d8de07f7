#!/bin/bash
# Run ON THE GPU BOX: SQ / GRBM counters of the matrix-core X-engine (tools/bench_xengine.py, the antenna counts given),
# one rocprofv3 --pmc pass per counter group.  Prints per-kernel means.
#   gpurun -- 'bash tools/collect_sq_xengine.sh 16 32'
set -u
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/xsq
mkdir -p "$out"
ants=${*:-16 32}
cd /tmp && export TMPDIR=/tmp
pass() {
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/$name" -o t -- python3 "$root/tools/bench_xengine.py" --ants $ants --reps 2 > "$out/$name.log" 2>&1
}
pass cycles SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_F32
pass active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
python3 - "$out" <<'PY'
import csv, glob, os, sys, collections
base = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for cc in glob.glob(os.path.join(base, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(cc, newline="")):
        if "xengine_mfma" in row["Kernel_Name"]:
            t = row["Kernel_Name"].split("<")[1][:1]
            vals[t][row["Counter_Name"]].append(float(row["Counter_Value"]))
            vals[t]["_us_" + row["Counter_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for t in sorted(vals):
    print("== xengine_mfma_kernel<%s>" % t)
    for k in sorted(vals[t]):
        v = vals[t][k]
        print("   %-34s %14.1f  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
