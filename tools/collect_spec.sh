#!/bin/bash
# Run ON THE GPU BOX: the specialised F+X kernel (fx_spec.h) at --resolution 1000 (tools/prof_workload.py res1000): kernel
# trace + stats, FETCH_SIZE / WRITE_SIZE / GRBM_GUI_ACTIVE in --pmc passes of their own (MI355X_MICROARCH.md), and the SQ counter
# groups (8 slots per pass).  tools/summarize_profiles.py + tools/summarize_sq_any.py turn them into the files kept under profiles/.
#   gpurun -- 'bash tools/collect_spec.sh r05 res1000'
set -u
tag=${1:-r05}
w=${2:-res1000}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag/raw
sq=$root/gpurun_out/$tag/sq_$w
mkdir -p "$out" "$sq"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${w}_trace" -o t -- python3 "$root/tools/prof_workload.py" $w 10 > "$out/${w}_trace.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/${w}_$c" -o t -- python3 "$root/tools/prof_workload.py" $w 10 > "$out/${w}_$c.log" 2>&1
done
pass() {
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$sq/$name" -o t -- python3 "$root/tools/prof_workload.py" $w 6 > "$sq/$name.log" 2>&1
}
pass cycles SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
pass active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
python3 "$root/tools/summarize_profiles.py" "$root/gpurun_out/$tag"
python3 "$root/tools/summarize_sq_any.py" "$sq" "${3:-fxm_fx2_kernel}" > "$root/gpurun_out/$tag/sq_$w.json"
cat "$root/gpurun_out/$tag/sq_$w.json"
