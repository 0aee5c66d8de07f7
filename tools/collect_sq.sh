#!/bin/bash
# Run ON THE GPU BOX: SQ / GRBM counters of the headline kernel (tools/kbench.py, 10 000 frames), one rocprofv3 --pmc
# pass per counter group (8 SQ slots per pass, MI355X_MICROARCH.md "rocprofv3 PMC slots").  tools/summarize_sq.py reads them.
#   gpurun -- 'bash tools/collect_sq.sh r02'
set -u
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag/sq
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
pass() {
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/$name" -o t -- python3 "$root/tools/kbench.py" --frames 10000 --reps 4 --path fused > "$out/$name.log" 2>&1
}
pass cycles SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
pass active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
python3 "$root/tools/summarize_sq.py" "$out"
