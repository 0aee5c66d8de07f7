#!/bin/bash
# Run ON THE GPU BOX: timing ablations of the matrix-core X-engine (FXC_XMFMA_ABL bits: 2 no multiplies, 4 no fetches,
# 8 no park / barrier; wrong results by design), kernel times from a rocprofv3 trace.
#   gpurun -- 'bash tools/abl_xmfma.sh 0 2 4 28'
for v in ${*:-0 2 4 8 16 6 28}; do
  export FXC_XMFMA_ABL=$v
  XMODES=mfma bash tools/prof_xengine.sh 16 32 48 64 > /dev/null 2>&1
  python3 - $v <<'PY'
import csv, collections, sys, os
root=os.environ.get('GRAFT_REPO_ROOT','.')
rows=list(csv.DictReader(open(root+'/gpurun_out/xengine/mfma/t_kernel_trace.csv')))
agg=collections.defaultdict(list)
for r in rows:
    name=r['Kernel_Name']
    if 'xengine_mfma' in name:
        agg[name.split('<')[1][:1]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
print('abl', sys.argv[1], {k: round(sorted(v)[len(v)//2],3) for k,v in sorted(agg.items())})
PY
done
