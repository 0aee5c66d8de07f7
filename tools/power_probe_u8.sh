#!/bin/bash
# Developer probe: package power / sclk while the uint8-ingest kernel loops
python - <<'PY' &
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from effex_amd.plan import FxPlan
u8 = torch.randint(0, 256, (10000, 2, 262144, 2), dtype=torch.uint8, device="cuda")
with FxPlan(2, 4096, 4, 262144) as plan:
    for _ in range(2500):
        plan.fx_accumulate_u8(u8, remove_dc=False)
    plan.sync()
PY
KB=$!
sleep 12
while kill -0 $KB 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk" | sed -e 's/.*: //' | tr '\n' ' '; echo
  sleep 2
done
