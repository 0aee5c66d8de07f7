import sys, os
sys.path.insert(0, os.environ.get("R", "."))
import torch
from effex_amd.plan import FxPlan, synth_fill
x = torch.empty((128, 8, 262144), dtype=torch.complex64, device="cuda")
synth_fill(x, 1)
p = FxPlan(8, 4096, 4, 262144)
for _ in range(4):
    p.acc_reset(); p.fx_accumulate(x); p.finalize("SPECTRUM")
