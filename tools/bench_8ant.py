import os, sys, json
sys.path.insert(0, os.getcwd())
import torch
from effex_amd.plan import FxPlan, synth_fill
for nchan, n_chunks in ((1024, 256), (2048, 256), (4096, 256)):
    x = torch.empty((n_chunks, 8, 262144), dtype=torch.complex64, device="cuda")
    synth_fill(x, 1234)
    with FxPlan(8, nchan, 4, 262144) as plan:
        def fn():
            plan.acc_reset(); plan.fx_accumulate(x); plan.finalize("SPECTRUM")
        fn(); plan.sync()
        ts = []
        for _ in range(5):
            plan.timer_start(); fn(); ts.append(plan.timer_stop())
        ts.sort(); ms = ts[2]
        print(json.dumps({"nchan": nchan, "path": plan.path, "ms": round(ms, 3), "algorithmic_GBps": round(n_chunks * 8 * 262144 * 8 / ms / 1e6, 1)}))
    del x
