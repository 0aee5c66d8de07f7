#!/usr/bin/env python3
"""Developer aid: receiver bytes with DC removal on the plans that convert first (3+ antennas, more than four taps, 8192 channels,
more than 4096 channels off the powers of two): F + X from bytes, and the conversion pass alone; 4.3 GB-equivalent of samples."""
import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from effex_amd.plan import FxPlan
num_samp = 2 ** 18
for n_ant, nchan, ntaps in ((8, 4096, 4), (2, 2048, 32), (2, 8192, 4), (2, 5000, 4), (4, 1000, 4)):
    n_chunks = 2048 // n_ant
    u8 = torch.randint(0, 256, (n_chunks, n_ant, num_samp, 2), dtype=torch.uint8, device="cuda")
    with FxPlan(n_ant, nchan, ntaps, num_samp) as plan:
        plan.fx_accumulate_u8(u8, remove_dc=True); plan.finalize()
        ms = []
        for _ in range(5):
            plan.timer_start(); plan.fx_accumulate_u8(u8, remove_dc=True); ms.append(plan.timer_stop()); plan.finalize()
        ms.sort()
        c = plan.convert_u8(u8, remove_dc=True); plan.sync()
        plan.timer_start(); c = plan.convert_u8(u8, remove_dc=True); conv = plan.timer_stop()
        print(json.dumps({"n_ant": n_ant, "nchan": nchan, "ntaps": ntaps, "path": plan.path, "fx_u8_ms": round(ms[2], 3), "convert_only_ms": round(conv, 3)}), flush=True)
    del u8
