#!/usr/bin/env python3
"""Developer aid: BASELINE configs[4] (8 antennas, 4096 channels, 512 chunks of 2^18 samples: F pass + X-engine) timed end to end with
the workspace bound the process was started under (FXC_WS_MB: the library runs the call in passes of the chunks whose spectra fit) and
the library FXCORR_LIB names -- the experiment of VERDICT r05 item 7(b): do batches whose spectra fit the 256 MiB Infinity Cache, with
default-policy stores and loads on the spectra, beat one pass of everything with nontemporal ones?

    FXC_WS_MB=128 FXCORR_LIB=var/libfxcorr_defpol.so python tools/exp_8ant_cache.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from effex_amd.plan import FxPlan, synth_fill

n_chunks, n_ant, nchan, num_samp = 512, 8, 4096, 2 ** 18
x = torch.empty((n_chunks, n_ant, num_samp), dtype=torch.complex64, device="cuda")
synth_fill(x, 1234)
with FxPlan(n_ant, nchan, 4, num_samp) as plan:
    def fn():
        plan.acc_reset()
        plan.fx_accumulate(x)
    fn()
    res0 = plan.finalize("SPECTRUM")
    ts = []
    for _ in range(9):
        plan.timer_start()
        fn()
        ts.append(plan.timer_stop())
        plan.finalize("SPECTRUM")
    ts.sort()
    ms = ts[len(ts) // 2]
    gb = n_chunks * n_ant * num_samp * 8 / 1e9
    print(json.dumps({"ws_mb": os.environ.get("FXC_WS_MB", "default"), "lib": os.path.basename(os.environ.get("FXCORR_LIB", "libfxcorr.so")),
                      "ms": round(ms, 3), "min_ms": round(ts[0], 3), "frac_of_8TBs": round(gb / ms * 1e3 / 8000, 4),
                      "workspace_MB": plan.info["workspace_bytes"] >> 20, "checksum": float(abs(res0).sum())}), flush=True)
