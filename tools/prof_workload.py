#!/usr/bin/env python3
"""Developer aid: one named workload a few times, for rocprofv3 (kernel trace / PMC passes; tools/collect_profiles.sh).

    python3 tools/prof_workload.py 8ant | stream1 | nfft2048 | taps32 | nfft256 | nfft16 | 16ant | 32ant | res1000 | res3000 | res6000 | nfft8192 [reps]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from effex_amd.plan import FxPlan, synth_fill

WORKLOADS = {
    # name: (n_ant, nchan, ntaps, num_samp, n_chunks, mode, rows)
    "8ant": (8, 4096, 4, 2 ** 18, 512, "SPECTRUM", False),          # BASELINE configs[4]
    "stream1": (2, 1, 4, 2 ** 20, 2048, "CONTINUUM", True),          # BASELINE configs[2](i)
    "nfft2048": (2, 2048, 4, 2 ** 18, 10000, "SPECTRUM", False),     # --nfft 2048, tiled ring kernel
    "taps32": (2, 2048, 32, 2 ** 18, 1024, "SPECTRUM", False),       # the reference test's taps = 32 shape
    "nfft256": (2, 256, 4, 2 ** 18, 10000, "SPECTRUM", False),       # --nfft 256, the wave-local kernel (k_small.h)
    "nfft16": (2, 16, 4, 2 ** 18, 10000, "SPECTRUM", False),         # --nfft 16, one lane per work item
    "16ant": (16, 4096, 4, 2 ** 18, 128, "SPECTRUM", False),         # 120 baselines: F-only tiled kernel + matrix-core X-engine
    "32ant": (32, 4096, 4, 2 ** 18, 64, "SPECTRUM", False),          # 496 baselines
    "res1000": (2, 1000, 4, 2 ** 18, 1024, "SPECTRUM", False),       # --resolution 1000: mixed-radix kernel, F and X in one pass
    "nfft8192": (2, 8192, 4, 2 ** 18, 1024, "SPECTRUM", False),      # --nfft 8192: two passes (f8192_ring_kernel, then its XM form)
    "res6000": (2, 6000, 4, 2 ** 18, 1024, "SPECTRUM", False),       # --resolution 6000: the F stage built for the channel count + xmul_kernel
    "res3000": (2, 3000, 4, 2 ** 18, 1024, "SPECTRUM", False),       # --resolution 3000: the lean build of fx_spec.h (above 2048 channels)
    "res1000t1": (2, 1000, 1, 2 ** 18, 1024, "SPECTRUM", False),     # ... with one tap: every sample read once (calibrates FETCH_SIZE)
}


def main():
    name = sys.argv[1]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    n_ant, nchan, ntaps, num_samp, n_chunks, mode, rows = WORKLOADS[name]
    x = torch.empty((n_chunks, n_ant, num_samp), dtype=torch.complex64, device="cuda")
    synth_fill(x, 1234, delays=None if n_ant <= 8 else [a % 7 for a in range(n_ant)])
    window = np.array([0.4, 0.3, 0.2, 0.1]) if nchan == 1 else None
    with FxPlan(n_ant, nchan, ntaps, num_samp, window=window) as plan:
        for _ in range(reps):
            if rows:
                plan.fx_rows(x, mode, 2.4e6)
            else:
                plan.acc_reset()
                plan.fx_accumulate(x)
                plan.finalize(mode, 2.4e6)
        plan.sync()
    print(name, "bytes_in_per_call", n_chunks * n_ant * num_samp * 8)


if __name__ == "__main__":
    main()
