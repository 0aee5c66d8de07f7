#!/usr/bin/env python3
"""Developer aid: run the tiled F+X and F-only kernels once per shape (for rocprofv3 --pmc SQ_LDS_BANK_CONFLICT ...)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from effex_amd.plan import FxPlan, synth_fill

num_samp = 262144
x = torch.empty((256, 2, num_samp), dtype=torch.complex64, device="cuda")
synth_fill(x, 1234)
for nchan in (512, 1024, 2048, 4096):
    with FxPlan(2, nchan, 4, num_samp, path="tiled") as plan:
        plan.fx_accumulate(x)
        plan.sync()
    with FxPlan(1, nchan, 4, num_samp) as plan:
        out = plan.channelize(x.view(512, num_samp))
        plan.sync()
        del out
