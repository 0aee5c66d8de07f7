#!/usr/bin/env python3
"""Developer aid: wall time of the drop-in's own per-pair call path (effex.py:391-395, 490-527) per chunk pair:

* ``_run_task()`` on the pinned ``gpu_iq_0/1`` staging buffers (host complex64 in, host row out -- the reference's call),
* ``_stage(pair)`` + ``_run_task()`` from a complex128 source pair (narrowing on the host, DC removal on the device),
* the same call on pageable arrays bound to ``gpu_iq_0/1`` (what round 3 measured), and on the receivers' bytes,
* where the time goes: the host's narrowing pass, ``fxc_fx_rows`` on pinned / pageable buffers, complex128 handed over as it
  is (FXC_IQ_C128, narrowed on the device) against narrowing on the host."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def timed(fn, n=200, warm=10):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    from effex_amd import synth
    from effex_amd.correlator import Correlator, SyntheticSource
    from effex_amd.plan import pinned_empty
    x = synth.synth_iq(5, 1, 2, 2 ** 18)[0].astype(np.complex128) + (0.1 - 0.2j)
    x64 = x.astype(np.complex64)
    for mode in ("SPECTRUM", "CONTINUUM"):
        cor = Correlator(source=SyntheticSource(), mode=mode)
        try:
            cor._state = 'RUN'
            cor.gpu_iq_0[:] = x[0]
            cor.gpu_iq_1[:] = x[1]
            res = {"mode": mode, "pinned_staging": bool(cor._pinned)}
            res["run_task_pinned_ms"] = round(timed(cor._run_task), 4)
            res["stage_c128_plus_run_task_ms"] = round(timed(lambda: (cor._stage((x[0], x[1])), cor._run_task())), 4)
            res["of_which_stage_ms"] = round(timed(lambda: cor._stage((x[0], x[1]))), 4)
            cor.gpu_iq_0, cor.gpu_iq_1 = x64[0].copy(), x64[1].copy()            # pageable arrays bound by the caller
            res["run_task_rebound_pageable_c64_ms"] = round(timed(cor._run_task), 4)
            cor.gpu_iq_0, cor.gpu_iq_1 = x[0].copy(), x[1].copy()
            res["run_task_rebound_pageable_c128_ms"] = round(timed(cor._run_task), 4)
            plan = cor._plan()
            lib_mode = 'SPECTRUM' if mode == 'SPECTRUM' else 'CONTINUUM'
            pin = pinned_empty((1, 2, x.shape[1]), np.complex64)
            pin[0] = x64
            page = np.ascontiguousarray(pin)
            out = pinned_empty((1, 1, 4096), np.complex64) if mode == 'SPECTRUM' else pinned_empty((1, 1), np.complex128)
            res["fx_rows_pinned_in_pinned_out_ms"] = round(timed(lambda: plan.fx_rows(pin, lib_mode, cor.bandwidth, out=out)), 4)
            res["fx_rows_pinned_in_dc_ms"] = round(timed(lambda: plan.fx_rows(pin, lib_mode, cor.bandwidth, remove_dc=True, out=out)), 4)
            res["fx_rows_pinned_in_pageable_out_ms"] = round(timed(lambda: plan.fx_rows(pin, lib_mode, cor.bandwidth)), 4)
            res["fx_rows_pageable_ms"] = round(timed(lambda: plan.fx_rows(page, lib_mode, cor.bandwidth)), 4)
            pin128 = pinned_empty((1, 2, x.shape[1]), np.complex128)
            pin128[0] = x
            res["fx_rows_c128_narrowed_on_device_ms"] = round(timed(
                lambda: plan.fx_rows(pin128, lib_mode, cor.bandwidth, remove_dc=True, out=out, c128=True)), 4)
            res["host_narrow_c128_to_pinned_c64_ms"] = round(timed(lambda: np.copyto(pin[0], x, casting='same_kind')), 4)
            res["Msamples_per_s_run_task"] = round(2 ** 18 / res["run_task_pinned_ms"] / 1e3, 1)
            res["x_realtime_at_2.4Msps"] = round(2 ** 18 / (res["stage_c128_plus_run_task_ms"] * 1e-3) / 2.4e6, 1)
            if mode == "SPECTRUM":
                # the same call fed the receivers' bytes (FileSource / SocketSource hand these over, effex.py:652 on the device)
                rng = np.random.default_rng(3)
                b = rng.integers(0, 256, size=(2, x.shape[1], 2), dtype=np.uint8)
                res["stage_u8_plus_run_task_ms"] = round(timed(lambda: (cor._stage((b[0], b[1])), cor._run_task())), 4)
            print(json.dumps(res))
        finally:
            cor.close()


if __name__ == "__main__":
    main()
